"""Weight ingest for one MoE layer: checkpoint tensors -> the tensors `lkm_create` takes.

Replaces, for the routed-expert path only, what the reference spreads over
  RoutedExperts.weight_loader / _load_w13 / _load_w2 / _load_model_weight_or_group_weight_scale
      (vllm/model_executor/layers/fused_moe/routed_experts.py:383-612, 644-967),
  the per-quantisation create_weights (compressed_tensors_moe_wna16.py:239-420, fp8.py:524-668) and
  the _process_* hand-off to lk_moe (routed_experts.py:1440-1813).

The reference first fills vLLM parameter tensors in whatever orientation the GPU kernels of that
quantisation method want (compressed-tensors int4 is stored transposed: [E, K/8, N] int32) and then
re-transposes / re-views them for lk_moe.  Here the staging buffers already ARE the lk_moe layouts
    w13 [E_local, 2*I_p, K...]   w2 [E_local, H, K...]      (rows = output features, K contiguous)
so a checkpoint tensor is narrowed for TP, viewed as bytes where needed and copied once; the result is
bit-identical to the reference's two-step route (tests/test_ingest.py against tests/golden/ingest.npz,
which was produced by running the reference's own loader helpers).

Sharding rules kept from the reference:
  * EP: global expert id -> local id through determine_expert_map; non-local experts are skipped
    (weight_loader returns False), routed_experts.py:678-688.
  * TP: w1/w3 ("MergedColumnParallel") are narrowed on the OUTPUT dim, w2 ("RowParallel") on the
    INPUT dim; the per-rank slice is loaded.shape[dim] // tp_size starting at tp_rank * that
    (routed_experts.py:545-560, 592-603) -- for packed formats the division happens in packed units,
    which is the same slice because the packing runs along K.
  * w1 -> first half of w13, w3 -> second half (routed_experts.py:561-568).

torch is used for host tensors only; the engine copies and pre-shuffles them (lkm_create).
"""
from __future__ import annotations

import re
from typing import Iterable

import torch

from . import ops

_QUANTS = ("none", "fp8_block", "int4", "mxfp4", "nvfp4")
_SHARD_OF = {"w1": "w1", "w2": "w2", "w3": "w3"}


class IngestError(RuntimeError):
    pass


class ExpertWeightIngest:
    """Stages the routed-expert tensors of ONE MoE layer for one (tp_rank, ep_rank).

    quant: "none" (params_dtype weights), "fp8_block" (e4m3fn + fp32 block scales `weight_scale_inv`),
           "int4" (compressed-tensors pack-quantized uint4b8: `weight_packed` int32 [N, K/8] +
           `weight_scale` [N, K/group]), "mxfp4" (`weight_packed`/`weight` uint8 [N, K/2] + E8M0
           `weight_scale` [N, K/32]), "nvfp4" (uint8 [N, K/2] + e4m3fn `weight_scale` [N, K/16] +
           per-tensor `weight_global_scale` / `weight_scale_2`).
    """

    def __init__(self, *, num_experts: int, hidden_size: int, intermediate_size: int,
                 params_dtype: torch.dtype = torch.bfloat16, quant: str = "none", group_size: int = 128,
                 block: tuple[int, int] = (128, 128), tp_size: int = 1, tp_rank: int = 0, ep_size: int = 1,
                 ep_rank: int = 0, expert_placement: str = "linear", has_gate_proj: bool = True,
                 global_scale_is_divisor: bool = True,
                 ckpt_names: tuple[str, str, str] = ("gate_proj", "down_proj", "up_proj")):
        if quant not in _QUANTS:
            raise IngestError(f"quant must be one of {_QUANTS}, got {quant!r}")
        if intermediate_size % tp_size:
            raise IngestError(f"intermediate_size={intermediate_size} is not divisible by tp_size={tp_size}")
        self.E_global, self.H, self.I_full = num_experts, hidden_size, intermediate_size
        self.I = intermediate_size // tp_size
        self.dtype, self.quant = params_dtype, quant
        self.tp_size, self.tp_rank = tp_size, tp_rank
        self.has_gate = has_gate_proj
        self.halves = 2 if has_gate_proj else 1
        self.global_scale_is_divisor = global_scale_is_divisor
        self.E, self.expert_map = ops.determine_expert_map(ep_size, ep_rank, num_experts, expert_placement)
        gate, down, up = ckpt_names
        self._proj_to_shard = {gate: "w1", down: "w2", up: "w3", "w1": "w1", "w2": "w2", "w3": "w3"}
        E, H, Ip, hv = self.E, self.H, self.I, self.halves
        self.group = {"none": 0, "fp8_block": block[1], "int4": group_size, "mxfp4": 32, "nvfp4": 16}[quant]
        self.block_n = block[0] if quant == "fp8_block" else 1
        t: dict[str, torch.Tensor] = {}
        if quant == "none":
            t["w13"] = torch.zeros((E, hv * Ip, H), dtype=params_dtype)
            t["w2"] = torch.zeros((E, H, Ip), dtype=params_dtype)
        elif quant == "fp8_block":
            bn, bk = block
            if Ip % bn or Ip % bk or H % bn or H % bk:
                raise IngestError(f"fp8 block {block}: hidden={H} and intermediate/tp={Ip} must be multiples of the block")
            t["w13"] = torch.zeros((E, hv * Ip, H), dtype=torch.uint8)
            t["w2"] = torch.zeros((E, H, Ip), dtype=torch.uint8)
            t["s13"] = torch.zeros((E, hv * Ip // bn, H // bk), dtype=torch.float32)
            t["s2"] = torch.zeros((E, H // bn, Ip // bk), dtype=torch.float32)
        else:
            g = self.group
            if Ip % max(g, 8) or H % max(g, 8):
                raise IngestError(f"{quant}: hidden={H} and intermediate/tp={Ip} must be multiples of the group size {g}")
            sdt = {"int4": params_dtype, "mxfp4": torch.uint8, "nvfp4": torch.uint8}[quant]
            t["w13"] = torch.zeros((E, hv * Ip, H // 2), dtype=torch.uint8)
            t["w2"] = torch.zeros((E, H, Ip // 2), dtype=torch.uint8)
            t["s13"] = torch.zeros((E, hv * Ip, H // g), dtype=sdt)
            t["s2"] = torch.zeros((E, H, Ip // g), dtype=sdt)
            if quant == "nvfp4":
                t["gs13"] = torch.zeros((E, hv), dtype=torch.float32)     # per logical shard, merged at build
                t["gs2"] = torch.zeros((E,), dtype=torch.float32)
        self.t = t
        self._seen: set[tuple[int, str, str]] = set()

    # ------------------------------------------------------------------ reference surface
    def map_global_to_local(self, expert_id: int) -> int:
        """expert_map_manager.map_global_to_local: -1 when the expert lives on another EP rank."""
        if self.expert_map is None:
            return expert_id if 0 <= expert_id < self.E_global else -1
        return int(self.expert_map[expert_id])

    def weight_loader(self, loaded_weight: torch.Tensor, weight_name: str, shard_id: str, expert_id: int) -> bool:
        """One checkpoint tensor of one GLOBAL expert (cf. RoutedExperts.weight_loader, return_success=True).
        weight_name only needs to END in the tensor kind (`weight`, `weight_packed`, `weight_scale`,
        `weight_scale_inv`, `weight_global_scale`, `weight_scale_2`, `weight_shape`, ...)."""
        if shard_id not in _SHARD_OF:
            raise IngestError(f"shard_id must be ['w1','w2','w3'] but got {shard_id}.")
        if shard_id == "w3" and not self.has_gate:
            raise IngestError("w3 (up_proj) tensor for a layer without gate projection")
        le = self.map_global_to_local(expert_id)
        if le < 0:
            return False
        kind = weight_name.rsplit(".", 1)[-1]
        if kind in ("weight_shape", "input_scale", "input_global_scale"):
            return True                      # carried by the checkpoint, not needed by the W4A16/W8A16 engine
        if kind in ("weight_g_idx", "g_idx"):
            if not torch.equal(loaded_weight.cpu().to(torch.int64).sort().values, loaded_weight.cpu().to(torch.int64)):
                raise IngestError("activation-ordered (g_idx) int4 checkpoints are not supported")
            return True
        if kind in ("weight_zero_point", "qzeros"):
            raise IngestError("asymmetric int4 (zero points) is not supported: the engine implements uint4b8")
        dst_w, dst_s = ("w13", "s13") if shard_id != "w2" else ("w2", "s2")
        is_scale = "scale" in kind
        if kind in ("weight_global_scale", "weight_scale_2"):
            if self.quant != "nvfp4":
                raise IngestError(f"{kind} in a {self.quant} layer")
            v = float(loaded_weight.reshape(()).float())
            v = 1.0 / v if (kind == "weight_global_scale" and self.global_scale_is_divisor) else v
            if shard_id == "w2":
                self.t["gs2"][le] = v
            else:
                self.t["gs13"][le][0 if shard_id == "w1" else 1] = v
            self._seen.add((le, shard_id, "global"))
            return True
        if is_scale and self.quant == "none":
            raise IngestError(f"{kind} in an unquantised layer")
        lw = loaded_weight
        if not is_scale:
            lw = self._as_weight_bytes(lw, shard_id)
        else:
            lw = self._as_scale(lw)
        dst = self.t[dst_s if is_scale else dst_w][le]
        self._copy_shard(dst, lw, shard_id)
        self._seen.add((le, shard_id, "scale" if is_scale else "weight"))
        return True

    def load_weights(self, weights: Iterable[tuple[str, torch.Tensor]]) -> list[str]:
        """(name, tensor) pairs of ONE layer's experts, per-expert (`...experts.<id>.<proj>.<kind>`) or fused
        3-D (`...experts.gate_up_proj`, `...experts.down_proj`; cf. RoutedExperts.load_weights :977-1040).
        Returns the names consumed (names that are not routed-expert tensors are ignored)."""
        used = []
        for name, tensor in weights:
            m = re.search(r"experts\.(\d+)\.([A-Za-z0-9_]+)\.(.+)$", name)
            if m:
                proj = m.group(2)
                if proj not in self._proj_to_shard:
                    continue
                self.weight_loader(tensor, m.group(3), self._proj_to_shard[proj], int(m.group(1)))
                used.append(name)
                continue
            m = re.search(r"experts\.(gate_up_proj|w13|down_proj|w2)(?:\.(.+))?$", name)
            if m and tensor.dim() == 3:
                kind = m.group(2) or "weight"
                fused = tensor
                if m.group(1) in ("gate_up_proj", "w13"):
                    if fused.shape[-1] != self.H and fused.shape[-2] == self.H:
                        fused = fused.transpose(-1, -2)          # _orient_fused_weight :468-487
                    w1, w3 = fused.chunk(2, dim=1)
                    for e in range(fused.shape[0]):
                        self.weight_loader(w1[e], kind, "w1", e)
                        self.weight_loader(w3[e], kind, "w3", e)
                else:
                    if fused.shape[-2] != self.H and fused.shape[-1] == self.H:
                        fused = fused.transpose(-1, -2)
                    for e in range(fused.shape[0]):
                        self.weight_loader(fused[e], kind, "w2", e)
                used.append(name)
        return used

    def load_safetensors(self, paths: Iterable[str], layer_prefix: str = "") -> list[str]:
        """Reads the routed-expert tensors of one layer straight from safetensors shards."""
        from safetensors import safe_open
        used = []
        for path in paths:
            with safe_open(path, framework="pt", device="cpu") as f:
                for name in f.keys():
                    if layer_prefix and not name.startswith(layer_prefix):
                        continue
                    if ".experts." not in name and not name.startswith("experts."):
                        continue
                    used += self.load_weights([(name, f.get_tensor(name))])
        return used

    # ------------------------------------------------------------------ results
    def missing(self) -> list[str]:
        """what a complete layer still lacks (an incomplete layer must fail loudly, not run on zeros)"""
        out = []
        shards = ("w1", "w2", "w3") if self.has_gate else ("w1", "w2")
        kinds = ["weight"] + (["scale"] if self.quant != "none" else []) + (["global"] if self.quant == "nvfp4" else [])
        for le in range(self.E):
            for sid in shards:
                for k in kinds:
                    if (le, sid, k) not in self._seen:
                        out.append(f"local expert {le} {sid} {k}")
        return out

    def tensors(self) -> dict[str, torch.Tensor]:
        """the tensors lkm_create takes: w13, w2 [, s13, s2 [, gs13, gs2]] (host memory)"""
        miss = self.missing()
        if miss:
            raise IngestError(f"{len(miss)} expert tensors were never loaded, e.g. {miss[:4]}")
        out = dict(self.t)
        if self.quant == "nvfp4":
            g = self.t["gs13"]
            if self.has_gate and not torch.equal(g[:, 0], g[:, 1]):
                # one multiplier per expert for the merged w13 (routed_experts.py:1681 takes a single
                # [E] tensor; modelopt/compressed-tensors checkpoints carry equal w1/w3 global scales)
                raise IngestError("w1 and w3 global scales of an expert differ; the merged w13 needs one")
            out["gs13"] = g[:, 0].contiguous()
        return out

    def build_engine(self, *, top_k: int, **kw):
        """process_weights_after_loading + _process_* : hands the staged tensors to the engine."""
        t = self.tensors()
        fmt = {"none": "bf16" if self.dtype == torch.bfloat16 else "fp16", "fp8_block": "fp8", "int4": "int4",
               "mxfp4": "mxfp4", "nvfp4": "nvfp4"}[self.quant]
        args = dict(top_k=top_k, act_dtype=self.dtype, fmt=fmt, has_gate_proj=self.has_gate)
        if self.quant != "none":
            args.update(w13_scale=t["s13"], w2_scale=t["s2"], group_n=self.block_n, group_k=self.group)
        if self.quant == "nvfp4":
            args.update(w13_global_scale=t["gs13"], w2_global_scale=t["gs2"])
        if not self.has_gate:
            args.setdefault("activation_type", 2)
        args.update(kw)
        return ops.RoutedExpertsEngine(t["w13"], t["w2"], **args)

    # ------------------------------------------------------------------ helpers
    def _as_weight_bytes(self, w: torch.Tensor, shard_id: str) -> torch.Tensor:
        """checkpoint weight [N, K*] -> the engine's element layout ([N, K] dtype / uint8 bytes)"""
        n_full = self.H if shard_id == "w2" else self.I_full
        if w.dim() != 2:
            raise IngestError(f"expected a 2-D expert weight, got shape {tuple(w.shape)}")
        if w.shape[0] != n_full and w.shape[1] == n_full and self.quant == "none":
            w = w.t()                                   # [K, N] checkpoints (e.g. transposed fused slices)
        if self.quant == "none":
            return w.to(self.dtype)
        if self.quant == "fp8_block":
            if w.dtype != torch.uint8:
                if w.dtype != torch.float8_e4m3fn:
                    raise IngestError(f"fp8 weight has dtype {w.dtype}")
                w = w.contiguous().view(torch.uint8)
            return w
        # 4-bit: int32 [N, K/8] (little-endian nibbles = k order) or uint8 [N, K/2]
        if w.dtype == torch.int32:
            w = w.contiguous().view(torch.uint8)
        if w.dtype != torch.uint8:
            raise IngestError(f"packed 4-bit weight has dtype {w.dtype} (expected int32 or uint8)")
        return w

    def _as_scale(self, s: torch.Tensor) -> torch.Tensor:
        if self.quant == "fp8_block":
            return s.to(torch.float32)
        if self.quant == "int4":
            return s.to(self.dtype)
        if s.dtype in (torch.float8_e4m3fn,) or (hasattr(torch, "float8_e8m0fnu") and s.dtype == torch.float8_e8m0fnu):
            s = s.contiguous().view(torch.uint8)
        if s.dtype != torch.uint8:
            raise IngestError(f"{self.quant} block scales have dtype {s.dtype} (expected 1-byte floats)")
        return s

    def _copy_shard(self, dst: torch.Tensor, lw: torch.Tensor, shard_id: str) -> None:
        """_load_w13 / _load_w2 (routed_experts.py:528-612) on the final layout: narrow for TP, then copy into
        the w1 / w3 half (or all of w2)."""
        dim = 1 if shard_id == "w2" else 0
        per_rank = lw.shape[dim] // self.tp_size
        start = per_rank * self.tp_rank
        avail = lw.shape[dim] - start
        if avail <= 0:
            return
        lw = lw.narrow(dim, start, min(per_rank, avail))
        if shard_id != "w2":
            half = dst.shape[0] // self.halves
            dst = dst.narrow(0, 0 if shard_id == "w1" else half, half)
        if tuple(lw.shape) != tuple(dst.shape):
            raise IngestError(f"{shard_id}: checkpoint slice {tuple(lw.shape)} does not fit the parameter {tuple(dst.shape)}")
        dst.copy_(lw)
