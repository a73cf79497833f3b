"""Expert-parallel sharding of the routed experts across the GPUs of one MI355X node.

This replaces the reference's CPU-NUMA offload tier: instead of parking experts in host DRAM
(lk_moe) each rank keeps E/ep experts resident in its 288 GB of HBM.  Two data paths:

  "a2a"  (default, BASELINE.json north_star)  tokens stay DP-sharded; routed rows + (local id,
         weight) travel by all_to_all_single -> local grouped GEMMs -> reverse all_to_all -> local
         sum.  On the 8-GPU xGMI full mesh all 7 links of a GPU carry traffic concurrently
         (SURVEY 8e).  Two flavours, chosen by size:
           fixed   (decode, M*K <= fixed_capacity_slots): every peer gets M*K row slots, unrouted
                   slots carry id -1; equal splits => NO split-size exchange and NO host sync.  At
                   decode sizes the step is latency-bound (0.5 MB per peer ~ 4 us of xGMI time),
                   so padding is free and removing the synchronisation is what matters.
           ragged  (prefill): exact split sizes are exchanged first, only routed rows travel.
  "ar"   reference-compatible mode (what LvLLM does today, moe_runner.py:600,494 and
         routed_experts.py:1332-1342): every rank sees ALL tokens (all_gather), computes only
         its local experts (other ids -> -1), and the [M,H] partial outputs are summed with
         reduce_scatter (= the reference's all-reduce, kept sharded).

Placement is the reference's linear map (expert_map_manager.py:62-79).  With EPLB (lvllm_amd/eplb.py) the ids
handed to forward() are PHYSICAL expert ids and `num_experts` is the number of physical slots: the linear map over
the slots is exactly the EPLB placement (slot p lives on rank p // (P / ep)); tests/test_eplb.py runs that end to end.  `torch.distributed` with
backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests (tests/test_ep_gloo.py),
where the local expert computation is injected.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist

from .ops import determine_expert_map

# local_compute(rows [R,H] act dtype, local_ids int32 [R,1], weights fp32 [R,1]) -> fp32 [R,H]
LocalCompute = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]
# pack(hidden [M,H], weights [M,K], ids [M,K], num_experts, ep) -> (send_x [ep,MK,H], send_ids [ep,MK], send_w [ep,MK])
PackFn = Callable[[torch.Tensor, torch.Tensor, torch.Tensor, int, int], tuple]


def owner_of(ids: torch.Tensor, num_experts: int, ep_size: int) -> torch.Tensor:
    """rank that owns each global expert id under linear placement (remainder to the first ranks,
    expert_map_manager.py:66-74); ids < 0 -> -1."""
    base, rem = divmod(num_experts, ep_size)
    idl = ids.to(torch.int64)
    cut = rem * (base + 1)
    if base > 0:
        r = torch.where(idl < cut, idl // (base + 1), rem + (idl - cut) // base)
    else:
        r = idl // (base + 1)
    return torch.where(idl < 0, torch.full_like(idl, -1), r)


class ExpertParallelExperts:
    def __init__(self, local_compute: LocalCompute, num_experts: int, hidden_size: int,
                 group: dist.ProcessGroup | None = None, mode: str = "a2a",
                 pack: PackFn | None = None, fixed_capacity_slots: int = 2048):
        if mode not in ("a2a", "ar"):
            raise ValueError(f"unknown EP mode {mode!r} (expected 'a2a' or 'ar')")
        self.local_compute = local_compute
        if pack is None:
            from .ops import ep_pack as pack       # the HIP kernel (no CPU path)
        self.pack = pack
        self.fixed_capacity_slots = fixed_capacity_slots
        self.E, self.H = num_experts, hidden_size
        self.group = group
        self.ep = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.mode = mode
        self.local_num, emap = determine_expert_map(self.ep, self.rank, num_experts, "linear")
        self.expert_map = emap if emap is not None else torch.arange(num_experts, dtype=torch.int32)
        base, rem = divmod(num_experts, self.ep)
        self.first_expert = [r * base + min(r, rem) for r in range(self.ep)]

    # -------------------------------------------------------------------------------- a2a, fixed
    def dispatch_fixed(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor):
        """tokens -> owners: (rows [ep*M*K, H], local ids int32 [ep*M*K] (-1 = empty slot), weights fp32 [ep*M*K]).
        Equal splits: no split-size exchange and no host sync."""
        send_x, send_ids, send_w = self.pack(hidden, tw, ids, self.E, self.ep)
        recv_x = torch.empty_like(send_x)
        recv_ids = torch.empty_like(send_ids)
        recv_w = torch.empty_like(send_w)
        dist.all_to_all_single(recv_x, send_x, group=self.group)
        dist.all_to_all_single(recv_ids, send_ids, group=self.group)
        dist.all_to_all_single(recv_w, send_w, group=self.group)
        n = ids.numel()
        return recv_x.view(self.ep * n, self.H), recv_ids.view(self.ep * n), recv_w.view(self.ep * n)

    def combine_fixed(self, y: torch.Tensor, M: int, K: int) -> torch.Tensor:
        """owners -> tokens: y fp32 [ep*M*K, H] (weighted rows, zeros in empty slots) -> fp32 [M, H]"""
        n = M * K
        y_back = torch.empty((self.ep, n, self.H), dtype=torch.float32, device=y.device)
        dist.all_to_all_single(y_back, y.view(self.ep, n, self.H).contiguous(), group=self.group)
        # every slot was computed by exactly one rank (rows of the others are zero): fixed-order fp32 sum
        return y_back.view(self.ep, M, K, self.H).sum(dim=(0, 2))

    def _forward_a2a_fixed(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        M, K = ids.shape
        rows, lids, ws = self.dispatch_fixed(hidden, tw, ids)
        y = self.local_compute(rows, lids.view(-1, 1), ws.view(-1, 1))
        return self.combine_fixed(y, M, K)

    # -------------------------------------------------------------------------------- a2a, ragged
    def _forward_a2a(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        M, K = ids.shape
        dev = hidden.device
        flat_ids = ids.reshape(-1)
        owner = owner_of(flat_ids, self.E, self.ep)                       # [M*K], -1 = dropped
        # stable order by destination rank; dropped slots last
        key = torch.where(owner < 0, torch.full_like(owner, self.ep), owner)
        order = torch.argsort(key, stable=True)
        send_counts = torch.bincount(key, minlength=self.ep + 1)[: self.ep]
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()              # host sync: split sizes
        n_send, n_recv = sum(sc), sum(rc)
        order = order[:n_send]
        tok = torch.div(order, K, rounding_mode="floor")
        x_send = hidden.index_select(0, tok)
        first = torch.tensor(self.first_expert, device=dev, dtype=torch.int64)
        lid_send = (flat_ids[order].to(torch.int64) - first[owner[order]]).to(torch.int32)
        w_send = tw.reshape(-1)[order].contiguous()
        x_recv = torch.empty((n_recv, self.H), dtype=hidden.dtype, device=dev)
        lid_recv = torch.empty((n_recv,), dtype=torch.int32, device=dev)
        w_recv = torch.empty((n_recv,), dtype=torch.float32, device=dev)
        dist.all_to_all_single(x_recv, x_send, rc, sc, group=self.group)
        dist.all_to_all_single(lid_recv, lid_send, rc, sc, group=self.group)
        dist.all_to_all_single(w_recv, w_send, rc, sc, group=self.group)
        # local experts: each received row is one (token, slot) pair => top_k = 1
        y_recv = self.local_compute(x_recv, lid_recv.view(-1, 1), w_recv.view(-1, 1))
        y_back = torch.empty((n_send, self.H), dtype=torch.float32, device=dev)
        dist.all_to_all_single(y_back, y_recv.contiguous(), sc, rc, group=self.group)
        out = torch.zeros((M, self.H), dtype=torch.float32, device=dev)
        out.index_add_(0, tok, y_back)
        return out

    # -------------------------------------------------------------------------------- ar
    def _forward_ar(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        M, K = ids.shape
        dev = hidden.device
        ep = self.ep
        xs = torch.empty((ep * M, self.H), dtype=hidden.dtype, device=dev)
        ii = torch.empty((ep * M, K), dtype=torch.int32, device=dev)
        ww = torch.empty((ep * M, K), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(xs, hidden.contiguous(), group=self.group)
        dist.all_gather_into_tensor(ii, ids.contiguous(), group=self.group)
        dist.all_gather_into_tensor(ww, tw.contiguous(), group=self.group)
        emap = self.expert_map.to(dev)
        local = torch.where(ii < 0, torch.full_like(ii, -1),
                            emap[ii.clamp(0, self.E - 1).to(torch.int64)])    # routed_experts.py:1332-1342
        part = self.local_compute(xs, local.contiguous(), ww)                   # [ep*M, H] fp32
        out = torch.empty((M, self.H), dtype=torch.float32, device=dev)
        dist.reduce_scatter_tensor(out, part.contiguous(), group=self.group)
        return out

    def forward(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                force_collectives: bool = False) -> torch.Tensor:
        """hidden [M,H] (this rank's tokens), GLOBAL expert ids int32 [M,K] -> fp32 [M,H].
        force_collectives runs the collective data path even on a single rank (plumbing checks)."""
        if self.ep == 1 and not force_collectives:
            return self.local_compute(hidden, topk_ids, topk_weights)
        if self.mode == "a2a":
            if topk_ids.numel() <= self.fixed_capacity_slots:
                return self._forward_a2a_fixed(hidden, topk_weights, topk_ids)
            return self._forward_a2a(hidden, topk_weights, topk_ids)
        return self._forward_ar(hidden, topk_weights, topk_ids)
