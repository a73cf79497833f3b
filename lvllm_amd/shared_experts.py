"""Shared experts folded into the routed grouped GEMM (second half).

The reference runs a model's shared expert (GLM-4.5-Air: +1, DeepSeek-V3: +1, Qwen2-MoE: gated) as a separate
dense MLP and hides it behind the routed experts on an auxiliary stream
(vllm/model_executor/layers/fused_moe/runner/shared_experts.py:39-174: MULTI_STREAM_OVERLAPPED), or -- on its ROCm
path -- appends it to the routed experts as extra always-selected experts
(`num_fused_shared_experts`: experts/rocm_aiter_moe.py:61-158 `init_aiter_topK_meta_data` /
`inject_shared_expert_weights`, layer.py:84-96, expert_map_manager.py:27-113,
router/aiter_shared_routed_fused_moe_router.py).  On MI355X the second form is the natural one: the decode step is
bound by streaming weights from HBM, a second stream cannot add bandwidth, and the shared expert's rows are just
more (expert, token-tile) work items of the SAME launches -- no extra launches, no stream join, and its output is
summed in the same fp32 combine.

What makes it exact: a gated MLP is separable over its intermediate dimension,
    down( act(gate x) * (up x) ) = sum over chunks c of  down[:, c] ( act(gate[c] x) * (up[c] x) ),
so a shared expert of intermediate size n*I IS n experts of size I with routing weight 1 (or the per-token sigmoid
gate of Qwen2-MoE) each.  `split_shared_expert` cuts the weights (and block / group scales) accordingly,
`append_shared_experts` places them behind the routed experts, `SharedExpertSlots` extends (weights, ids) from
[M, K] to [M, K + n] the way the reference's buffers do.  Host logic only; the engine sees E + n experts, top-(K + n).
"""
from __future__ import annotations

import torch

__all__ = ["split_shared_expert", "append_shared_experts", "SharedExpertSlots", "shared_expert_map_tail"]


def _split_rows(t: torch.Tensor, n: int, halves: int) -> torch.Tensor:
    """[halves * n * r, ...] (gate rows then up rows, each n chunks of r) -> [n, halves * r, ...]"""
    rows = t.size(0)
    if rows % (halves * n):
        raise ValueError(f"{rows} rows do not split into {halves} x {n} chunks")
    r = rows // (halves * n)
    parts = t.reshape(halves, n, r, *t.shape[1:])
    return parts.transpose(0, 1).reshape(n, halves * r, *t.shape[1:]).contiguous()


def _split_cols(t: torch.Tensor, n: int) -> torch.Tensor:
    """[rows, n * c] -> [n, rows, c]"""
    rows, cols = t.shape
    if cols % n:
        raise ValueError(f"{cols} columns do not split into {n} chunks")
    return t.reshape(rows, n, cols // n).transpose(0, 1).contiguous()


def split_shared_expert(w13: torch.Tensor, w2: torch.Tensor, n: int, *, has_gate_proj: bool = True,
                        w13_scale: torch.Tensor | None = None, w2_scale: torch.Tensor | None = None):
    """One shared expert of intermediate size n*I -> n experts of size I.

    w13 [2*n*I, H] ("gate rows then up rows", routed_experts.py:563-569; [n*I, H] without a gate), w2 [H, n*I];
    packed 4-bit weights keep their byte layout (w13 [2nI, H/2], w2 [H, nI/2]: two k per byte, so I must be even).
    Scales follow their tensor: w13 scales [rows / gN, H / gK] split by rows, w2 scales [H / gN, nI / gK] by columns
    -- I must be a multiple of the block / group size along the intermediate dimension (checked by divisibility).
    Returns (w13 [n, 2I, H], w2 [n, H, I], w13_scale | None, w2_scale | None)."""
    if w13.dim() != 2 or w2.dim() != 2:
        raise ValueError("expected one shared expert: w13 [rows, H], w2 [H, cols]")
    halves = 2 if has_gate_proj else 1
    return (_split_rows(w13, n, halves), _split_cols(w2, n),
            None if w13_scale is None else _split_rows(w13_scale, n, halves),
            None if w2_scale is None else _split_cols(w2_scale, n))


def append_shared_experts(routed: torch.Tensor, shared: torch.Tensor) -> torch.Tensor:
    """[E, ...] routed experts + [n, ...] shared chunks -> [E + n, ...]: the shared experts take the ids
    E .. E+n-1 (rocm_aiter_moe.py:87 `n_routed_experts + i`; on an EP rank: local_num_experts + i,
    expert_map_manager.py:100-110)."""
    if routed.shape[1:] != shared.shape[1:] or routed.dtype != shared.dtype:
        raise ValueError(f"shared experts {tuple(shared.shape)} {shared.dtype} do not match the routed experts "
                         f"{tuple(routed.shape)} {routed.dtype} (same format and per-expert shape required)")
    return torch.cat((routed, shared.to(routed.device)), dim=0)


def shared_expert_map_tail(local_num_experts: int, num_fused_shared_experts: int) -> torch.Tensor:
    """What an EP rank appends to its global -> local expert map: the shared experts sit behind its local routed
    experts (expert_map_manager.py:100-110), followed by ONE sentinel entry -1 for the id `n_routed + n_shared`
    that marks "not this rank's token" (rocm_aiter_moe.py:72-93; the reference masks it with expert_mask[-1] = 0)."""
    return torch.tensor([local_num_experts + i for i in range(num_fused_shared_experts)] + [-1], dtype=torch.int32)


class SharedExpertSlots:
    """(topk_weights, topk_ids) [M, K] -> [M, K + n (+1 under EP)] with the shared experts' slots filled in.

    One pre-built buffer pair serves every layer and every step (as `init_aiter_topK_meta_data` does,
    rocm_aiter_moe.py:61-110): the shared columns are constant -- ids n_routed .. n_routed+n-1, weight
    `shared_experts_score` -- so a step only copies the K routed columns in (two strided copies, graph-capturable)
    and, for a gated shared expert, the per-token gate values (`inject_shared_expert_weights`, :113-158).

    Under expert parallelism with replicated tokens (`is_ep`, the reference's TP/EP mode) every rank holds the shared
    expert but each token must be served once: token i belongs to rank i % ep_size; on the other ranks its shared
    slots (and the one extra sentinel column) carry the id n_routed + n_shared, which the expert map sends to -1
    (`shared_expert_map_tail`)."""

    def __init__(self, n_routed_experts: int, n_shared_experts: int, top_k: int, *, ep_rank: int = 0, ep_size: int = 1,
                 is_ep: bool = False, shared_experts_score: float = 1.0, max_num_tokens: int = 32768,
                 device: torch.device | str = "cpu"):
        if n_shared_experts <= 0:
            raise ValueError("n_shared_experts must be positive")
        self.n_routed, self.n_shared, self.top_k = n_routed_experts, n_shared_experts, top_k
        self.shared_experts_score = float(shared_experts_score)
        self.extra = n_shared_experts + int(is_ep)
        self.fake_id = n_routed_experts + n_shared_experts
        width = top_k + self.extra
        ids = torch.empty((max_num_tokens, width), dtype=torch.int32)
        own = torch.arange(n_routed_experts, n_routed_experts + self.extra, dtype=torch.int32)
        if is_ep:
            ids[:, top_k:] = self.fake_id
            ids[ep_rank::ep_size, top_k:] = own          # own[-1] == fake_id: the sentinel column stays masked
        else:
            ids[:, top_k:] = own
        ids[:, :top_k] = 0
        w = torch.zeros((max_num_tokens, width), dtype=torch.float32)
        w[:, top_k:] = shared_experts_score
        self.total_topk_ids = ids.to(device)
        self.total_topk_weights = w.to(device)

    @property
    def width(self) -> int:
        return self.top_k + self.extra

    def inject(self, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
               shared_expert_weights: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
        """-> views [M, K + n (+1)] of the shared buffers (valid until the next inject)"""
        M = topk_weights.size(0)
        if M > self.total_topk_ids.size(0):
            raise ValueError(f"shared-expert slot buffers hold {self.total_topk_ids.size(0)} tokens, got {M}")
        if topk_ids.size(1) != self.top_k or topk_weights.shape != topk_ids.shape:
            raise ValueError(f"expected routed results [M, {self.top_k}]")
        w, ids = self.total_topk_weights[:M], self.total_topk_ids[:M]
        w[:, :self.top_k] = topk_weights
        ids[:, :self.top_k] = topk_ids
        if shared_expert_weights is not None:
            w[:, self.top_k:self.top_k + self.n_shared] = shared_expert_weights[:M]
        return w, ids
