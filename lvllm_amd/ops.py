"""Torch-tensor front-ends of the router / scatter kernels and of the expert engine.

These mirror the reference's *operator* signatures (same names, argument meaning, error
behaviour) so the parity tests read like the reference's own tests; torch is used only for
device memory and the current stream.  Everything executes in liblkm.so -- no eager fallback.

Reference interfaces mirrored (relative to the reference tree):
  fused_topk        vllm/model_executor/layers/fused_moe/router/fused_topk_router.py:81-125
  fused_topk_bias   vllm/model_executor/layers/fused_moe/router/fused_topk_bias_router.py
  grouped_topk      vllm/model_executor/layers/fused_moe/router/grouped_topk_router.py:80-161
  global_to_local_expert_ids   vllm/model_executor/layers/fused_moe/routed_experts.py:1332-1342
  determine_expert_map         vllm/model_executor/layers/fused_moe/expert_map_manager.py:22-92
  eplb_map_to_physical_and_record   vllm/model_executor/layers/fused_moe/router/base_router.py:24-146
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _clib
from .lk_moe_api import (MOE_BF16, MOE_FP16, MOE_FP8, MOE_FP8_FP16, MOE_MXFP4, MOE_MXFP4_FP16,
                         MOE_NVFP4, MOE_NVFP4_FP16, MOE_WNA16, MOE_WNA16_FP16, MOEConfigV2)

_DT = {torch.float32: _clib.DT_F32, torch.bfloat16: _clib.DT_BF16, torch.float16: _clib.DT_F16}
_SCORING = {"softmax": 0, "sigmoid": 1}


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream or None)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(None if t is None else t.data_ptr())


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("lvllm_amd ops run on the GPU only (got a CPU tensor); there is no CPU path")


def topk_softmax(gating_output: torch.Tensor, topk: int, renormalize: bool,
                 e_score_correction_bias: torch.Tensor | None = None,
                 scoring_func: str = "softmax", routed_scaling_factor: float = 1.0):
    """-> (topk_weights fp32 [M,K], topk_ids int32 [M,K])"""
    _need_cuda(gating_output, e_score_correction_bias)
    if scoring_func not in _SCORING:
        raise ValueError(f"Unsupported scoring function: {scoring_func}")
    if gating_output.dtype not in _DT:
        raise ValueError(f"unsupported gating dtype {gating_output.dtype}")
    g = gating_output.contiguous()
    M, E = g.shape
    bias = None
    if e_score_correction_bias is not None:
        bias = e_score_correction_bias.to(torch.float32).contiguous()
    w = torch.empty((M, topk), dtype=torch.float32, device=g.device)
    ids = torch.empty((M, topk), dtype=torch.int32, device=g.device)
    _clib.check(_clib.lib().lkm_topk_softmax(_stream(g), _ptr(g), _DT[g.dtype], _ptr(bias), M, E, topk,
                                             _SCORING[scoring_func], int(renormalize),
                                             float(routed_scaling_factor), _ptr(w), _ptr(ids)))
    return w, ids


def fused_topk(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int, renormalize: bool,
               indices_type: torch.dtype | None = None, scoring_func: str = "softmax"):
    """fused_topk_router.py:81-125 -> (topk_weights, topk_ids, token_expert_indices)"""
    assert hidden_states.size(0) == gating_output.size(0), "Number of tokens mismatch"
    w, ids = topk_softmax(gating_output, topk, renormalize, None, scoring_func)
    M = gating_output.size(0)
    # source_rows[m,k] = k*M + m  (topk_softmax_kernels.cu:568)
    dev = gating_output.device
    tei = (torch.arange(topk, device=dev, dtype=torch.int32)[None, :] * M
           + torch.arange(M, device=dev, dtype=torch.int32)[:, None])
    if indices_type is not None and indices_type != torch.int32:
        ids = ids.to(indices_type)
    return w, ids, tei


def fused_topk_bias(hidden_states: torch.Tensor, gating_output: torch.Tensor,
                    e_score_correction_bias: torch.Tensor, topk: int, renormalize: bool,
                    scoring_func: str = "softmax", routed_scaling_factor: float = 1.0):
    assert hidden_states.size(0) == gating_output.size(0), "Number of tokens mismatch"
    return topk_softmax(gating_output, topk, renormalize, e_score_correction_bias, scoring_func,
                        routed_scaling_factor)


def grouped_topk(hidden_states: torch.Tensor, gating_output: torch.Tensor, topk: int,
                 renormalize: bool, num_expert_group: int = 0, topk_group: int = 0,
                 scoring_func: str = "softmax", routed_scaling_factor: float = 1.0,
                 e_score_correction_bias: torch.Tensor | None = None):
    """grouped_topk_router.py:80-161 -> (topk_weights fp32, topk_ids int32)."""
    assert hidden_states.size(0) == gating_output.size(0), "Number of tokens mismatch"
    _need_cuda(gating_output, e_score_correction_bias)
    if scoring_func not in _SCORING:
        raise ValueError(f"Unsupported scoring function: {scoring_func}")
    g = gating_output.contiguous()
    M, E = g.shape
    bias = None
    if e_score_correction_bias is not None:
        bias = e_score_correction_bias.to(torch.float32).contiguous()
    w = torch.empty((M, topk), dtype=torch.float32, device=g.device)
    ids = torch.empty((M, topk), dtype=torch.int32, device=g.device)
    _clib.check(_clib.lib().lkm_grouped_topk(_stream(g), _ptr(g), _DT[g.dtype], _ptr(bias), M, E, topk,
                                             num_expert_group, topk_group, _SCORING[scoring_func],
                                             int(renormalize), float(routed_scaling_factor),
                                             _ptr(w), _ptr(ids)))
    return w, ids


_router_ws: dict = {}


def router_topk(hidden_states: torch.Tensor, gate_weight: torch.Tensor, topk: int, renormalize: bool, *,
                gate_bias: torch.Tensor | None = None, scoring_func: str = "softmax",
                num_expert_group: int = 0, topk_group: int = 0, routed_scaling_factor: float = 1.0,
                e_score_correction_bias: torch.Tensor | None = None,
                logits_dtype: torch.dtype = torch.float32, return_logits: bool = False):
    """Gate projection + routing in one call: what moe_runner.py:903-908 + fused_topk /
    grouped_topk do in two operators.  gate_weight [E,H] in the activation dtype or fp32 (GateLinear
    force_fp32_compute); logits_dtype fp32 (router GEMMs with fp32 output) or the activation dtype
    (F.linear in the gate's dtype).  -> (topk_weights fp32, topk_ids int32[, router_logits fp32])."""
    _need_cuda(hidden_states, gate_weight, gate_bias, e_score_correction_bias)
    if scoring_func not in _SCORING:
        raise ValueError(f"Unsupported scoring function: {scoring_func}")
    x, gw = hidden_states.contiguous(), gate_weight.contiguous()
    M, H = x.shape
    E = gw.shape[0]
    assert gw.shape[1] == H, "gate weight / hidden size mismatch"
    dev = x.device
    gb = None if gate_bias is None else gate_bias.to(torch.float32).contiguous()
    sb = None if e_score_correction_bias is None else e_score_correction_bias.to(torch.float32).contiguous()
    need = int(_clib.lib().lkm_router_workspace_bytes(M, H, E))
    ws = _router_ws.get(dev)
    if ws is None or ws.numel() < need:           # grow-only per-device scratch (split-K partials)
        ws = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=dev)
        _router_ws[dev] = ws
    w = torch.empty((M, topk), dtype=torch.float32, device=dev)
    ids = torch.empty((M, topk), dtype=torch.int32, device=dev)
    logits = torch.empty((M, E), dtype=torch.float32, device=dev) if return_logits else None
    _clib.check(_clib.lib().lkm_router_gemm_topk(
        _stream(x), _ptr(x), _DT[x.dtype], _ptr(gw), _DT[gw.dtype], _ptr(gb), _ptr(sb), M, H, E, topk,
        _SCORING[scoring_func], int(renormalize), float(routed_scaling_factor), num_expert_group, topk_group,
        _DT[logits_dtype], _ptr(ws), ws.numel(), _ptr(logits), _ptr(w), _ptr(ids)))
    return (w, ids, logits) if return_logits else (w, ids)


def determine_expert_map(ep_size: int, ep_rank: int, global_num_experts: int,
                         expert_placement_strategy: str = "linear", num_fused_shared_experts: int = 0):
    """expert_map_manager.py:22-113 -> (local_num_experts, expert_map int32 | None).  Host logic.
    With num_fused_shared_experts = n > 0 (shared experts folded into the engine behind this rank's routed
    experts, lvllm_amd/shared_experts.py) the map is the reference's extended map [E + n] (ids E+i ->
    local_num_experts + i, :100-110) plus ONE more entry -1 for the sentinel id E + n: the reference masks
    that id with a separate expert_mask whose last entry is 0 (:94-99), here the id map itself drops it."""
    assert ep_size > 0
    if ep_size == 1:
        return global_num_experts, None
    base, rem = divmod(global_num_experts, ep_size)
    local = base + 1 if ep_rank < rem else base
    emap = torch.full((global_num_experts,), -1, dtype=torch.int32)
    if expert_placement_strategy == "linear":
        start = ep_rank * base + min(ep_rank, rem)
        emap[start:start + local] = torch.arange(local, dtype=torch.int32)
    elif expert_placement_strategy == "round_robin":
        emap[torch.arange(ep_rank, global_num_experts, ep_size)] = torch.arange(local, dtype=torch.int32)
    else:
        raise ValueError(f"Unsupported expert placement strategy '{expert_placement_strategy}', "
                         "expected one of ('linear', 'round_robin')")
    if num_fused_shared_experts > 0:
        tail = [local + i for i in range(num_fused_shared_experts)] + [-1]
        emap = torch.cat((emap, torch.tensor(tail, dtype=torch.int32)))
    return local, emap


def global_to_local_expert_ids(topk_ids: torch.Tensor, expert_map: torch.Tensor) -> torch.Tensor:
    """routed_experts.py:1332-1342: non-local / negative ids -> -1."""
    _need_cuda(topk_ids)
    ids = topk_ids.to(torch.int32).contiguous()
    emap = expert_map.to(device=ids.device, dtype=torch.int32).contiguous()
    out = torch.empty_like(ids)
    _clib.check(_clib.lib().lkm_map_expert_ids(_stream(ids), _ptr(ids), ids.numel(), _ptr(emap),
                                               emap.numel(), _ptr(out)))
    return out


def eplb_map_to_physical_and_record(topk_ids: torch.Tensor, expert_load_view: torch.Tensor | None,
                                    logical_to_physical_map: torch.Tensor, logical_replica_count: torch.Tensor,
                                    record_enabled: torch.Tensor | None = None,
                                    num_unpadded_tokens: torch.Tensor | None = None) -> torch.Tensor:
    """Logical -> physical expert ids (replica picked by a hash of the token index) and, when recording
    is on, expert_load_view[physical] += 1 per routed slot of an unpadded token; base_router.py:129-146
    (same argument names and order).  Maps / counters / switches are used as they are when they already are
    contiguous int32 tensors on the device (what lvllm_amd.eplb.EplbState keeps; required for the load view,
    which is updated in place, and for switches that must stay live inside a captured graph); other integer
    dtypes (the reference keeps int64 maps and a bool switch) are converted per call."""
    _need_cuda(topk_ids, expert_load_view, logical_to_physical_map, logical_replica_count, record_enabled,
               num_unpadded_tokens)
    if topk_ids.numel() == 0:
        return topk_ids
    if topk_ids.dim() != 2:
        raise ValueError("topk_ids must be [tokens, top_k]")
    dev = topk_ids.device
    ids = topk_ids.to(torch.int32).contiguous()

    def i32(t):
        return None if t is None else t.to(device=dev, dtype=torch.int32).contiguous()
    l2p, cnt = i32(logical_to_physical_map), i32(logical_replica_count)
    if l2p.dim() != 2 or cnt.dim() != 1 or l2p.size(0) != cnt.size(0):
        raise ValueError("expected logical_to_physical_map [logical, slots] and logical_replica_count [logical]")
    if expert_load_view is not None and (expert_load_view.dtype != torch.int32 or not expert_load_view.is_contiguous()):
        raise ValueError("expert_load_view must be a contiguous int32 tensor (it is updated in place)")
    rec, unpadded = i32(record_enabled), i32(num_unpadded_tokens)     # converted copies stay referenced until the launch
    out = torch.empty_like(ids)
    _clib.check(_clib.lib().lkm_eplb_map_record(
        _stream(ids), _ptr(ids), ids.numel(), ids.size(1), _ptr(l2p), _ptr(cnt), l2p.size(0), l2p.size(1),
        _ptr(expert_load_view), 0 if expert_load_view is None else expert_load_view.numel(),
        _ptr(rec), _ptr(unpadded), _ptr(out)))
    return out.to(topk_ids.dtype)


def ep_row_bytes(hidden_size: int, top_k: int) -> int:
    """bytes of one exchanged token record [H x 16-bit | K x int32 ids | K x fp32 weights], padded to 16"""
    return int(_clib.lib().lkm_ep_row_bytes(hidden_size, top_k))


def ep_pack_tokens(hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor, num_experts: int,
                   ep_size: int, capacity: int, send: torch.Tensor, slot_of: torch.Tensor, overflow: torch.Tensor,
                   global_ids: bool = False) -> None:
    """Token-granular fixed-capacity EP dispatch pack into caller-owned buffers (persistent: a captured graph
    replays on the same addresses): send uint8 [ep, capacity, ep_row_bytes(H, K)], slot_of int32 [ep, M],
    overflow int32 [1] (+= tokens dropped for lack of capacity); see lkm_ep_pack_tokens in include/lkm.h."""
    _need_cuda(hidden, topk_weights, topk_ids, send, slot_of, overflow)
    M, K = topk_ids.shape
    H = hidden.size(1)
    assert hidden.is_contiguous() and topk_ids.dtype == torch.int32 and topk_weights.dtype == torch.float32
    assert topk_ids.is_contiguous() and topk_weights.is_contiguous()
    assert send.dtype == torch.uint8 and send.is_contiguous() and send.numel() == ep_size * capacity * ep_row_bytes(H, K)
    assert slot_of.dtype == torch.int32 and slot_of.is_contiguous() and slot_of.numel() == ep_size * M
    assert overflow.dtype == torch.int32 and overflow.numel() == 1
    _clib.check(_clib.lib().lkm_ep_pack_tokens(_stream(hidden), _ptr(hidden), _ptr(topk_ids), _ptr(topk_weights), M, K,
                                               H, num_experts, ep_size, capacity, int(global_ids), _ptr(send),
                                               _ptr(slot_of), _ptr(overflow)))


def ep_combine(back: torch.Tensor, slot_of: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """back [ep, capacity, H] (fp32 or 16-bit), slot_of int32 [ep, M] -> out [M, H] (fp32 or 16-bit):
    out[m] = sum_p back[p, slot_of[p, m]] in fp32, p ascending; see lkm_ep_combine."""
    _need_cuda(back, slot_of, out)
    ep, cap, H = back.shape
    M = out.size(0)
    assert back.is_contiguous() and out.is_contiguous() and slot_of.is_contiguous()
    assert slot_of.dtype == torch.int32 and slot_of.numel() == ep * M and out.size(1) == H
    _clib.check(_clib.lib().lkm_ep_combine(_stream(back), _ptr(back), _DT[back.dtype], _ptr(slot_of), M, H, ep, cap,
                                           _ptr(out), _DT[out.dtype]))
    return out


def per_token_group_quant_fp8(x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """x [rows, cols] bf16 / fp16 (row stride a multiple of 8) -> (q uint8 [rows, cols] e4m3fn bytes, scales fp32
    [rows, ceil(cols / 128)]): per_token_group_quant_fp8 with groups of 128 (fp8_utils.py:533-660), the quantiser the
    W8A8 engines run on their inputs."""
    _need_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype in (torch.bfloat16, torch.float16)
    rows, cols = x.shape
    q = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    sc = torch.empty((rows, -(-cols // 128)), dtype=torch.float32, device=x.device)
    _clib.check(_clib.lib().lkm_per_token_group_quant_fp8(_stream(x), _ptr(x), _DT[x.dtype], x.stride(0), rows, cols,
                                                          _ptr(q), _ptr(sc)))
    return q, sc


def wna16_expand(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor | None, weight_bits: int,
                 group: int) -> torch.Tensor:
    """Weight-only integer experts -> 16-bit weights T((q - zp) * s), the in-tree operator's dequantisation
    (fused_moe.py:207-276; layouts of tests/kernels/moe/test_moe.py:565-693): qweight uint8 [E, N, K/2] (4-bit) or
    [E, N, K]; scales [E, N, K/group] bf16 / fp16; zeros uint8 [E, N/2, K/group] (4-bit) / [E, N, K/group] (8-bit) or
    None (symmetric: 8 / 128).  -> [E, N, K] in the scales' dtype."""
    _need_cuda(qweight)
    assert qweight.dtype == torch.uint8 and qweight.dim() == 3 and qweight.is_contiguous()
    assert scales.dtype in (torch.bfloat16, torch.float16) and scales.is_contiguous() and scales.is_cuda
    E, N = qweight.shape[:2]
    K = qweight.size(2) * (2 if weight_bits == 4 else 1)
    if tuple(scales.shape) != (E, N, K // group):
        raise ValueError(f"wna16 scales {tuple(scales.shape)}: expected {(E, N, K // group)} for group {group}")
    if zeros is not None:
        want = (E, N // 2, K // group) if weight_bits == 4 else (E, N, K // group)
        if zeros.dtype != torch.uint8 or tuple(zeros.shape) != want or not zeros.is_cuda:
            raise ValueError(f"wna16 zero points {tuple(zeros.shape)} {zeros.dtype}: expected uint8 {want} on the GPU")
        zeros = zeros.contiguous()
    out = torch.empty((E, N, K), dtype=scales.dtype, device=qweight.device)
    _clib.check(_clib.lib().lkm_wna16_expand(_stream(qweight), _ptr(qweight), _ptr(scales),
                                             _ptr(zeros) if zeros is not None else None, _ptr(out), E * N, K, group,
                                             weight_bits, _DT[scales.dtype]))
    return out


def sort_slots(topk_ids: torch.Tensor, num_experts: int):
    """Stable counting sort of the M*K slots by expert.
    -> counts [E], offsets [E+1], sorted_slot [M*K] (tail -1), pos_of_slot [M*K] (-1 = skipped)."""
    _need_cuda(topk_ids)
    ids = topk_ids.to(torch.int32).contiguous().view(-1)
    n = ids.numel()
    dev = ids.device
    counts = torch.empty(num_experts, dtype=torch.int32, device=dev)
    offsets = torch.empty(num_experts + 1, dtype=torch.int32, device=dev)
    sorted_slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    pos = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    _clib.check(_clib.lib().lkm_sort_slots(_stream(ids), _ptr(ids), n, num_experts, _ptr(counts),
                                           _ptr(offsets), _ptr(sorted_slot), _ptr(pos)))
    return counts, offsets, sorted_slot[:n], pos[:n]


def _ops_workspace(n_slots: int, n_keys: int, dev) -> torch.Tensor:
    nbytes = int(_clib.lib().lkm_moe_ops_workspace_bytes(int(n_slots), int(n_keys)))
    return torch.empty((max(nbytes, 16) + 3) // 4, dtype=torch.int32, device=dev)


def moe_align_block_size(topk_ids: torch.Tensor, block_size: int, num_experts: int,
                         expert_map: torch.Tensor | None = None, pad_sorted_ids: bool = False,
                         ignore_invalid_experts: bool = False):
    """moe_align_block_size.py:11-103, same arguments and return values: (sorted_token_ids int32 [max_num_tokens_padded],
    expert_ids int32 [max_num_m_blocks], num_tokens_post_padded int32 [1]).  Unused sorted ids = topk_ids.numel(), unused
    blocks = -1, rows of an expert in token order (what the reference's golden implementation produces,
    tests/kernels/moe/test_moe_align_block_size.py:96-172)."""
    _need_cuda(topk_ids, expert_map)
    ids = topk_ids.to(torch.int32).contiguous()
    n = ids.numel()
    max_padded = n + num_experts * (block_size - 1)
    if pad_sorted_ids:
        max_padded = -(-max_padded // block_size) * block_size
    if n < num_experts:
        max_padded = min(n * block_size, max_padded)
    dev = ids.device
    sorted_ids = torch.empty((max_padded,), dtype=torch.int32, device=dev)
    max_blocks = -(-max_padded // block_size)
    expert_ids = torch.empty((max_blocks,), dtype=torch.int32, device=dev)
    post = torch.empty((1,), dtype=torch.int32, device=dev)
    emap = None
    if expert_map is not None and ignore_invalid_experts:
        emap = expert_map.to(torch.int32).contiguous()
    if n == 0:                      # (no slots: no blocks; the output tensors are empty, there is nothing to launch)
        expert_ids.fill_(-1)
        sorted_ids.fill_(0)
        return sorted_ids, expert_ids, post.zero_()
    ws = _ops_workspace(n, num_experts, dev)
    _clib.check(_clib.lib().lkm_moe_align_block_size(_stream(ids), _ptr(ids), n, num_experts, block_size, _ptr(emap),
                                                     _ptr(sorted_ids), max_padded, _ptr(expert_ids), max_blocks, _ptr(post),
                                                     _ptr(ws)))
    if expert_map is not None and not ignore_invalid_experts:
        expert_ids = expert_map[expert_ids.to(torch.int64)]          # (the reference's own post-step, :99-100)
    return sorted_ids, expert_ids, post


def moe_permute(hidden_states: torch.Tensor, a1q_scale: torch.Tensor | None, topk_ids: torch.Tensor, n_expert: int,
                n_local_expert: int = -1, expert_map: torch.Tensor | None = None,
                permuted_hidden_states: torch.Tensor | None = None):
    """moe_permute_unpermute.py:105-242 -> (permuted_hidden_states, a1q_scale, expert_first_token_offset int64
    [n_local_expert + 1], inv_permuted_idx int32 [n_token * topk], permuted_idx int32 [n_token * topk]).  The rows of experts
    that are not local are not written (the reference's test compares the valid rows only)."""
    _need_cuda(hidden_states, topk_ids, expert_map)
    n_token, n_hidden = hidden_states.size()
    topk = topk_ids.size(1)
    assert (n_hidden * hidden_states.element_size()) % 16 == 0, "permue kernel need hidden dim align to 16B"
    rows = n_token * topk
    if n_local_expert == -1:
        n_local_expert = n_expert
    dev = hidden_states.device
    if permuted_hidden_states is None:
        permuted_hidden_states = torch.empty((rows, n_hidden), dtype=hidden_states.dtype, device=dev)
    assert permuted_hidden_states.size() == (rows, n_hidden), (
        f"Expected permuted hidden states to be {(rows, n_hidden)} but got {permuted_hidden_states.size()}")
    assert permuted_hidden_states.is_contiguous()
    h = hidden_states.contiguous()
    ids = topk_ids.to(torch.int32).contiguous()
    emap = None if expert_map is None else expert_map.to(torch.int32).contiguous()
    first = torch.empty((n_local_expert + 1,), dtype=torch.int64, device=dev)
    inv = torch.empty((n_token, topk), dtype=torch.int32, device=dev)
    perm = torch.empty((rows,), dtype=torch.int32, device=dev)
    if rows == 0:
        return permuted_hidden_states, a1q_scale, first.zero_(), inv.flatten(), perm
    ws = _ops_workspace(rows, n_expert if emap is None else n_local_expert + n_expert, dev)
    _clib.check(_clib.lib().lkm_moe_permute(_stream(h), _ptr(h), n_hidden * h.element_size(), n_token, _ptr(ids), topk,
                                            _ptr(emap), n_expert, n_local_expert, _ptr(permuted_hidden_states), _ptr(first),
                                            _ptr(inv), _ptr(perm), _ptr(ws)))
    if a1q_scale is not None and a1q_scale.dim() > 1:
        a1q_scale = a1q_scale[perm.clamp(max=rows - 1).to(torch.int64) // topk]
    return permuted_hidden_states, a1q_scale, first, inv.flatten(), perm


def moe_unpermute(out: torch.Tensor, permuted_hidden_states: torch.Tensor, topk_weights: torch.Tensor,
                  inv_permuted_idx: torch.Tensor, expert_first_token_offset: torch.Tensor | None = None) -> None:
    """moe_permute_unpermute.py:245-283: out[t] = sum_k topk_weights[t][k] * permuted_hidden_states[inv_permuted_idx[t][k]]
    over the valid rows (fp32 sum, one rounding to the rows' dtype)."""
    _need_cuda(out, permuted_hidden_states, topk_weights, inv_permuted_idx, expert_first_token_offset)
    topk = topk_weights.size(1)
    n_hidden = permuted_hidden_states.size(-1)
    assert (n_hidden * permuted_hidden_states.element_size()) % 16 == 0, "unpermue kernel need hidden dim align to 16B"
    assert permuted_hidden_states.dtype in (torch.bfloat16, torch.float16) and out.dtype == permuted_hidden_states.dtype
    assert out.is_contiguous() and permuted_hidden_states.is_contiguous()
    tw = topk_weights.to(torch.float32).contiguous()
    inv = inv_permuted_idx.to(torch.int32).contiguous()
    first = None if expert_first_token_offset is None else expert_first_token_offset.to(torch.int64).contiguous()
    _clib.check(_clib.lib().lkm_moe_unpermute(_stream(out), _ptr(permuted_hidden_states), _DT[out.dtype], _ptr(tw), _ptr(inv),
                                              _ptr(first), 0 if first is None else first.numel() - 1, out.size(0), topk,
                                              n_hidden, _ptr(out)))


def moe_permute_unpermute_supported() -> bool:
    return True


_CLS = {("bf16", torch.bfloat16): MOE_BF16, ("bf16", torch.float16): MOE_FP16,
        ("fp16", torch.float16): MOE_FP16,
        ("fp8", torch.bfloat16): MOE_FP8, ("fp8", torch.float16): MOE_FP8_FP16,
        ("int4", torch.bfloat16): MOE_WNA16, ("int4", torch.float16): MOE_WNA16_FP16,
        ("mxfp4", torch.bfloat16): MOE_MXFP4, ("mxfp4", torch.float16): MOE_MXFP4_FP16,
        ("nvfp4", torch.bfloat16): MOE_NVFP4, ("nvfp4", torch.float16): MOE_NVFP4_FP16}


class RoutedExpertsEngine:
    """Torch-level convenience over the lk_moe classes: builds the MOEConfigV2 from tensors exactly
    the way RoutedExperts._process_{bf6_fp16,wna16,fp8} do (routed_experts.py:1440-1668) and exposes
    the three forward paths on tensors (cf. _cpu_decode/_cpu_prefill/_gpu_prefill, :1840-1899)."""

    def __init__(self, w13: torch.Tensor, w2: torch.Tensor, *, top_k: int, act_dtype: torch.dtype,
                 fmt: str = "bf16", w13_scale: torch.Tensor | None = None,
                 w2_scale: torch.Tensor | None = None, group_n: int = 0, group_k: int = 0,
                 has_gate_proj: bool = True, activation_type: int = 0, swiglu_alpha: float = 1.702,
                 swiglu_limit: float = 7.0, max_num_seqs: int = 256, max_batch_size: int = 8192,
                 group_max_len: int = 0, num_processes: int = 1, process_id: int = 0,
                 gpu_id: int | None = None, fp8_mode: int = _clib.FP8_W8A16, int4_mode: int = _clib.INT4_EXACT,
                 w13_global_scale: torch.Tensor | None = None,
                 w2_global_scale: torch.Tensor | None = None,
                 w13_zp: torch.Tensor | None = None, w2_zp: torch.Tensor | None = None):
        """w13_zp / w2_zp (fmt "int4" only): the zero points of asymmetric uint4 experts, uint8 [E, rows, K / group_k], one byte
        (0..15) per weight row and scale group -> LkmConfig.int4_mode = LKM_INT4_ZP, weights dequantised to T((q - zp) * s)
        (fused_moe.py:272-276); they travel in the global-scale pointer slots of lkm_create (include/lkm.h)."""
        E = w13.shape[0]
        H = w2.shape[1]
        inter = w13.shape[1] // (2 if has_gate_proj else 1)
        cfg = MOEConfigV2()
        cfg.num_processes, cfg.process_id = num_processes, process_id
        cfg.gpu_id = torch.cuda.current_device() if gpu_id is None else gpu_id
        cfg.has_gate_proj = has_gate_proj
        cfg.expert_num, cfg.top_k = E, top_k
        cfg.hidden_size, cfg.intermediate_size = H, inter
        cfg.max_batch_size, cfg.max_num_seqs = max_batch_size, max_num_seqs
        cfg.group_max_len = group_max_len or (min(4096, max_batch_size) + 128)
        cfg.groupN, cfg.groupK = group_n, group_k
        cfg.activation_type = activation_type
        cfg.swiglu_alpha, cfg.swiglu_limit = swiglu_alpha, swiglu_limit
        cfg.fp8_mode = fp8_mode
        if (w13_zp is None) != (w2_zp is None):
            raise ValueError("zero points: w13_zp and w2_zp come together")
        if w13_zp is not None:
            if fmt != "int4" or int4_mode not in (_clib.INT4_EXACT, _clib.INT4_ZP):
                raise ValueError("zero points belong to fmt='int4' in the exact mode")
            if w13_zp.dtype != torch.uint8 or w2_zp.dtype != torch.uint8 or tuple(w13_zp.shape) != tuple(w13_scale.shape) \
                    or tuple(w2_zp.shape) != tuple(w2_scale.shape):
                raise ValueError("zero points: uint8 tensors shaped like the scales ([E, rows, K / group])")
            int4_mode = _clib.INT4_ZP
        elif int4_mode == _clib.INT4_ZP:
            raise ValueError("int4_mode ZP without zero points")
        cfg.int4_mode = int4_mode
        self.cfg = cfg
        self.H, self.K, self.act_dtype = H, top_k, act_dtype
        cls = _CLS[(fmt, act_dtype)]
        w13c, w2c = w13.contiguous(), w2.contiguous()
        s13 = None if w13_scale is None else w13_scale.contiguous()
        s2 = None if w2_scale is None else w2_scale.contiguous()
        g13 = None if w13_global_scale is None else w13_global_scale.to(torch.float32).contiguous()
        g2 = None if w2_global_scale is None else w2_global_scale.to(torch.float32).contiguous()
        if w13_zp is not None:
            g13, g2 = w13_zp.contiguous(), w2_zp.contiguous()
        self.engine = cls(cfg, w13c.data_ptr(), w2c.data_ptr(), 0 if s13 is None else s13.data_ptr(),
                          0 if s2 is None else s2.data_ptr(), 0 if g13 is None else g13.data_ptr(),
                          0 if g2 is None else g2.data_ptr())
        if w13c.is_cuda:
            torch.cuda.synchronize()

    def decode(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
               out: torch.Tensor | None = None) -> torch.Tensor:
        """fp32 [M,H]   (cf. RoutedExperts._cpu_decode)"""
        _need_cuda(hidden, topk_weights, topk_ids)
        M = hidden.size(0)
        if out is None:
            out = torch.empty((M, self.H), dtype=torch.float32, device=hidden.device)
        assert hidden.dtype == self.act_dtype and hidden.is_contiguous()
        assert topk_ids.dtype == torch.int32 and topk_weights.dtype == torch.float32
        assert topk_ids.is_contiguous() and topk_weights.is_contiguous()
        self.engine.cpu_decode(torch.cuda.current_stream(hidden.device).cuda_stream, M, topk_ids.size(1),
                               hidden.data_ptr(), topk_ids.data_ptr(), topk_weights.data_ptr(),
                               out.data_ptr())
        return out

    def prefill(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        """activation dtype [M,H]   (cf. RoutedExperts._gpu_prefill)"""
        _need_cuda(hidden, topk_weights, topk_ids)
        assert hidden.dtype == self.act_dtype and hidden.is_contiguous()
        assert topk_ids.dtype == torch.int32 and topk_weights.dtype == torch.float32
        out = torch.empty_like(hidden)
        self.engine.gpu_prefill(hidden.data_ptr(), out.data_ptr(), topk_ids.data_ptr(),
                                topk_weights.data_ptr(), hidden.size(0), topk_ids.size(1),
                                torch.cuda.current_stream(hidden.device).cuda_stream)
        return out

    def forward_rows(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                     out: torch.Tensor | None = None, out_dtype: torch.dtype = torch.float32,
                     id_offset: int = 0, valid_den: int | None = None) -> torch.Tensor:
        """The operator on ROW-STRIDED views (last dimension contiguous), as the expert-parallel exchange hands
        them over: hidden [R,H] act dtype, ids int32 [R,K], weights fp32 [R,K]; `id_offset` is subtracted from
        ids >= 0 (global -> local under linear placement).  valid_den: the caller's records are sparse (expert-parallel
        exchange: ep x capacity slots, ~1/valid_den of them local).  -> [R,H] contiguous, fp32 or the activation dtype."""
        _need_cuda(hidden, topk_weights, topk_ids)
        R, K = topk_ids.shape
        assert hidden.dtype == self.act_dtype and topk_ids.dtype == torch.int32 and topk_weights.dtype == torch.float32
        assert hidden.size(0) == R and topk_weights.shape == topk_ids.shape
        for t_ in (hidden, topk_ids, topk_weights):
            assert t_.dim() == 2 and (t_.size(1) == 1 or t_.stride(1) == 1), "rows must be contiguous in their last dimension"
        if out is None:
            out = torch.empty((R, self.H), dtype=out_dtype, device=hidden.device)
        assert out.is_contiguous() and out.dtype in (torch.float32, self.act_dtype)
        if R == 0:
            return out
        vd = int(valid_den or 0)
        if vd != getattr(self, "_valid_den", 0):
            # expert-parallel records: ~1 / valid_den of the R x K slots carry a local id -- the launch plan is made for
            # the rows that exist (host-side hint, lkm_set_tuning "valid_den"); a call without the hint plans densely again
            self.engine.set_tuning(valid_den=vd)
            self._valid_den = vd
        self.engine.forward_strided(torch.cuda.current_stream(hidden.device).cuda_stream, R, K, hidden.data_ptr(),
                                    hidden.stride(0) if R > 1 else max(hidden.stride(0), self.H), topk_ids.data_ptr(),
                                    topk_ids.stride(0) if R > 1 else max(topk_ids.stride(0), K), int(id_offset),
                                    topk_weights.data_ptr(),
                                    topk_weights.stride(0) if R > 1 else max(topk_weights.stride(0), K),
                                    out.data_ptr(), _DT[out.dtype])
        return out

    def forward_logits(self, hidden: torch.Tensor, router_logits: torch.Tensor, topk: int, renormalize: bool, *,
                       scoring_func: str = "softmax", num_expert_group: int = 0, topk_group: int = 0,
                       routed_scaling_factor: float = 1.0, e_score_correction_bias: torch.Tensor | None = None,
                       out: torch.Tensor | None = None, out_dtype: torch.dtype = torch.float32,
                       id_offset: int = 0):
        """Routing + experts in one call (include/lkm.h lkm_forward_routed): topk_softmax / grouped_topk on
        `router_logits` [M, E_router], then the operator on the result; for decode batches the router and the
        scatter metadata are one launch.  Bit-identical to topk_softmax()/grouped_topk() followed by
        forward_rows().  -> (out [M,H], topk_weights fp32 [M,K], topk_ids int32 [M,K])"""
        _need_cuda(hidden, router_logits, e_score_correction_bias)
        if scoring_func not in _SCORING:
            raise ValueError(f"Unsupported scoring function: {scoring_func}")
        if router_logits.dtype not in _DT:
            raise ValueError(f"unsupported gating dtype {router_logits.dtype}")
        assert hidden.dtype == self.act_dtype and hidden.dim() == 2 and (hidden.size(1) == 1 or hidden.stride(1) == 1)
        g = router_logits.contiguous()
        M, E = g.shape
        assert hidden.size(0) == M, "Number of tokens mismatch"
        bias = None
        if e_score_correction_bias is not None:
            bias = e_score_correction_bias.to(torch.float32).contiguous()
        w = torch.empty((M, topk), dtype=torch.float32, device=g.device)
        ids = torch.empty((M, topk), dtype=torch.int32, device=g.device)
        if out is None:
            out = torch.empty((M, self.H), dtype=out_dtype, device=hidden.device)
        assert out.is_contiguous() and out.dtype in (torch.float32, self.act_dtype)
        if M == 0:
            return out, w, ids
        self.engine.forward_routed(torch.cuda.current_stream(hidden.device).cuda_stream, M, topk, hidden.data_ptr(),
                                   hidden.stride(0) if M > 1 else max(hidden.stride(0), self.H), g.data_ptr(),
                                   _DT[g.dtype], E, 0 if bias is None else bias.data_ptr(), int(num_expert_group), int(topk_group),
                                   _SCORING[scoring_func], renormalize, float(routed_scaling_factor), int(id_offset),
                                   w.data_ptr(), ids.data_ptr(), out.data_ptr(), _DT[out.dtype])
        return out, w, ids

    def prefill_host(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
        """CPU tensors in, fp32 CPU tensor out   (cf. RoutedExperts._cpu_prefill)"""
        assert not hidden.is_cuda
        hidden = hidden.contiguous()
        ids = topk_ids.to(torch.int32).contiguous()
        tw = topk_weights.to(torch.float32).contiguous()
        out = torch.empty((hidden.size(0), self.H), dtype=torch.float32)
        self.engine.cpu_prefill(hidden.size(0), ids.size(1), ids.data_ptr(), tw.data_ptr(),
                                hidden.data_ptr(), out.data_ptr())
        return out
