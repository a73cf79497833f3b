"""ctypes front-end of the CPU oracle (oracle/lkm_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by lvllm_amd/ or lk_moe/ (tests/test_boundary.py checks).
Parity status: see the header of lkm_oracle.c ("lk_moe boundary unpinned"; in-tree operator
pinned by tests/golden/ and by oracle/_ref = the reference's own CPU kernel, see oracle/ref.py).

Arrays are numpy; bf16/fp16 tensors travel as uint16 bit patterns.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from dataclasses import dataclass
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liblkm_oracle.so"

F32, BF16, F16 = 0, 1, 2
W_BF16, W_F16, W_FP8, W_INT4, W_NVFP4, W_MXFP4 = 0, 1, 2, 3, 4, 5
ACT_SILU, ACT_SWIGLUOAI, ACT_RELU2 = 0, 1, 2


def build(force: bool = False) -> Path:
    src = HERE / "lkm_oracle.c"
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        r = subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s", "liblkm_oracle.so"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.lkm_or_expf.restype = C.c_float
        _lib.lkm_or_expf.argtypes = [C.c_float]
        _lib.lkm_or_expert_map.restype = C.c_int
        _lib.lkm_or_moe.restype = C.c_int
        _lib.lkm_or_num_threads.restype = C.c_int
        _lib.lkm_or_configure(C.c_int(0))
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt=None):
    a = np.ascontiguousarray(a)
    return a if dt is None or a.dtype == dt else a.astype(dt)


def dtype_code(a: np.ndarray, hint: int | None = None) -> int:
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint16:
        if hint is None:
            raise ValueError("uint16 array needs an explicit dtype hint (BF16 or F16)")
        return hint
    if a.dtype == np.float16:
        return F16
    raise ValueError(f"unsupported dtype {a.dtype}")


def expf(x: float) -> float:
    return float(lib().lkm_or_expf(C.c_float(x)))


def f32_to_bits(a: np.ndarray, dt: int) -> np.ndarray:
    a = _c(a, np.float32)
    out = np.empty(a.shape, np.uint16)
    lib().lkm_or_cvt_f32_to(_p(a), C.c_int(dt), C.c_int64(a.size), _p(out))
    return out


def bits_to_f32(a: np.ndarray, dt: int) -> np.ndarray:
    a = _c(a)
    out = np.empty(a.shape, np.float32)
    lib().lkm_or_cvt_to_f32(_p(a), C.c_int(dt), C.c_int64(a.size), _p(out))
    return out


def fp8_to_f32(a: np.ndarray) -> np.ndarray:
    a = _c(a, np.uint8)
    out = np.empty(a.shape, np.float32)
    lib().lkm_or_fp8_to_f32(_p(a), C.c_int64(a.size), _p(out))
    return out


def f32_to_fp8(a: np.ndarray) -> np.ndarray:
    a = _c(a, np.float32)
    out = np.empty(a.shape, np.uint8)
    lib().lkm_or_f32_to_fp8(_p(a), C.c_int64(a.size), _p(out))
    return out


def topk_softmax(logits: np.ndarray, K: int, *, dt: int | None = None, bias=None, scoring: int = 0,
                 renormalize: bool = True, routed_scaling: float = 1.0):
    logits = _c(logits)
    code = dtype_code(logits, dt)
    if logits.dtype == np.float16:
        logits = logits.view(np.uint16)
    M, E = logits.shape
    bias = None if bias is None else _c(bias, np.float32)
    w = np.empty((M, K), np.float32)
    ids = np.empty((M, K), np.int32)
    lib().lkm_or_topk_softmax(_p(logits), C.c_int(code), _p(bias), C.c_int(M), C.c_int(E), C.c_int(K),
                              C.c_int(scoring), C.c_int(int(renormalize)), C.c_float(routed_scaling),
                              _p(w), _p(ids))
    return w, ids


def grouped_topk(logits: np.ndarray, K: int, n_group: int, topk_group: int, *, dt: int | None = None,
                 bias=None, scoring: int = 1, renormalize: bool = True, routed_scaling: float = 1.0):
    logits = _c(logits)
    code = dtype_code(logits, dt)
    if logits.dtype == np.float16:
        logits = logits.view(np.uint16)
    M, E = logits.shape
    bias = None if bias is None else _c(bias, np.float32)
    w = np.empty((M, K), np.float32)
    ids = np.empty((M, K), np.int32)
    lib().lkm_or_grouped_topk(_p(logits), C.c_int(code), _p(bias), C.c_int(M), C.c_int(E), C.c_int(K),
                              C.c_int(n_group), C.c_int(topk_group), C.c_int(scoring),
                              C.c_int(int(renormalize)), C.c_float(routed_scaling), _p(w), _p(ids))
    return w, ids


def expert_map(ep_size: int, ep_rank: int, E: int, strategy: int = 0):
    m = np.empty(E, np.int32)
    n = lib().lkm_or_expert_map(C.c_int(ep_size), C.c_int(ep_rank), C.c_int(E), C.c_int(strategy), _p(m))
    return int(n), m


def map_ids(ids: np.ndarray, emap: np.ndarray) -> np.ndarray:
    ids = _c(ids, np.int32)
    emap = _c(emap, np.int32)
    out = np.empty(ids.shape, np.int32)
    lib().lkm_or_map_ids(_p(ids), C.c_int64(ids.size), _p(emap), C.c_int(emap.size), _p(out))
    return out


def sort_slots(ids: np.ndarray, E: int):
    flat = _c(ids, np.int32).reshape(-1)
    n = flat.size
    counts = np.empty(E, np.int32)
    offsets = np.empty(E + 1, np.int32)
    sorted_slot = np.empty(max(n, 1), np.int32)
    pos = np.empty(max(n, 1), np.int32)
    lib().lkm_or_sort(_p(flat), C.c_int(n), C.c_int(E), _p(counts), _p(offsets), _p(sorted_slot), _p(pos))
    return counts, offsets, sorted_slot[:n], pos[:n]


class _Desc(C.Structure):
    _fields_ = [("E", C.c_int32), ("H", C.c_int32), ("I", C.c_int32), ("has_gate", C.c_int32),
                ("activation", C.c_int32), ("swiglu_alpha", C.c_float), ("swiglu_limit", C.c_float),
                ("act_dtype", C.c_int32), ("wfmt", C.c_int32), ("groupN", C.c_int32),
                ("groupK", C.c_int32), ("round_gemm1", C.c_int32), ("w8a8", C.c_int32),
                ("gs13", C.c_void_p), ("gs2", C.c_void_p), ("int4_unrounded", C.c_int32)]


@dataclass
class MoeDesc:
    E: int
    H: int
    I: int
    has_gate: bool = True
    activation: int = ACT_SILU
    swiglu_alpha: float = 1.702
    swiglu_limit: float = 7.0
    act_dtype: int = BF16
    wfmt: int = W_BF16
    groupN: int = 0
    groupK: int = 0
    round_gemm1: bool = False
    w8a8: bool = False
    int4_unrounded: bool = False       # candidate int4 mode (scale on fp32 partial sums); not the reference's semantics


def moe(d: MoeDesc, w13, w2, x, ids, tw, s13=None, s2=None, gs13=None, gs2=None) -> np.ndarray:
    """Routed experts; returns fp32 [M,H].  Arrays: see lkm_or_moe.  gs13/gs2: NVFP4 per-expert
    f32 multipliers [E]."""
    x = _c(x)
    ids = _c(ids, np.int32)
    tw = _c(tw, np.float32)
    M, K = ids.shape
    assert x.shape == (M, d.H), (x.shape, M, d.H)
    w13, w2 = _c(w13), _c(w2)
    s13 = None if s13 is None else _c(s13)
    s2 = None if s2 is None else _c(s2)
    gs13 = None if gs13 is None else _c(gs13, np.float32)
    gs2 = None if gs2 is None else _c(gs2, np.float32)
    cd = _Desc(d.E, d.H, d.I, int(d.has_gate), d.activation, d.swiglu_alpha, d.swiglu_limit,
               d.act_dtype, d.wfmt, d.groupN, d.groupK, int(d.round_gemm1), int(d.w8a8),
               None if gs13 is None else gs13.ctypes.data, None if gs2 is None else gs2.ctypes.data,
               int(d.int4_unrounded))
    out = np.empty((M, d.H), np.float32)
    rc = lib().lkm_or_moe(C.byref(cd), _p(w13), _p(w2), _p(s13), _p(s2), _p(x), _p(ids), _p(tw),
                          C.c_int(M), C.c_int(K), _p(out))
    assert rc == 0
    return out


def quant_int4(w_bits: np.ndarray, dt: int, g: int):
    """w_bits: uint16 [..., N, K] in dtype dt -> (packed uint8 [..., N, K/2], scales uint16 [..., N, K/g])"""
    w_bits = _c(w_bits, np.uint16)
    *lead, N, K = w_bits.shape
    flat = w_bits.reshape(-1, K)
    packed = np.zeros((flat.shape[0], K // 2), np.uint8)
    scales = np.empty((flat.shape[0], K // g), np.uint16)
    lib().lkm_or_quant_int4(_p(flat), C.c_int(dt), C.c_int64(flat.shape[0]), C.c_int64(K), C.c_int(g),
                            _p(packed), _p(scales))
    return packed.reshape(*lead, N, K // 2), scales.reshape(*lead, N, K // g)


def quant_fp8_block(w: np.ndarray, gN: int, gK: int):
    """w: f32 [E, N, K] -> (q uint8 [E,N,K], scales f32 [E, ceil(N/gN), ceil(K/gK)])"""
    w = _c(w, np.float32)
    E, N, K = w.shape
    nb, kb = -(-N // gN), -(-K // gK)
    q = np.empty((E, N, K), np.uint8)
    s = np.empty((E, nb, kb), np.float32)
    for e in range(E):
        lib().lkm_or_quant_fp8_block(_p(w[e]), C.c_int64(N), C.c_int64(K), C.c_int(gN), C.c_int(gK),
                                     _p(q[e]), _p(s[e]))
    return q, s


def dequant_rows(wfmt: int, act_dtype: int, w: np.ndarray, scale: np.ndarray, K: int, groupK: int,
                 gs=None) -> np.ndarray:
    """packed 4-bit weights uint8 [E,N,K/2] (+ scales, + NVFP4 multipliers) -> act-dtype bits [E,N,K]"""
    w = _c(w, np.uint8)
    scale = _c(scale)
    E, N = w.shape[0], w.shape[1]
    gs = None if gs is None else _c(gs, np.float32)
    out = np.empty((E, N, K), np.uint16)
    lib().lkm_or_dequant_rows(C.c_int(wfmt), C.c_int(act_dtype), _p(w), _p(scale), _p(gs), C.c_int64(E),
                              C.c_int64(N), C.c_int64(K), C.c_int(groupK), _p(out))
    return out


def dequant_wna16(q: np.ndarray, scale_bits: np.ndarray, zp: np.ndarray | None, weight_bits: int, group: int,
                  act_dtype: int) -> np.ndarray:
    """Weight-only integer experts with or without zero points -> act-dtype bits [E, N, K]: the dequantisation of
    the in-tree operator, `((b - zp) * scale).to(compute_type)` with zp = 8 / 128 when symmetric
    (vllm/model_executor/layers/fused_moe/fused_moe.py:207-276), layouts as the reference's test packs them
    (tests/kernels/moe/test_moe.py:634-641): q uint8 [E, N, K/2] (4-bit, low nibble = even k) or [E, N, K];
    scales [E, N, K/group] act-dtype bits; zp uint8 [E, N/2, K/group] (4-bit, low nibble = even n) or
    [E, N, K/group] (8-bit), None = symmetric.  The product of a small integer and a 16-bit scale is exact in
    fp32, so one rounding to the act dtype reproduces quantize_weights' w_ref (quant_utils.py:703-710) bit for
    bit -- pinned by tests/golden/moe_wna16.npz."""
    q = _c(q, np.uint8)
    E, N = q.shape[0], q.shape[1]
    if weight_bits == 4:
        v = np.empty((E, N, q.shape[2] * 2), np.float32)
        v[..., 0::2] = q & 0xF
        v[..., 1::2] = q >> 4
    else:
        assert weight_bits == 8
        v = q.astype(np.float32)
    K = v.shape[2]
    if zp is None:
        z = np.float32(8.0 if weight_bits == 4 else 128.0)
    else:
        zp = _c(zp, np.uint8)
        if weight_bits == 4:
            zf = np.empty((E, N, zp.shape[2]), np.float32)
            zf[:, 0::2] = zp & 0xF
            zf[:, 1::2] = zp >> 4
        else:
            zf = zp.astype(np.float32)
        z = np.repeat(zf, group, axis=2)[..., :K]
    s = np.repeat(bits_to_f32(scale_bits, act_dtype), group, axis=2)[..., :K]
    return f32_to_bits(((v - z) * s).astype(np.float32), act_dtype)


def router_logits(x: np.ndarray, x_dt: int, w: np.ndarray, w_dt: int, bias=None, round_dt: int = F32) -> np.ndarray:
    """x [M,H] (uint16 bits of x_dt), w [E,H] (uint16 bits or float32) -> fp32 logits [M,E]"""
    x, w = _c(x), _c(w)
    M, H = x.shape
    E = w.shape[0]
    bias = None if bias is None else _c(bias, np.float32)
    out = np.empty((M, E), np.float32)
    lib().lkm_or_router_logits(_p(x), C.c_int(x_dt), _p(w), C.c_int(w_dt), _p(bias), C.c_int(M), C.c_int(H),
                               C.c_int(E), C.c_int(round_dt), _p(out))
    return out


def set_threads(n: int) -> None:
    lib().lkm_or_configure(C.c_int(int(n)))


def num_threads() -> int:
    return int(lib().lkm_or_num_threads())


# ------------------------------------------------------------------ EPLB id map + load recording
def eplb_map_record(topk_ids: np.ndarray, log2phy: np.ndarray, logcnt: np.ndarray, load: np.ndarray | None = None,
                    record_enabled: bool = True, num_unpadded: int | None = None):
    """Restates the reference's logical -> physical map + load recording kernel
    (vllm/model_executor/layers/fused_moe/router/base_router.py:24-97, `_eplb_map_and_record_i32_kernel`):
      slot i of token t = i // top_k;  valid = 0 <= id < num_logical
      replica = ((t * 2654435769) & 0xFFFFFFFF) % max(logcnt[id], 1)             (:48-54)
      phys    = log2phy[id, replica] if valid else -1                             (:55-60)
      load[phys] += 1 if record_enabled and i < num_unpadded * top_k and 0 <= phys < len(load)   (:76-93)
    topk_ids [M, K] int; log2phy [E, R]; logcnt [E]; load [P] int32 (a copy is updated and returned).
    Parity: the Triton kernel itself cannot run here (no Triton, no GPU); the reference's only test of it
    (tests/kernels/moe/test_routing.py:155-188, identity map, one replica: ids unchanged) is reproduced in
    tests/test_eplb.py.  Deviation, unreachable with consistent maps: a replica count above the map width
    is clamped to it (the reference would read out of bounds).  Returns (physical ids int32 [M, K], load)."""
    ids = np.asarray(topk_ids, dtype=np.int64)
    M, K = ids.shape
    l2p = np.asarray(log2phy, dtype=np.int64)
    cnt = np.asarray(logcnt, dtype=np.int64)
    E, R = l2p.shape
    flat = ids.reshape(-1)
    valid = (flat >= 0) & (flat < E)
    safe = np.where(valid, flat, 0)
    c = np.clip(cnt[safe], 1, R)
    tok = np.arange(flat.size, dtype=np.int64) // K
    hashed = (tok * 2654435769) & 0xFFFFFFFF
    phys = np.where(valid, l2p[safe, hashed % c], -1)
    out_load = None if load is None else np.array(load, dtype=np.int32, copy=True)
    if out_load is not None and record_enabled:
        sel = (phys >= 0) & (phys < out_load.size)
        if num_unpadded is not None:
            sel &= np.arange(flat.size) < int(num_unpadded) * K
        np.add.at(out_load, phys[sel], 1)
    return phys.reshape(M, K).astype(np.int32), out_load


# ------------------------------------------------------------------ the scatter / gather step as operators (numpy)
# CPU restatements of the reference's operator forms of SURVEY 8 a9, pinned against vectors produced by the reference's own
# golden functions (tests/golden/moe_ops.npz, tests/test_oracle_golden.py); the GPU tests compare lvllm_amd.ops with these at
# sizes the goldens do not hold (tests/test_gpu_moe_ops.py).
def moe_align_block_size(topk_ids: np.ndarray, block_size: int, num_experts: int, expert_map: np.ndarray | None = None,
                         pad_sorted_ids: bool = False):
    """moe_align_block_size (vllm/model_executor/layers/fused_moe/moe_align_block_size.py:11-103; golden
    tests/kernels/moe/test_moe_align_block_size.py:96-172): slots sorted by expert (stable), every expert padded to a multiple
    of block_size with the value numel; expert_map (the ignore_invalid_experts form): ids mapped to -1 -- and ids < 0 -- take
    no part, expert_ids hold the mapped ids.  -> (sorted_ids int32 [max_padded], expert_ids int32 [blocks], total int)."""
    ids = np.asarray(topk_ids, np.int64).reshape(-1)
    n = ids.size
    max_padded = n + num_experts * (block_size - 1)
    if pad_sorted_ids:
        max_padded = -(-max_padded // block_size) * block_size
    if n < num_experts:
        max_padded = min(n * block_size, max_padded)
    sorted_ids = np.full(max_padded, n, np.int32)
    expert_ids = np.full(-(-max_padded // block_size) if max_padded else 0, -1, np.int32)
    pos = blk = 0
    order = range(num_experts)
    if expert_map is not None:
        # the KERNEL counts and ranks by the mapped id (get_local_expert_id, csrc/.../moe_align_sum_kernels.cu:86-100,144-185):
        # blocks come in LOCAL-id order.  The reference's golden walks global ids (test_moe_align_block_size.py:150-172);
        # the two agree for the monotone maps its test uses (:250-262), and only there.
        em = np.asarray(expert_map, np.int64)
        order = [int(e) for e in np.argsort(np.where(em >= 0, em, np.iinfo(np.int64).max), kind="stable") if em[e] >= 0]
    for e in order:
        if expert_map is not None and expert_map[e] < 0:
            continue
        rows = np.nonzero(ids == e)[0]
        if rows.size == 0:
            continue
        padded = -(-rows.size // block_size) * block_size
        sorted_ids[pos:pos + rows.size] = rows
        expert_ids[blk:blk + padded // block_size] = e if expert_map is None else expert_map[e]
        pos += padded
        blk += padded // block_size
    return sorted_ids, expert_ids, pos


def moe_permute(topk_ids: np.ndarray, n_expert: int, n_local_expert: int | None = None, expert_map: np.ndarray | None = None):
    """moe_permute's index outputs (moe_permute_unpermute.py:105-242; golden tests/kernels/moe/test_moe_permute_unpermute.py:
    37-89): slots sorted (stable) by local expert id, slots of experts that are not local behind them by global id.
    -> (expert_first_token_offset int64 [n_local + 1], inv_permuted_idx int32 [n] slot -> row, permuted_idx int32 [n] row ->
    slot with n for the rows of non-local experts)."""
    ids = np.asarray(topk_ids, np.int64).reshape(-1)
    n = ids.size
    n_local = n_expert if n_local_expert is None else n_local_expert
    if expert_map is None:
        key = ids.copy()
    else:
        loc = np.asarray(expert_map, np.int64)[ids]
        key = np.where(loc >= 0, loc, ids + n_expert)
    order = np.argsort(key, kind="stable")
    first = np.zeros(n_local + 1, np.int64)
    first[1:] = np.cumsum(np.bincount(key[key < n_local], minlength=n_local)[:n_local])
    inv = np.empty(n, np.int32)
    inv[order] = np.arange(n, dtype=np.int32)
    perm = order.astype(np.int32)
    perm[first[-1]:] = n
    return first, inv, perm


def moe_unpermute(rows_bits: np.ndarray, dt: int, topk_weights: np.ndarray, inv_permuted_idx: np.ndarray, n_valid: int) -> np.ndarray:
    """moe_unpermute (moe_permute_unpermute.py:245-283): out[t] = T(sum_k w[t, k] * rows[inv[t, k]]) over the valid rows, fp32
    sum in slot order, one rounding to the rows' dtype.  rows as bit patterns [R, H]; -> bit patterns [M, H]."""
    rows = bits_to_f32(rows_bits, dt)
    tw = np.asarray(topk_weights, np.float32)
    M, K = tw.shape
    inv = np.asarray(inv_permuted_idx, np.int64).reshape(M, K)
    acc = np.zeros((M, rows.shape[1]), np.float32)
    for k in range(K):
        ok = (inv[:, k] >= 0) & (inv[:, k] < n_valid)
        acc[ok] += tw[ok, k:k + 1] * rows[inv[ok, k]]
    return f32_to_bits(acc, dt)
