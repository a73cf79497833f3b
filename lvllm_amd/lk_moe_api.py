"""MI355X-native re-creation of the `lk_moe` Python surface that LvLLM's RoutedExperts drives.

Reference call sites (relative to the reference tree), all in
vllm/model_executor/layers/fused_moe/routed_experts.py:
  config object        lk_moe.MOEConfigV2()                          :1490-1511
  engine construction  lk_moe.MOE_{BF16,FP16,FP8,FP8_FP16,WNA16,WNA16_FP16,NVFP4,...}(cfg, 6 ptrs)
                                                                      :1514-1533, 1596-1616, 1648-1668
  decode               .cpu_decode(stream, M, top_k, hidden, ids, weights, out_f32)   :1840-1855
  host prefill         .cpu_prefill(M, top_k, ids, weights, hidden, out_f32)          :1858-1882
  device prefill       .gpu_prefill(hidden, out, ids, weights, M, top_k, stream)      :1884-1899

Same names, argument order and meaning (raw integer pointers; 0 = absent).  The method names keep
the reference's "cpu_" prefix for drop-in compatibility, but here *every* path runs on the GPU
through liblkm.so: the CPU-NUMA tier of the original engine is replaced by HBM residency.  Errors
surface as Python exceptions (LkmError) both at construction -- where the reference's caller
catches and logs them, :1406-1418 -- and in the forward calls.
"""
from __future__ import annotations

import ctypes as C

from . import _clib


class MOEConfigV2:
    """Plain settable attributes, default-constructible (routed_experts.py:1490-1511)."""

    def __init__(self) -> None:
        self.num_processes = 1
        self.process_id = 0
        self.gpu_id = 0
        self.has_gate_proj = True
        self.expert_num = 0
        self.top_k = 0
        self.hidden_size = 0
        self.intermediate_size = 0
        self.max_batch_size = 0
        self.max_num_seqs = 0
        self.stride = 32
        self.group_min_len = 10
        self.group_max_len = 0
        self.groupN = 0
        self.groupK = 0
        self.activation_type = 0          # 0 silu-gated, 1 swigluoai, 2 relu2 (no gate)
        self.swiglu_alpha = 1.702
        self.swiglu_limit = 7.0
        self.use_gpu_prefill = False
        # extension (not in the reference): fp8 compute mode, 0 = W8A16 (lk_moe semantics)
        self.fp8_mode = _clib.FP8_W8A16
        # extension: uint4b8 compute mode, 0 = the reference's rounding T((q-8) s), 1 = scale on fp32 partial sums
        self.int4_mode = _clib.INT4_EXACT

    def _to_c(self, weight_format: int, act_dtype: int) -> _clib.LkmConfig:
        c = _clib.LkmConfig()
        c.abi_version = _clib.LKM_ABI_VERSION
        for name in ("num_processes", "process_id", "gpu_id", "expert_num", "top_k", "hidden_size",
                     "intermediate_size", "max_batch_size", "max_num_seqs", "stride",
                     "group_min_len", "group_max_len", "groupN", "groupK", "activation_type",
                     "fp8_mode", "int4_mode"):
            setattr(c, name, int(getattr(self, name)))
        c.has_gate_proj = int(bool(self.has_gate_proj))
        c.use_gpu_prefill = int(bool(self.use_gpu_prefill))
        c.swiglu_alpha = float(self.swiglu_alpha)
        c.swiglu_limit = float(self.swiglu_limit)
        c.weight_format = weight_format
        c.act_dtype = act_dtype
        return c


def _vp(x: int) -> C.c_void_p:
    return C.c_void_p(int(x) if x else None)


class _MOE:
    _WEIGHT_FORMAT = -1
    _ACT_DTYPE = -1

    def __init__(self, cfg: MOEConfigV2, w13_ptr: int, w2_ptr: int, w13_scale_ptr: int = 0,
                 w2_scale_ptr: int = 0, w13_global_scale_ptr: int = 0,
                 w2_global_scale_ptr: int = 0) -> None:
        self._h = C.c_void_p()
        self._lib = _clib.lib()
        ccfg = cfg._to_c(self._WEIGHT_FORMAT, self._ACT_DTYPE)
        _clib.check(self._lib.lkm_create(C.byref(ccfg), _vp(w13_ptr), _vp(w2_ptr), _vp(w13_scale_ptr),
                                         _vp(w2_scale_ptr), _vp(w13_global_scale_ptr),
                                         _vp(w2_global_scale_ptr), C.byref(self._h)))

    # ---- reference surface -------------------------------------------------------------
    def cpu_decode(self, stream: int, num_tokens: int, top_k: int, hidden_ptr: int,
                   topk_ids_ptr: int, topk_weights_ptr: int, out_f32_ptr: int) -> None:
        _clib.check(self._lib.lkm_decode(self._h, _vp(stream), num_tokens, top_k, _vp(hidden_ptr),
                                         _vp(topk_ids_ptr), _vp(topk_weights_ptr), _vp(out_f32_ptr)))

    def cpu_prefill(self, num_tokens: int, top_k: int, ids_i32_ptr: int, w_f32_ptr: int,
                    hidden_ptr: int, out_f32_ptr: int) -> None:
        _clib.check(self._lib.lkm_prefill_host(self._h, num_tokens, top_k, _vp(ids_i32_ptr),
                                               _vp(w_f32_ptr), _vp(hidden_ptr), _vp(out_f32_ptr)))

    def gpu_prefill(self, hidden_ptr: int, out_ptr: int, topk_ids_ptr: int, topk_weights_ptr: int,
                    num_tokens: int, top_k: int, stream: int) -> None:
        _clib.check(self._lib.lkm_prefill_device(self._h, _vp(hidden_ptr), _vp(out_ptr),
                                                 _vp(topk_ids_ptr), _vp(topk_weights_ptr),
                                                 num_tokens, top_k, _vp(stream)))

    # ---- extension: the same operator on strided records (expert-parallel exchange, include/lkm.h) ----
    def forward_strided(self, stream: int, num_tokens: int, top_k: int, hidden_ptr: int, hidden_ld: int,
                        topk_ids_ptr: int, ids_ld: int, id_offset: int, topk_weights_ptr: int, weights_ld: int,
                        out_ptr: int, out_dtype: int) -> None:
        _clib.check(self._lib.lkm_forward_strided(self._h, _vp(stream), num_tokens, top_k, _vp(hidden_ptr),
                                                  int(hidden_ld), _vp(topk_ids_ptr), int(ids_ld), int(id_offset),
                                                  _vp(topk_weights_ptr), int(weights_ld), _vp(out_ptr),
                                                  int(out_dtype)))

    # ---- extension: router + experts in one call (decode: router and scatter metadata in one launch) ----
    def forward_routed(self, stream: int, num_tokens: int, top_k: int, hidden_ptr: int, hidden_ld: int,
                       logits_ptr: int, logits_dtype: int, router_experts: int, score_bias_ptr: int,
                       n_group: int, topk_group: int, scoring: int, renormalize: bool, routed_scaling: float,
                       id_offset: int, topk_weights_ptr: int, topk_ids_ptr: int, out_ptr: int,
                       out_dtype: int) -> None:
        _clib.check(self._lib.lkm_forward_routed(self._h, _vp(stream), num_tokens, top_k, _vp(hidden_ptr),
                                                 int(hidden_ld), _vp(logits_ptr), int(logits_dtype),
                                                 int(router_experts), _vp(score_bias_ptr), int(n_group),
                                                 int(topk_group), int(scoring), int(bool(renormalize)),
                                                 float(routed_scaling), int(id_offset), _vp(topk_weights_ptr),
                                                 _vp(topk_ids_ptr), _vp(out_ptr), int(out_dtype)))

    # ---- measurement / introspection (extensions) --------------------------------------
    def set_profiling(self, enable: bool) -> None:
        _clib.check(self._lib.lkm_set_profiling(self._h, int(enable)))

    def get_profile(self) -> dict[str, float]:
        ms = (C.c_float * _clib.PROF_N)()
        _clib.check(self._lib.lkm_get_profile(self._h, ms))
        return {"sort": ms[0], "gemm1": ms[1], "gemm2": ms[2], "combine": ms[3]}

    def weight_bytes(self) -> int:
        return int(self._lib.lkm_weight_bytes(self._h))

    def describe(self) -> str:
        buf = C.create_string_buffer(1024)
        _clib.check(self._lib.lkm_describe(self._h, buf, 1024))
        return buf.value.decode()

    def set_tuning(self, **kv: int) -> None:
        for k, v in kv.items():
            _clib.check(self._lib.lkm_set_tuning(self._h, k.encode(), int(v)))

    def last_kernels(self) -> dict[str, list[str]]:
        """{"gemm1": [...], "gemm2": [...]}: the GEMM kernels of the last step, named as rocprofv3 prints them
        (include/lkm.h lkm_last_kernels; two names where a hybrid plan ran the streamer and the tile kernel)"""
        buf = C.create_string_buffer(2048)
        _clib.check(self._lib.lkm_last_kernels(self._h, buf, 2048))
        out = {}
        for part in buf.value.decode().split(";"):
            k, _, v = part.partition("=")
            out[k] = [n for n in v.split("+") if n]
        return out

    def tuned_plans(self) -> list[tuple[int, int]]:
        """[(shape key, candidate index)] of the plans the first-call autotune remembers, ascending keys"""
        n = int(self._lib.lkm_tuned_plans(self._h, None, None, 0))
        if n < 0:
            _clib.check(n)
        keys, idx = (C.c_int64 * max(n, 1))(), (C.c_int32 * max(n, 1))()
        n = min(n, int(self._lib.lkm_tuned_plans(self._h, keys, idx, n)))
        return [(int(keys[i]), int(idx[i])) for i in range(n)]

    def set_tuned_plan(self, key: int, index: int) -> None:
        _clib.check(self._lib.lkm_tuned_plan_set(self._h, int(key), int(index)))

    # ---- expert images for placement changes (include/lkm_eplb.h; used by lvllm_amd/eplb.py) --------
    def expert_bytes(self) -> int:
        n = int(self._lib.lkm_expert_bytes(self._h))
        if n < 0:
            _clib.check(n)
        return n

    def export_expert(self, stream: int, expert: int, dst_ptr: int) -> None:
        _clib.check(self._lib.lkm_export_expert(self._h, _vp(stream), int(expert), _vp(dst_ptr)))

    def import_expert(self, stream: int, expert: int, src_ptr: int) -> None:
        _clib.check(self._lib.lkm_import_expert(self._h, _vp(stream), int(expert), _vp(src_ptr)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.lkm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:
            pass


def _mk(name: str, wf: int, dt: int, doc: str):
    return type(name, (_MOE,), {"_WEIGHT_FORMAT": wf, "_ACT_DTYPE": dt, "__doc__": doc})


MOE_BF16 = _mk("MOE_BF16", _clib.W_BF16, _clib.DT_BF16, "bf16 weights, bf16 activations")
MOE_FP16 = _mk("MOE_FP16", _clib.W_F16, _clib.DT_F16, "fp16 weights, fp16 activations")
MOE_FP8 = _mk("MOE_FP8", _clib.W_FP8_E4M3, _clib.DT_BF16, "e4m3fn block-scaled weights, bf16 activations")
MOE_FP8_FP16 = _mk("MOE_FP8_FP16", _clib.W_FP8_E4M3, _clib.DT_F16, "e4m3fn weights, fp16 activations")
MOE_WNA16 = _mk("MOE_WNA16", _clib.W_INT4_B8, _clib.DT_BF16, "uint4b8 group-scaled weights, bf16 activations")
MOE_WNA16_FP16 = _mk("MOE_WNA16_FP16", _clib.W_INT4_B8, _clib.DT_F16, "uint4b8 weights, fp16 activations")
# E2M1 formats (routed_experts.py:1673-1813).  NVFP4: fp8 e4m3fn scale per 16 k (linear
# [E,N,K/16]) + per-expert f32 multipliers (the reference passes 1/global_scale when
# need_reciprocal_global_scale); MXFP4: E8M0 scale per 32 k, no global scale (pass 0, 0).
MOE_NVFP4 = _mk("MOE_NVFP4", _clib.W_NVFP4, _clib.DT_BF16, "NVFP4 weights (W4A16), bf16 activations")
MOE_NVFP4_FP16 = _mk("MOE_NVFP4_FP16", _clib.W_NVFP4, _clib.DT_F16, "NVFP4 weights (W4A16), fp16 activations")
MOE_MXFP4 = _mk("MOE_MXFP4", _clib.W_MXFP4, _clib.DT_BF16, "MXFP4 weights (W4A16), bf16 activations")
MOE_MXFP4_FP16 = _mk("MOE_MXFP4_FP16", _clib.W_MXFP4, _clib.DT_F16, "MXFP4 weights (W4A16), fp16 activations")

__all__ = ["MOEConfigV2", "MOE_BF16", "MOE_FP16", "MOE_FP8", "MOE_FP8_FP16", "MOE_WNA16",
           "MOE_WNA16_FP16", "MOE_NVFP4", "MOE_NVFP4_FP16", "MOE_MXFP4", "MOE_MXFP4_FP16"]
