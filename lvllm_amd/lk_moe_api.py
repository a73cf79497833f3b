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

import contextlib
import ctypes as C
import os
import threading

from . import _clib

# ---- the spill tier behind the same three calls (routed_experts.py:1344-1357, 1884-1899; vllm/envs.py:265,1942-1943) -------
# The reference keeps a GPU-prefill layer's experts in host RAM and streams them through the GPU.  Here every layer is
# HBM-resident -- until it does not fit: lkm_create answers LKM_E_NOMEM (or the planning cap LKM_HBM_CAP_BYTES says the layer
# is too big, residency.expert_layer_bytes).  With HOST weight pointers -- what the reference's glue passes -- the constructor
# then builds lvllm_amd.spill.HostResidentExperts from the same six pointers and cpu_decode / cpu_prefill / gpu_prefill run
# on it: pinned host images, 2 x LVLLM_GPU_PREFETCH_WINDOW device slots, copies under compute.  Eager only, as the reference's
# gpu_prefill (never under graph capture, :1350-1355).
_tls = threading.local()
_FMT = {_clib.W_BF16: "bf16", _clib.W_F16: "fp16", _clib.W_FP8_E4M3: "fp8", _clib.W_INT4_B8: "int4", _clib.W_NVFP4: "nvfp4",
        _clib.W_MXFP4: "mxfp4"}


@contextlib.contextmanager
def spill_disabled():
    """inside: a MOE_* that does not fit raises instead of spilling (the spill tier builds its own window engines with it)"""
    prev = getattr(_tls, "no_spill", False)
    _tls.no_spill = True
    try:
        yield
    finally:
        _tls.no_spill = prev


def hbm_cap_bytes(env=None) -> int:
    """LKM_HBM_CAP_BYTES: treat the GPU as having this many bytes for ONE layer's experts (planning / tests; 0 = no cap, the
    allocation itself decides)"""
    env = os.environ if env is None else env
    try:
        return max(0, int(env.get("LKM_HBM_CAP_BYTES", "0")))
    except ValueError:
        return 0


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
        # extension: uint4 compute mode, 0 = the reference's rounding T((q-8) s), 1 = scale on fp32 partial sums,
        # 2 = zero points (uint8 [E, rows, K / group] in the two global-scale pointer slots): T((q - zp) s)
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
        self._spill = None
        ccfg = cfg._to_c(self._WEIGHT_FORMAT, self._ACT_DTYPE)
        ptrs = (w13_ptr, w2_ptr, w13_scale_ptr, w2_scale_ptr, w13_global_scale_ptr, w2_global_scale_ptr)
        # (zero-point experts -- LkmConfig.int4_mode ZP, an extension the reference's lk_moe surface does not have -- stay resident)
        may_spill = not getattr(_tls, "no_spill", False) and int(getattr(cfg, "int4_mode", 0)) != _clib.INT4_ZP
        cap = hbm_cap_bytes() if may_spill else 0
        too_big = False
        if cap:
            from .residency import expert_layer_bytes
            too_big = expert_layer_bytes(int(cfg.expert_num), int(cfg.hidden_size), int(cfg.intermediate_size),
                                         _FMT[self._WEIGHT_FORMAT], int(cfg.groupK) or 128, bool(cfg.has_gate_proj)) > cap
        if not too_big:
            rc = self._lib.lkm_create(C.byref(ccfg), *[_vp(x) for x in ptrs], C.byref(self._h))
            if rc == _clib.E_NOMEM and may_spill and self._host_pointers(ptrs):
                too_big = True
            else:
                _clib.check(rc)
        if too_big:
            if not self._host_pointers(ptrs):
                raise _clib.LkmError(_clib.E_NOMEM, "the layer's experts exceed the HBM budget and its weights are device "
                                                    "tensors: the spill tier needs host-resident weights")
            self._spill = self._build_spill(cfg, ptrs)

    # ---- spill tier (see the module header) ------------------------------------------------------
    def _host_pointers(self, ptrs) -> bool:
        return all(not self._lib.lkm_pointer_is_device(_vp(x)) for x in ptrs if x)

    def _build_spill(self, cfg: "MOEConfigV2", ptrs):
        import numpy as np
        import torch

        from .spill import HostResidentExperts
        wf, adt = self._WEIGHT_FORMAT, self._ACT_DTYPE
        tdt = torch.bfloat16 if adt == _clib.DT_BF16 else torch.float16
        E, H, I = int(cfg.expert_num), int(cfg.hidden_size), int(cfg.intermediate_size)
        n13 = (2 if cfg.has_gate_proj else 1) * I
        gN, gK = max(1, int(cfg.groupN)), int(cfg.groupK)

        def host(ptr, shape, dtype):       # a view of the caller's host array (read during construction only)
            if not ptr:
                return None
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            raw = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(int(ptr)))
            return torch.from_numpy(raw).view(dtype).reshape(shape)

        def gk(k):                         # the scale group of a GEMM: min(groupK, its K)  (include/lkm.h LkmConfig.groupK)
            return min(gK, k) if gK > 0 else k
        cdiv = lambda a, b: -(-a // b)
        if wf in (_clib.W_BF16, _clib.W_F16):
            shapes = [((E, n13, H), tdt), ((E, H, I), tdt), None, None, None, None]
        elif wf == _clib.W_FP8_E4M3:
            shapes = [((E, n13, H), torch.uint8), ((E, H, I), torch.uint8),
                      ((E, cdiv(n13, gN), cdiv(H, gk(H))), torch.float32), ((E, cdiv(H, gN), cdiv(I, gk(I))), torch.float32), None, None]
        elif wf == _clib.W_INT4_B8:
            shapes = [((E, n13, H // 2), torch.uint8), ((E, H, I // 2), torch.uint8),
                      ((E, n13, H // gk(H)), tdt), ((E, H, I // gk(I)), tdt), None, None]
        else:
            g = 16 if wf == _clib.W_NVFP4 else 32
            gs = ((E,), torch.float32) if wf == _clib.W_NVFP4 else None
            shapes = [((E, n13, H // 2), torch.uint8), ((E, H, I // 2), torch.uint8),
                      ((E, n13, H // g), torch.uint8), ((E, H, I // g), torch.uint8), gs, gs]
        t = [None if (sh is None or not p_) else host(p_, *sh) for p_, sh in zip(ptrs, shapes)]
        return HostResidentExperts(
            t[0], t[1], top_k=int(cfg.top_k), act_dtype=tdt, fmt=_FMT[wf], w13_scale=t[2], w2_scale=t[3],
            w13_global_scale=t[4], w2_global_scale=t[5], device=torch.device("cuda", int(cfg.gpu_id)),
            group_n=int(cfg.groupN), group_k=int(cfg.groupK), has_gate_proj=bool(cfg.has_gate_proj),
            activation_type=int(cfg.activation_type), swiglu_alpha=float(cfg.swiglu_alpha), swiglu_limit=float(cfg.swiglu_limit),
            max_num_seqs=int(cfg.max_num_seqs), max_batch_size=int(cfg.max_batch_size), group_max_len=int(cfg.group_max_len),
            num_processes=int(cfg.num_processes), process_id=int(cfg.process_id), fp8_mode=int(cfg.fp8_mode),
            int4_mode=int(cfg.int4_mode))

    def _spill_forward(self, stream: int, M: int, K: int, hidden_ptr: int, ids_ptr: int, w_ptr: int, out_ptr: int, out_f32: bool):
        import torch
        sp = self._spill
        dev = sp.dev

        class _Dev:                        # a device array behind a raw pointer (__cuda_array_interface__)
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}
        tdt = sp.act_dtype
        with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev) if stream
                                                       else torch.cuda.default_stream(dev)):
            x = torch.as_tensor(_Dev(hidden_ptr, (M, sp.H), "<i2"), device=dev).view(tdt)
            ids = torch.as_tensor(_Dev(ids_ptr, (M, K), "<i4"), device=dev)
            tw = torch.as_tensor(_Dev(w_ptr, (M, K), "<f4"), device=dev)
            if out_f32:
                out = torch.as_tensor(_Dev(out_ptr, (M, sp.H), "<f4"), device=dev)
                out.copy_(sp.forward(x, tw, ids, torch.float32))
            else:
                out = torch.as_tensor(_Dev(out_ptr, (M, sp.H), "<i2"), device=dev).view(tdt)
                out.copy_(sp.forward(x, tw, ids, tdt))

    # ---- reference surface -------------------------------------------------------------
    def cpu_decode(self, stream: int, num_tokens: int, top_k: int, hidden_ptr: int,
                   topk_ids_ptr: int, topk_weights_ptr: int, out_f32_ptr: int) -> None:
        if self._spill is not None:
            return self._spill_forward(stream, num_tokens, top_k, hidden_ptr, topk_ids_ptr, topk_weights_ptr, out_f32_ptr, True)
        _clib.check(self._lib.lkm_decode(self._h, _vp(stream), num_tokens, top_k, _vp(hidden_ptr),
                                         _vp(topk_ids_ptr), _vp(topk_weights_ptr), _vp(out_f32_ptr)))

    def cpu_prefill(self, num_tokens: int, top_k: int, ids_i32_ptr: int, w_f32_ptr: int,
                    hidden_ptr: int, out_f32_ptr: int) -> None:
        if self._spill is not None:
            import numpy as np
            import torch
            sp = self._spill

            def host(ptr, n, dt):
                return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * n).from_address(int(ptr)))).view(dt)
            M, K = int(num_tokens), int(top_k)
            x = host(hidden_ptr, M * sp.H * 2, sp.act_dtype).reshape(M, sp.H).to(sp.dev)
            ids = host(ids_i32_ptr, M * K * 4, torch.int32).reshape(M, K).to(sp.dev)
            tw = host(w_f32_ptr, M * K * 4, torch.float32).reshape(M, K).to(sp.dev)
            host(out_f32_ptr, M * sp.H * 4, torch.float32).reshape(M, sp.H).copy_(sp.forward(x, tw, ids, torch.float32))
            return
        _clib.check(self._lib.lkm_prefill_host(self._h, num_tokens, top_k, _vp(ids_i32_ptr),
                                               _vp(w_f32_ptr), _vp(hidden_ptr), _vp(out_f32_ptr)))

    def gpu_prefill(self, hidden_ptr: int, out_ptr: int, topk_ids_ptr: int, topk_weights_ptr: int,
                    num_tokens: int, top_k: int, stream: int) -> None:
        if self._spill is not None:
            return self._spill_forward(stream, num_tokens, top_k, hidden_ptr, topk_ids_ptr, topk_weights_ptr, out_ptr, False)
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
        if self._spill is not None:
            return int(self._spill.device_bytes())        # what the tier keeps in HBM (the experts themselves: host_bytes())
        return int(self._lib.lkm_weight_bytes(self._h))

    def describe(self) -> str:
        if self._spill is not None:
            sp = self._spill
            return (f"spill tier: {sp.E} experts in pinned host memory ({sp.host_bytes()} B), {sp.slots} device slots "
                    f"(window {sp.window}) | " + sp.engine.engine.describe())
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
        if getattr(self, "_spill", None) is not None:
            self._spill.close()
            self._spill = None
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
