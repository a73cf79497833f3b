"""ctypes binding of liblkm.so (include/lkm.h).  This is the ONLY way Python reaches the kernels.

There is no fallback: if the shared library is missing or fails to load, importing the product
path raises.  (`python -m lvllm_amd.build` / `__graft_entry__.build()` compile it with hipcc.)
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os

PKG = Path(__file__).resolve().parent
# LKM_LIB_PATH: development override to A/B an experimental build of the same library
LIB_PATH = Path(os.environ.get("LKM_LIB_PATH", PKG / "liblkm.so"))

LKM_ABI_VERSION = 1
OK, E_INVALID, E_HIP, E_NOMEM, E_UNSUPPORTED = 0, -1, -2, -3, -4
DT_F32, DT_BF16, DT_F16 = 0, 1, 2
W_BF16, W_F16, W_FP8_E4M3, W_INT4_B8, W_NVFP4, W_MXFP4 = 0, 1, 2, 3, 4, 5
ACT_SILU, ACT_SWIGLUOAI, ACT_RELU2 = 0, 1, 2
FP8_W8A16, FP8_W8A8 = 0, 1
INT4_EXACT, INT4_FAST, INT4_ZP = 0, 1, 2
PROF_SORT, PROF_GEMM1, PROF_GEMM2, PROF_COMBINE, PROF_N = 0, 1, 2, 3, 4


class LkmConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("num_processes", C.c_int32), ("process_id", C.c_int32),
        ("gpu_id", C.c_int32), ("has_gate_proj", C.c_int32), ("expert_num", C.c_int32),
        ("top_k", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
        ("max_batch_size", C.c_int32), ("max_num_seqs", C.c_int32), ("stride", C.c_int32),
        ("group_min_len", C.c_int32), ("group_max_len", C.c_int32), ("groupN", C.c_int32),
        ("groupK", C.c_int32), ("activation_type", C.c_int32), ("swiglu_alpha", C.c_float),
        ("swiglu_limit", C.c_float), ("use_gpu_prefill", C.c_int32), ("weight_format", C.c_int32),
        ("act_dtype", C.c_int32), ("fp8_mode", C.c_int32), ("int4_mode", C.c_int32), ("reserved", C.c_int32 * 7),
    ]


class LkmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"liblkm error {code}: {msg}")
        self.code = code


_SIGS = {
    "lkm_create": (C.c_int, [C.POINTER(LkmConfig)] + [C.c_void_p] * 6 + [C.POINTER(C.c_void_p)]),
    "lkm_destroy": (None, [C.c_void_p]),
    "lkm_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p]),
    "lkm_prefill_host": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "lkm_prefill_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_void_p]),
    "lkm_topk_softmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                   C.c_void_p, C.c_void_p]),
    "lkm_grouped_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "lkm_router_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "lkm_router_gemm_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lkm_map_expert_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                     C.c_void_p]),
    "lkm_forward_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_int32]),
    "lkm_forward_routed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32]),
    "lkm_ep_row_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "lkm_ep_pack_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "lkm_ep_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p, C.c_int32]),
    "lkm_per_token_group_quant_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p]),
    "lkm_wna16_expand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_int32]),
    "lkm_pointer_is_device": (C.c_int, [C.c_void_p]),
    "lkm_moe_ops_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "lkm_moe_align_block_size": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lkm_moe_permute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lkm_moe_unpermute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "lkm_sort_slots": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "lkm_last_error": (C.c_char_p, []),
    "lkm_abi_version": (C.c_int, []),
    "lkm_device_info": (C.c_int, [C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "lkm_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "lkm_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "lkm_weight_bytes": (C.c_int64, [C.c_void_p]),
    "lkm_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "lkm_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "lkm_last_kernels": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32]),
    "lkm_tuned_plans": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.c_int32]),
    "lkm_tuned_plan_set": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32]),
    "lkm_hbm_read_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                     C.POINTER(C.c_float)]),
    # include/lkm_eplb.h
    "lkm_eplb_map_record": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "lkm_expert_bytes": (C.c_int64, [C.c_void_p]),
    "lkm_export_expert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "lkm_import_expert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


def _bind_hip_runtime() -> str:
    """liblkm.so is linked WITHOUT a DT_NEEDED on libamdhip64 (hipcc -no-hip-rt) so that a process
    holds exactly one HIP runtime: the one already mapped (PyTorch-ROCm bundles its own under
    torch/lib), else torch's bundled one if torch is installed (a later `import torch` then
    shares it), else the system ROCm one.  It is promoted to RTLD_GLOBAL so liblkm.so binds to it."""
    import importlib.util
    import os
    path = None
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
    except OSError:
        pass
    if path is None:
        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
    if path is None:
        for cand in ("/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so"):
            if cand.startswith("/") and not os.path.exists(cand):
                continue
            path = cand
            break
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


HIP_RUNTIME_PATH = None


def lib() -> C.CDLL:
    """Loads liblkm.so once; raises (never falls back) if it is not there."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -m lvllm_amd.build` (needs hipcc). There is no CPU fallback.")
        global HIP_RUNTIME_PATH
        HIP_RUNTIME_PATH = _bind_hip_runtime()
        cdll = C.CDLL(str(LIB_PATH))
        for name, (res, args) in _SIGS.items():
            fn = getattr(cdll, name)   # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        if cdll.lkm_abi_version() != LKM_ABI_VERSION:
            raise ImportError("liblkm.so ABI version mismatch; rebuild it")
        _lib = cdll
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise LkmError(rc, lib().lkm_last_error().decode(errors="replace"))


def device_info() -> tuple[int, str]:
    n = C.c_int32(0)
    buf = C.create_string_buffer(64)
    check(lib().lkm_device_info(C.byref(n), buf, 64))
    return n.value, buf.value.decode()
