"""lvllm_amd -- MI355X-native (gfx950) implementation of LvLLM's routed-expert MoE hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/lkm.h, built into liblkm.so),
_clib (ctypes binding), lk_moe_api (the reference's `lk_moe` class surface),
ops (router / scatter operators on torch tensors), ep (expert-parallel sharding over RCCL).
"""
__all__ = ["_clib", "lk_moe_api", "ops", "ep", "build"]
