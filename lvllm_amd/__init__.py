"""lvllm_amd -- MI355X-native (gfx950) implementation of LvLLM's routed-expert MoE hot path.

Layout: csrc/ (HIP kernels + the C ABI of include/lkm.h and include/lkm_eplb.h, built into liblkm.so),
_clib (ctypes binding), lk_moe_api (the reference's `lk_moe` class surface),
ops (router / scatter / EPLB-map operators on torch tensors + RoutedExpertsEngine),
modular (the in-tree FusedMoEExpertsModular / PrepareAndFinalize surface), layer (one MoE layer's step as an object),
ep (expert-parallel sharding over RCCL), eplb (placement policy, expert exchange, load window),
shared_experts (shared experts folded into the grouped GEMM), ingest (checkpoint -> engine layouts),
residency (layer tiers -> HBM capacity planning).
"""
__all__ = ["_clib", "lk_moe_api", "ops", "modular", "layer", "ep", "eplb", "shared_experts", "ingest", "residency",
           "build"]
