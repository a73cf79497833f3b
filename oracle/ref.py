"""oracle/_ref front-end -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).

Runs the reference's OWN in-tree CPU fused-MoE kernels -- 16-bit experts: /root/reference/csrc/cpu/cpu_fused_moe.cpp:640-702;
quantised experts (fp8-W8A16 block scales, MXFP4): /root/reference/csrc/cpu/sgl-kernels/moe.cpp:874-1224 --
compiled by oracle/Makefile (`make ref`) from the sources where they lie into oracle/_ref/liblkm_ref.so.  The
product path (lvllm_amd/, lk_moe/) never imports this module.  The Python wrappers below restate the two
thin wrappers of vllm/_custom_ops.py:3917-3962 (allocate the output, call the op).

The reference's real engine for the path is the closed lk_moe wheel (not in /root/reference); this kernel is
the nearest thing the reference tree can actually run: bf16/fp16 experts, gated activations, fp32 accumulate,
one rounding of the activation to the activation dtype, output in the activation dtype
(csrc/cpu/cpu_fused_moe.cpp:229-635) -- the rounding convention the oracle and the HIP kernels follow.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import torch

HERE = Path(__file__).resolve().parent
LIB = HERE / "_ref" / "liblkm_ref.so"
REF_ROOT = Path("/root/reference")
_loaded = False
_omp = None


def build(verbose: bool = False) -> Path | None:
    """make ref, when the reference tree is present (this container); the GPU box only uses the prebuilt file."""
    if not (REF_ROOT / "csrc" / "cpu" / "cpu_fused_moe.cpp").exists():
        return LIB if LIB.exists() else None
    r = subprocess.run(["make", "-C", str(HERE), "ref"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"oracle/_ref build failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
    if verbose:
        print(r.stdout[-400:])
    return LIB


def available() -> bool:
    return LIB.exists()


def load() -> None:
    global _loaded, _omp
    if _loaded:
        return
    if not LIB.exists():
        raise RuntimeError(f"{LIB} not built (run `make -C oracle ref` where /root/reference exists)")
    torch.ops.load_library(str(LIB))
    try:        # the kernel's OpenMP runtime is clang's libomp (see oracle/Makefile), not torch's libgomp
        _omp = ctypes.CDLL("libomp.so")
    except OSError:
        _omp = None
    _loaded = True
    # libomp's default team is every logical CPU of the host; containers usually may run far fewer, and
    # oversubscribed spinning teams are slow -- start from the CPUs this process may actually use, at most 32
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    set_threads(max(1, min(usable, 32)))


def set_threads(n: int) -> None:
    load()
    if _omp is not None:
        _omp.omp_set_num_threads(int(n))


def num_threads() -> int:
    load()
    return int(_omp.omp_get_max_threads()) if _omp is not None else 1


def prepack(weight: torch.Tensor, isa: str = "vec") -> torch.Tensor:
    """[E, N, K] contiguous bf16/fp16 (N % 32 == 0) -> the kernel's packed layout (same shape/dtype)."""
    load()
    packed = torch.empty_like(weight)
    torch.ops.lkm_ref.prepack_moe_weight(weight.contiguous(), packed, isa)
    return packed


def fused_moe(x: torch.Tensor, packed_w13: torch.Tensor, packed_w2: torch.Tensor, topk_weights: torch.Tensor,
              topk_ids: torch.Tensor, act: str = "silu", isa: str = "vec") -> torch.Tensor:
    """x [M, H] act dtype; topk_weights fp32 [M, K]; topk_ids int32 [M, K] (all valid) -> [M, H] act dtype."""
    load()
    out = torch.empty_like(x)
    torch.ops.lkm_ref.cpu_fused_moe(out, x.contiguous(), packed_w13, packed_w2, None, None,
                                    topk_weights.contiguous().float(), topk_ids.contiguous().int(), False, act, isa)
    return out


# ------------------------------------------------------------------ quantised experts (sgl-kernels MoE)
# csrc/cpu/sgl-kernels/gemm.h:92 -- enum class CPUQuantMethod
BF16, INT8_W8A8, FP8_W8A16, INT4_W4A8, MXFP4 = 0, 1, 2, 3, 4


def fused_experts_fp8_w8a16(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, w13_scale: torch.Tensor,
                            w2_scale: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                            block=(128, 128)) -> torch.Tensor:
    """The reference's fp8-W8A16 block-scaled CPU MoE (csrc/cpu/sgl-kernels/moe.cpp:874, FP8_W8A16; call site
    vllm/model_executor/layers/fused_moe/experts/cpu_moe.py:167-185): w13 [E,2I,H] / w2 [E,H,I] float8_e4m3fn,
    scales fp32 [E, rows/128, cols/128]; SiLU-gated; returns the activation dtype."""
    load()
    pw13 = torch.ops.lkm_ref.convert_weight_packed(w13.contiguous())
    pw2 = torch.ops.lkm_ref.convert_weight_packed(w2.contiguous())
    return torch.ops.lkm_ref.fused_experts_cpu(x.clone(), pw13, pw2, topk_weights.float().contiguous(),
                                               topk_ids.int().contiguous(), False, FP8_W8A16,
                                               w13_scale.float().contiguous(), w2_scale.float().contiguous(),
                                               None, None, list(block), None, None, None, None, True)


def fused_experts_mxfp4(x: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, w13_scale: torch.Tensor,
                        w2_scale: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
    """The reference's MXFP4 CPU MoE (same entry point, MXFP4; call site cpu_moe.py:321-339): w13 [E,2I,H/2] /
    w2 [E,H,I/2] uint8 (two E2M1 codes per byte, low nibble first), scales uint8 E8M0 [E, rows, cols/32]."""
    load()
    pw13 = torch.ops.lkm_ref.convert_weight_packed(w13.contiguous())
    pw2 = torch.ops.lkm_ref.convert_weight_packed(w2.contiguous())
    ps13 = torch.ops.lkm_ref.convert_scale_packed(w13_scale.contiguous())
    ps2 = torch.ops.lkm_ref.convert_scale_packed(w2_scale.contiguous())
    return torch.ops.lkm_ref.fused_experts_cpu(x.clone(), pw13, pw2, topk_weights.float().contiguous(),
                                               topk_ids.int().contiguous(), False, MXFP4, ps13, ps2,
                                               None, None, None, None, None, None, None, True)


GPTQ = 1    # csrc/cpu/sgl-kernels/gemm.h:102 -- enum class CPUQuantAlgo { AWQ = 0, GPTQ = 1 }


def fused_experts_int4_gptq(x: torch.Tensor, w13_packed: torch.Tensor, w2_packed: torch.Tensor, w13_scale: torch.Tensor,
                            w2_scale: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor) -> torch.Tensor:
    """The reference's int4 CPU MoE on the CHECKPOINT layout LvLLM hands to lk_moe before its transposition
    (routed_experts.py:1461-1479): w13_packed int32 [E, H/8, 2I] / w2_packed int32 [E, I/8, H], nibble j of a word
    = input channel 8*row + j, value + 8 (uint4b8, i.e. zero point 8); scales act dtype [E, groups, out].
    Weight preparation as vllm/model_executor/layers/fused_moe/experts/cpu_moe.py:347-405 (symmetric checkpoints get
    the synthetic zero points 0x77777777: the unpack adds 1).  NB the kernel computes W4A8 -- it quantises the
    activations to int8 per token -- so it pins the FORMAT (nibble order, zero point, group-scale layout), not the
    last bits of a W4A16 result."""
    load()
    E = w13_packed.size(0)
    z13 = torch.full((E, w13_scale.size(1), w13_scale.size(2) // 8), 0x77777777, dtype=torch.int32)
    z2 = torch.full((E, w2_scale.size(1), w2_scale.size(2) // 8), 0x77777777, dtype=torch.int32)
    bw13, bz13, bs13 = torch.ops.lkm_ref.convert_weight_packed_scale_zp(w13_packed.contiguous(), z13, w13_scale.contiguous(), GPTQ)
    bw2, bz2, bs2 = torch.ops.lkm_ref.convert_weight_packed_scale_zp(w2_packed.contiguous(), z2, w2_scale.contiguous(), GPTQ)
    return torch.ops.lkm_ref.fused_experts_cpu(x.clone(), bw13, bw2, topk_weights.float().contiguous(),
                                               topk_ids.int().contiguous(), False, INT4_W4A8, bs13, bs2, bz13, bz2,
                                               None, None, None, None, None, True)
