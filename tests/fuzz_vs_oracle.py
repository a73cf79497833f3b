#!/usr/bin/env python3
"""GPU: seeded random sweep of whole layers against the CPU oracle, for a time budget -- every weight format (16-bit, uint4b8 at
three group sizes, uint4 with zero points, fp8 W8A16 / W8A8, MXFP4, NVFP4), gated / relu2, batch sizes from 0 to a few thousand
(decode, fp32 output, up to the engine's max_num_seqs; prefill, activation-dtype output, above it and as a second look at the
decode sizes), skewed routing and dropped slots, whatever launch plan the planner picks.  (It lives under tests/: only tests/, smoke() and the bench cpu_baseline leg may import the oracle.)  The -m gpu suite holds a 28-case
version of this (tests/test_gpu_moe.py::test_randomised_shapes_and_formats_vs_oracle); this is the long run.

    [FUZZ_EMAX=256] python tests/fuzz_vs_oracle.py [seconds=240] [seed=1]        # prints one line per case, a summary, exit code 1 on a mismatch
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench                                  # noqa: E402
from lvllm_amd import _clib                  # noqa: E402
from lvllm_amd.ops import RoutedExpertsEngine  # noqa: E402
from oracle import oracle as orc             # noqa: E402
from tests.helpers import bits_to_torch, make_routing, torch_to_bits  # noqa: E402

DEV = "cuda:0"
FORMATS = ["bf16", "f16", "int4", "int4zp", "fp8", "fp8a8", "mxfp4", "nvfp4"]


def build(fmt, rng, E, K, H, I, gated, seed, oai=False):
    """-> (engine, oracle closure(x_bits, ids, tw, prefill) -> fp32 reference, torch activation dtype, tolerances)"""
    dt = torch.float16 if fmt == "f16" or (fmt in ("int4", "int4zp") and rng.integers(0, 3) == 0) else torch.bfloat16
    odt = orc.F16 if dt == torch.float16 else orc.BF16
    g = torch.Generator().manual_seed(seed)
    halves = 2 if gated else 1
    w13 = (torch.randn((E, halves * I, H), generator=g) / 10).to(dt)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(dt)
    kw = dict(has_gate_proj=False, activation_type=2) if not gated else (dict(activation_type=1) if oai else {})
    dk = dict(E=E, H=H, I=I, has_gate=gated, activation=(orc.ACT_SWIGLUOAI if oai else orc.ACT_SILU) if gated else orc.ACT_RELU2, act_dtype=odt)
    tol = (2e-3, 1e-2)
    if fmt in ("bf16", "f16"):
        eng = RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=dt, max_num_seqs=256, **kw)
        d = orc.MoeDesc(wfmt=orc.W_F16 if dt == torch.float16 else orc.W_BF16, **dk)
        a13, a2 = torch_to_bits(w13), torch_to_bits(w2)
        ref = lambda x, ids, tw: orc.moe(d, a13, a2, x, ids, tw)                      # noqa: E731
    elif fmt in ("int4", "int4zp"):
        gk = int(rng.choice([32, 64, 128]))
        q13, s13 = orc.quant_int4(torch_to_bits(w13), odt, gk)
        q2, s2 = orc.quant_int4(torch_to_bits(w2), odt, gk)
        if fmt == "int4":
            eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=dt, fmt="int4",
                                      w13_scale=bits_to_torch(s13, odt), w2_scale=bits_to_torch(s2, odt), group_n=1, group_k=gk, **kw)
            d = orc.MoeDesc(wfmt=orc.W_INT4, groupN=1, groupK=gk, **dk)
            ref = lambda x, ids, tw: orc.moe(d, q13, q2, x, ids, tw, s13=s13, s2=s2)  # noqa: E731
        else:
            q13 = rng.integers(0, 256, q13.shape, dtype=np.uint8)                     # (asymmetric codes: any nibble against any zero point)
            q2 = rng.integers(0, 256, q2.shape, dtype=np.uint8)
            z13 = rng.integers(0, 16, s13.shape, dtype=np.uint8)
            z2 = rng.integers(0, 16, s2.shape, dtype=np.uint8)
            s13 = orc.f32_to_bits(rng.uniform(0.002, 0.012, s13.shape).astype(np.float32), odt)
            s2 = orc.f32_to_bits(rng.uniform(0.002, 0.012, s2.shape).astype(np.float32), odt)
            eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=dt, fmt="int4",
                                      w13_scale=bits_to_torch(s13, odt), w2_scale=bits_to_torch(s2, odt), group_n=1, group_k=gk,
                                      w13_zp=torch.from_numpy(z13), w2_zp=torch.from_numpy(z2), **kw)
            pk = lambda z: (z[:, 0::2] | (z[:, 1::2] << 4)).astype(np.uint8)          # noqa: E731
            d13 = orc.dequant_wna16(q13, s13, pk(z13), 4, gk, odt)
            d2 = orc.dequant_wna16(q2, s2, pk(z2), 4, gk, odt)
            d = orc.MoeDesc(wfmt=orc.W_F16 if dt == torch.float16 else orc.W_BF16, **dk)
            ref = lambda x, ids, tw: orc.moe(d, d13, d2, x, ids, tw)                  # noqa: E731
    elif fmt in ("fp8", "fp8a8"):
        q13, s13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
        q2, s2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
        a8 = fmt == "fp8a8"
        eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=dt, fmt="fp8",
                                  w13_scale=torch.from_numpy(s13), w2_scale=torch.from_numpy(s2), group_n=128, group_k=128,
                                  fp8_mode=_clib.FP8_W8A8 if a8 else _clib.FP8_W8A16, **kw)
        d = orc.MoeDesc(wfmt=orc.W_FP8, groupN=128, groupK=128, **(dict(round_gemm1=True, w8a8=True) if a8 else {}), **dk)
        ref = lambda x, ids, tw: orc.moe(d, q13, q2, x, ids, tw, s13=s13, s2=s2)      # noqa: E731
        if a8:
            # the reference's tolerance for block-fp8 W8A8 (tests/kernels/moe/test_block_fp8.py:143-210: 0.035): a GEMM1 sum that
            # rounds to the other bf16 neighbour flips the fp8 code of the quantised intermediate -- one fp8 ulp (6 %) of that
            # element; a sparse tail of ~1e-2 |ref|max under relu2, a third of what separates W8A8 from W8A16 on the same weights
            tol = (2e-2, 3.5e-2)
    elif fmt == "mxfp4":
        q13 = rng.integers(0, 256, (E, halves * I, H // 2), dtype=np.uint8)
        q2 = rng.integers(0, 256, (E, H, I // 2), dtype=np.uint8)
        s13 = rng.integers(117, 121, (E, halves * I, H // 32), dtype=np.uint8)
        s2 = rng.integers(117, 121, (E, H, I // 32), dtype=np.uint8)
        eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=dt, fmt="mxfp4",
                                  w13_scale=torch.from_numpy(s13), w2_scale=torch.from_numpy(s2), group_n=1, group_k=32, **kw)
        d = orc.MoeDesc(wfmt=orc.W_MXFP4, groupN=1, groupK=32, **dk)
        ref = lambda x, ids, tw: orc.moe(d, q13, q2, x, ids, tw, s13=s13, s2=s2)      # noqa: E731
    else:
        q13, s13, m13 = [t.cpu() for t in bench.quantize_nvfp4(w13.to(DEV))]
        q2, s2, m2 = [t.cpu() for t in bench.quantize_nvfp4(w2.to(DEV))]
        eng = RoutedExpertsEngine(q13, q2, top_k=K, act_dtype=dt, fmt="nvfp4", w13_scale=s13, w2_scale=s2, group_n=1, group_k=16,
                                  w13_global_scale=m13, w2_global_scale=m2, **kw)
        d = orc.MoeDesc(wfmt=orc.W_NVFP4, groupN=1, groupK=16, **dk)
        a = (q13.numpy(), q2.numpy(), s13.numpy(), s2.numpy(), m13.numpy(), m2.numpy())
        ref = lambda x, ids, tw: orc.moe(d, a[0], a[1], x, ids, tw, s13=a[2], s2=a[3], gs13=a[4], gs2=a[5])   # noqa: E731
    return eng, ref, dt, tol


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, n, bad = time.time(), 0, []
    while time.time() - t0 < budget:
        fmt = FORMATS[n % len(FORMATS)]
        gated = bool(rng.integers(0, 4))
        E = int(rng.integers(1, int(os.environ.get("FUZZ_EMAX", "40")) + 1))      # (FUZZ_EMAX=256: many small experts, expert-parallel-like sparsity)
        K = int(rng.integers(1, min(E, 8) + 1))
        H = int(rng.integers(1, 9)) * 128
        I = int(rng.integers(1, 7)) * 128
        if fmt in ("bf16", "f16") and rng.integers(0, 2):           # 16-bit weights: any multiple of 8 / 16 (models off the 128 grid)
            H, I = int(rng.integers(2, 130)) * 8, int(rng.integers(1, 50)) * 16
        oai = gated and bool(rng.integers(0, 4) == 0)               # gpt-oss' clamped, interleaved swiglu
        try:
            eng, ref, dt, (atol, rtol) = build(fmt, rng, E, K, H, I, gated, seed * 100000 + n, oai)
        except Exception as ex:                                     # a refused configuration is a finding too
            bad.append(f"case {n}: {fmt} E={E} K={K} H={H} I={I} gated={gated}: constructor raised {type(ex).__name__}: {ex}")
            print(bad[-1], flush=True)
            n += 1
            continue
        for M in [int(m) for m in rng.choice([0, 1, 2, 5, 17, 33, 64, 130, 256, 300, 700, 1500, 4000], size=3, replace=False)]:
            if M * H > 3_000_000:                                   # (keeps the CPU oracle in seconds)
                M = max(1, 3_000_000 // H)
            g = torch.Generator().manual_seed(n * 7 + M)
            x = (torch.randn((M, H), generator=g) / 10).to(dt)
            tw, ids = make_routing(M, E, K, seed=n + M, skew=float(rng.choice([0.0, 1.5, 3.0])), drop=float(rng.choice([0.0, 0.2, 0.9] if E > 40 else [0.0, 0.2])))
            want = ref(torch_to_bits(x), ids, tw) if M else np.zeros((0, H), np.float32)
            scale = max(1.0, float(np.abs(want).max())) if M else 1.0
            xd, twd, idd = x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
            outs = {}
            if M <= 256:
                outs["decode"] = (eng.decode(xd, twd, idd).cpu().numpy(), atol, rtol)
            outs["prefill"] = (eng.prefill(xd, twd, idd).float().cpu().numpy(), max(atol, 4e-3), max(rtol, 1.5e-2))
            for name, (out, a_, r_) in outs.items():
                err = np.abs(out - want)
                ok = bool((err <= a_ * scale + r_ * np.abs(want)).all()) and np.isfinite(out).all()
                line = (f"case {n}: {fmt} {str(dt)[6:]} E={E} K={K} H={H} I={I} gated={gated}{' oai' if oai else ''} M={M} {name}: "
                        f"max err {float(err.max()) if M else 0.0:.3e} (scale {scale:.2e}) {'ok' if ok else 'MISMATCH'} | {eng.engine.describe()[-90:]}")
                print(line, flush=True)
                if not ok:
                    bad.append(line)
        del eng
        n += 1
    print(f"== {n} engines in {time.time() - t0:.0f} s, {len(bad)} findings")
    for b in bad:
        print("  ", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
