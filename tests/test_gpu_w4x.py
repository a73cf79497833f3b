"""GPU parity of the round-4 4-bit decode kernels (lvllm_amd/csrc/gemm_w4x.h: 32x32x16 MFMA, a lane decodes one
weight row, tuning key "pf" = 5; gemm_w4e.h: the same with a loader wave and an LDS-DMA ring, "pf" = 6) against the CPU
oracle -- never against the other HIP kernels alone.

  * dequantisation bit for bit: one-hot token rows read every T((q - 8) * s) back through a relu2 expert
    (fused_moe.py:237-276), all 16 codes, groups 32 / 64 / 128, bf16 and fp16, both decoders (dbg 0 / 1);
  * whole layers vs the oracle: ragged token counts, experts with 0 / few / > 64 rows (several token tiles), dropped
    slots (-1), gated and relu2, every launch variant (both kernels, 32- and 64-row tiles, both decoders, split K).
Tolerance vs the oracle as tests/test_gpu_moe.py: atol 2e-3, rtol 1e-2 (fp32 summation order differs).
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL, RTOL = 2e-3, 1e-2
# (pf, tile rows, waves, ring depth, decoder)
VARIANTS = [(5, tiled, 4, 2, dbg) for tiled in (32, 64) for dbg in (0, 1)] + \
           [(6, tiled, 0, 3, dbg) for tiled in (32, 64) for dbg in (0, 1)]
# round 6: gemm_w4e.h with a two-slot ring ("pd" = 2; the default of GEMM1 at 64-row tiles), with seven consumers per workgroup
# ("waves" = 7: GEMM1's default where the row groups come in sevens; here also on group counts that do not) and gemm_w4s.h (uint4b8, "pf" = 7:
# weight register ring 6 (token ring 3) / 4 (token ring 4), 4 or 7 consumer waves)
VARIANTS_INT4 = VARIANTS + [(6, 64, 0, 2, 0), (6, 64, 0, 2, 1), (6, 64, 7, 2, 0), (7, 32, 0, 6, 0), (7, 64, 0, 6, 0), (7, 64, 0, 4, 0), (7, 32, 0, 4, 0),
                            (7, 64, 7, 6, 0), (7, 64, 7, 4, 0),
                            (6, 64, 7, 32, 0),      # split rings: three weight slots, two token slots (gemm_w4e_kernel: SX)
                            (6, 64, 7, 34, 0),      # the consumers fetch the next unit's weights into registers (gemm_w4e_kernel: PW)
                            (6, 64, 7, 36, 0), (6, 64, 7, 37, 0)]   # two row groups per consumer against one set of token fragments (R), ring 2 / 3


def _eng(*a, **k):
    from lvllm_amd.ops import RoutedExpertsEngine
    return RoutedExpertsEngine(*a, **k)


def _run_decode(eng, a, tw, ids):
    return eng.decode(a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)).cpu().numpy()


def _set(eng, pf, tiled, waves, pd, dbg, sk2=0):
    eng.engine.set_tuning(pf=pf, tiled=tiled, waves=waves, pd1=pd, pd2=pd, dbg=dbg, sk2=sk2)


def _reset(eng):
    eng.engine.set_tuning(pf=0, tiled=0, waves=0, pd1=0, pd2=0, dbg=0, sk2=0)


@pytest.mark.parametrize("g", [32, 64, 128])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_w4x_int4_dequant_is_bit_exact(g, dt):
    E, H, I, K = 2, 256, 128, 1
    odt, tdt = (orc.BF16, torch.bfloat16) if dt == "bf16" else (orc.F16, torch.float16)
    rng = np.random.default_rng(11 + g)
    q13 = rng.integers(0, 256, (E, I, H // 2), dtype=np.uint8)
    q13[0, 0, :8] = np.arange(0, 256, 32, dtype=np.uint8) + np.arange(8, dtype=np.uint8)   # all 16 codes
    s13 = (rng.uniform(0.004, 0.03, (E, I, H // g)) * rng.choice([1.0, 37.0, 0.25], (E, I, H // g))).astype(np.float32)
    s13b = orc.f32_to_bits(s13, odt)
    wd = orc.bits_to_f32(orc.dequant_rows(orc.W_INT4, odt, q13, s13b, H, g), odt)      # [E, I, H]
    q2 = np.full((E, H, I // 2), 0x88, np.uint8)                                       # zeros ...
    for r in range(min(H, I)):
        q2[:, r, r // 2] = 0x88 + (1 << (4 * (r & 1)))                                 # ... and 1.0 on the diagonal
    s2b = orc.f32_to_bits(np.ones((E, H, max(1, I // g)), np.float32), odt)
    eng = _eng(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=tdt, fmt="int4",
               w13_scale=bits_to_torch(s13b, odt), w2_scale=bits_to_torch(s2b, odt), group_n=1, group_k=g,
               has_gate_proj=False, activation_type=2)
    x = torch.eye(H, dtype=tdt)
    for e in range(E):
        ids = np.full((H, 1), e, np.int32)
        tw = np.ones((H, 1), np.float32)
        for (pf, tiled, waves, pd, dbg) in VARIANTS_INT4:
            _set(eng, pf, tiled, waves, pd, dbg)
            for sign in (1.0, -1.0):
                out = _run_decode(eng, x * sign, tw, ids)[:, :I]                       # out[j, i] = T(relu(+-W[e,i,j])^2)
                want = np.maximum(sign * wd[e].T, 0.0) ** 2
                want = orc.bits_to_f32(orc.f32_to_bits(want.astype(np.float32), odt), odt)
                np.testing.assert_array_equal(out, want, err_msg=f"g={g} e={e} {pf}/{tiled}/{waves}/{pd}/{dbg} sign={sign}")
    _reset(eng)


def _int4_case(M, E, K, H, I, g, dt, seed, gated=True, drop=0.0, skew=0.0):
    odt, tdt = (orc.BF16, torch.bfloat16) if dt == "bf16" else (orc.F16, torch.float16)
    gen = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, H), generator=gen) / 10).to(tdt)
    w13 = (torch.randn((E, (2 if gated else 1) * I, H), generator=gen) / 10).to(tdt)
    w2 = (torch.randn((E, H, I), generator=gen) / 10).to(tdt)
    tw, ids = make_routing(M, E, K, seed, skew=skew, drop=drop)
    q13, s13 = orc.quant_int4(torch_to_bits(w13), odt, g)
    q2, s2 = orc.quant_int4(torch_to_bits(w2), odt, g)
    kw = {} if gated else dict(has_gate_proj=False, activation_type=2)
    eng = _eng(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=tdt, fmt="int4",
               w13_scale=bits_to_torch(s13, odt), w2_scale=bits_to_torch(s2, odt), group_n=1, group_k=g, **kw)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=odt, wfmt=orc.W_INT4, groupN=1, groupK=g,
                    **({} if gated else dict(has_gate=False, activation=orc.ACT_RELU2)))
    ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
    return eng, a, tw, ids, ref


@pytest.mark.parametrize("M,E,K,H,I,g,dt,gated,drop,skew", [
    (128, 8, 2, 512, 384, 128, "bf16", True, 0.0, 0.0),      # ~32 rows per expert (configs[2] in small)
    (77, 4, 2, 256, 128, 64, "bf16", True, 0.1, 0.0),        # ragged, dropped slots
    (200, 3, 2, 384, 256, 32, "f16", True, 0.0, 1.5),        # skew: one expert with > 64 rows (several token tiles)
    (33, 16, 4, 256, 128, 128, "bf16", False, 0.0, 0.0),     # relu2, non-gated: two consecutive tiles per wave
    (9, 6, 2, 128, 640, 128, "f16", True, 0.0, 0.0),         # K loop of GEMM2 = 5 units (odd, shorter than a deep ring + 2)
    (70, 2, 1, 128, 128, 128, "bf16", True, 0.0, 0.0),       # one K unit in GEMM1
])
def test_w4x_layers_vs_oracle(M, E, K, H, I, g, dt, gated, drop, skew):
    eng, a, tw, ids, ref = _int4_case(M, E, K, H, I, g, dt, seed=5 + M, gated=gated, drop=drop, skew=skew)
    assert np.abs(ref).max() > 0
    for (pf, tiled, waves, pd, dbg) in VARIANTS_INT4:
        for sk2 in ((0, 2) if I >= 512 else (0,)):
            _set(eng, pf, tiled, waves, pd, dbg, sk2)
            out = _run_decode(eng, a, tw, ids)
            assert f"pf={pf}" in eng.engine.describe(), eng.engine.describe()          # the plan that ran
            np.testing.assert_allclose(out, ref, atol=ATOL, rtol=RTOL, err_msg=f"{pf}/{tiled}/{waves}/{pd}/{dbg}/sk{sk2} {eng.engine.describe()}")
    _reset(eng)
    base = _run_decode(eng, a, tw, ids)                      # the default plan on the same inputs
    np.testing.assert_allclose(base, ref, atol=ATOL, rtol=RTOL)


@pytest.mark.parametrize("M,gated,dt", [(128, True, "bf16"), (70, False, "f16"), (300, True, "bf16")])
def test_w4e_two_sets_per_workgroup_vs_oracle(M, gated, dt):
    """round 6: gemm_w4e.h walking TWO sets of seven row groups per workgroup as one stream ("kw1" = 2 with seven consumers):
    row-group counts that are multiples of 14 (I = 896: 56 pairs gated, 28 non-gated), ~32 rows per expert and skewed
    multi-tile experts, against the oracle and against the one-set plan"""
    E, K, H, I = 4, 2, 512, 896
    eng, a, tw, ids, ref = _int4_case(M, E, K, H, I, 128, dt, seed=21 + M, gated=gated, skew=0.0 if M < 200 else 1.0)
    eng.engine.set_tuning(pf=6, tiled=64, waves=7, pd1=2, pd2=3, kw1=2)
    out = _run_decode(eng, a, tw, ids)
    assert "pf=6" in eng.engine.describe() and "waves=7" in eng.engine.describe(), eng.engine.describe()
    k1 = eng.engine.last_kernels()["gemm1"]
    # template arguments: WF, ADT, CB, NC, GATED, IS_G1, S, DECV, G, SX, PW -> G = 2
    assert any("gemm_w4e_kernel" in k and [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")][8] == "2" for k in k1), k1
    np.testing.assert_allclose(out, ref, atol=ATOL, rtol=RTOL, err_msg=eng.engine.describe())
    eng.engine.set_tuning(pf=6, tiled=64, waves=7, pd1=2, pd2=3, kw1=0)
    one = _run_decode(eng, a, tw, ids)
    np.testing.assert_array_equal(out, one)        # the same arithmetic in the same order, only the workgroup boundaries moved
    eng.engine.set_tuning(pf=0, tiled=0, waves=0, pd1=0, pd2=0, kw1=0)


@pytest.mark.parametrize("fmt", ["mxfp4", "nvfp4"])
@pytest.mark.parametrize("M,dt", [(128, "bf16"), (45, "f16")])
def test_w4x_fp4_formats_vs_oracle(fmt, M, dt):
    """the E2M1 formats on the same two kernels (the decoders are gemm_skinny.h's Dec<MXFP4 / NVFP4>, bit-exact by
    tests/test_gpu_moe.py::test_fp4_dequant_is_bit_exact_on_gpu): E8M0 scales per 32 k / e4m3 scales per 16 k + per-expert
    multipliers, against the oracle's dequantised-weight path"""
    import bench
    E, K, H, I = 6, 2, 512, 256
    odt, tdt = (orc.BF16, torch.bfloat16) if dt == "bf16" else (orc.F16, torch.float16)
    gen = torch.Generator().manual_seed(17 + M)
    a = (torch.randn((M, H), generator=gen) / 10).to(tdt)
    w13 = (torch.randn((E, 2 * I, H), generator=gen) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=gen) / 10).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, 3 + M, drop=0.05)
    kw, okw = {}, {}
    if fmt == "mxfp4":
        (q13, s13), (q2, s2) = bench.quantize_mxfp4(w13.to(DEV)), bench.quantize_mxfp4(w2.to(DEV))
        wfmt, gk = orc.W_MXFP4, 32
    else:
        (q13, s13, m13), (q2, s2, m2) = bench.quantize_nvfp4(w13.to(DEV)), bench.quantize_nvfp4(w2.to(DEV))
        wfmt, gk = orc.W_NVFP4, 16
        kw = dict(w13_global_scale=m13, w2_global_scale=m2)
        okw = dict(gs13=m13.cpu().numpy(), gs2=m2.cpu().numpy())
    eng = _eng(q13, q2, top_k=K, act_dtype=tdt, fmt=fmt, w13_scale=s13, w2_scale=s2, group_n=1, group_k=gk, **kw)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=odt, wfmt=wfmt, groupN=1, groupK=gk)
    ref = orc.moe(d, q13.cpu().numpy(), q2.cpu().numpy(), torch_to_bits(a), ids, tw, s13=s13.cpu().numpy(), s2=s2.cpu().numpy(), **okw)
    for (pf, tiled, waves, pd, dbg) in VARIANTS:
        if dbg:
            continue                                           # (the second decoder is uint4b8's)
        _set(eng, pf, tiled, waves, pd, 0)
        out = _run_decode(eng, a, tw, ids)
        assert f"pf={pf}" in eng.engine.describe(), eng.engine.describe()
        np.testing.assert_allclose(out, ref, atol=ATOL, rtol=RTOL, err_msg=f"{fmt} {pf}/{tiled}")
    _reset(eng)


@pytest.mark.parametrize("pf", [5, 6])
def test_w4x_knob_survives_prefill_sized_chunks(pf):
    """with "pf" = 5 / 6 forced and nothing else, a prefill-sized chunk used to plan 256-row tiles
    (which exist on gemm_prefill.h only, "pf" 0 / 8) and fail with "variant tm=256 ... not built".  The planner now keeps
    the 64 / 128-row tiles there; decode-sized calls of the same engine still take the forced kernel."""
    M, E, K, H, I, g = 1536, 4, 2, 256, 128, 128         # 768 rows per expert: > the 112-row threshold of the 256-row plan
    eng, a, tw, ids, ref = _int4_case(M, E, K, H, I, g, "bf16", seed=23)
    eng.engine.set_tuning(pf=pf)
    out = _run_decode(eng, a, tw, ids)
    assert "tm=256" not in eng.engine.describe(), eng.engine.describe()
    np.testing.assert_allclose(out, ref, atol=ATOL, rtol=RTOL, err_msg=eng.engine.describe())
    small = _run_decode(eng, a[:96], tw[:96], ids[:96])
    assert f"pf={pf}" in eng.engine.describe(), eng.engine.describe()
    np.testing.assert_allclose(small, ref[:96], atol=ATOL, rtol=RTOL)
    _reset(eng)


def test_last_kernels_names_what_was_launched():
    """lkm_last_kernels (include/lkm.h): the kernel names bench.py's roofline and the FETCH passes use come from the launch
    itself -- the default tile kernel, then the forced 32x32-MFMA kernel, then the streamer, on one engine"""
    eng, a, tw, ids, ref = _int4_case(128, 8, 2, 512, 384, 128, "bf16", seed=133)
    _run_decode(eng, a, tw, ids)
    kd = eng.engine.last_kernels()                # round 5 default for uint4b8 on 64-row tiles: the loader-wave kernel
    assert len(kd["gemm1"]) == 1 and "gemm_w4e_kernel<" in kd["gemm1"][0] and "gemm_w4e_kernel<" in kd["gemm2"][0], kd
    eng.engine.set_tuning(pf=-1)
    _run_decode(eng, a, tw, ids)
    k0 = eng.engine.last_kernels()
    assert len(k0["gemm1"]) == 1 and "gemm_tiled_kernel<" in k0["gemm1"][0] and "gemm_tiled_kernel<" in k0["gemm2"][0], k0
    eng.engine.set_tuning(pf=5, tiled=64)
    _run_decode(eng, a, tw, ids)
    k1 = eng.engine.last_kernels()
    assert "gemm_w4x_kernel<" in k1["gemm1"][0] and "gemm_w4x_kernel<" in k1["gemm2"][0], k1
    eng.engine.set_tuning(pf=0, tiled=-1)
    _run_decode(eng, a, tw, ids)
    k2 = eng.engine.last_kernels()
    assert "gemm1_act_kernel<" in k2["gemm1"][0] and "gemm2_kernel<" in k2["gemm2"][0], k2
    for k in (kd, k0, k1, k2):                  # as rocprofv3 prints them: namespace, template arguments, no parameter list
        assert all(n.startswith("lkm::") and n.endswith(">") and "(" not in n for n in k["gemm1"] + k["gemm2"]), k
    _reset(eng)
