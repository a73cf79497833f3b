"""uint4 experts WITH ZERO POINTS on the engine's native 4-bit path (LkmConfig.int4_mode = LKM_INT4_ZP, include/lkm.h): the
in-tree operator's asymmetric dequantisation `((b - zp) * scale).to(compute_type)` (fused_moe.py:207-208,237-238,272-276),
decoded in registers from the packed image -- fma(v * 2^-9, 512 s, -zp s), one rounding.  Against the oracle's
dequant_wna16 (pinned to the reference's quantize_weights w_ref by tests/golden/moe_wna16.npz, tests/test_oracle_golden.py):
the dequantisation read back through one-hot activations BIT FOR BIT for every group size and both activation dtypes, on the
streamer (few rows per expert) and on the tile kernels; whole layers at decode and prefill sizes within the suite's tolerance;
the same bits as the expanded 16-bit engine modular.py used until round 6."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL, RTOL = 2e-3, 1e-2


def _eng(*a, **k):
    from lvllm_amd.ops import RoutedExpertsEngine
    return RoutedExpertsEngine(*a, **k)


def _pack_zp(z: np.ndarray) -> np.ndarray:
    """uint8 [E, R, G] -> the reference's packed layout [E, R / 2, G] (low nibble = even row), what dequant_wna16 takes"""
    return (z[:, 0::2] | (z[:, 1::2] << 4)).astype(np.uint8)


def _case(rng, E, N, K, g, odt, wide=True):
    q = rng.integers(0, 256, (E, N, K // 2), dtype=np.uint8)
    s = rng.uniform(0.004, 0.03, (E, N, K // g))
    if wide:                                               # (scales over two decades: the dequantisation test; layers keep fp16 finite)
        s = s * rng.choice([1.0, 37.0, 0.25], (E, N, K // g))
    s = s.astype(np.float32)
    z = rng.integers(0, 16, (E, N, K // g), dtype=np.uint8)
    return q, orc.f32_to_bits(s, odt), z


@pytest.mark.parametrize("g", [32, 64, 128, 256])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_zero_point_dequant_is_bit_exact(g, dt):
    E, H, I, K = 2, 256, 128, 1
    odt, tdt = (orc.BF16, torch.bfloat16) if dt == "bf16" else (orc.F16, torch.float16)
    rng = np.random.default_rng(41 + g)
    q13, s13b, z13 = _case(rng, E, I, H, g, odt)
    q13[0, 0, :8] = np.arange(0, 256, 32, dtype=np.uint8) + np.arange(8, dtype=np.uint8)      # all 16 codes in one row
    z13[0, 0, :] = np.array([0, 15, 7, 8, 1, 14, 3, 12])[: H // g]                            # ... against the extreme zero points
    wd = orc.bits_to_f32(orc.dequant_wna16(q13, s13b, _pack_zp(z13), 4, g, odt), odt)          # [E, I, H]
    g2 = min(g, I)
    q2 = np.full((E, H, I // 2), 0x88, np.uint8)                                              # (8 - 8) * 1 = 0 ...
    for r in range(min(H, I)):
        q2[:, r, r // 2] = 0x88 + (1 << (4 * (r & 1)))                                        # ... and (9 - 8) * 1 = 1 on the diagonal
    s2b = orc.f32_to_bits(np.ones((E, H, I // g2), np.float32), odt)
    z2 = np.full((E, H, I // g2), 8, np.uint8)
    eng = _eng(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=tdt, fmt="int4",
               w13_scale=bits_to_torch(s13b, odt), w2_scale=bits_to_torch(s2b, odt), group_n=1, group_k=g,
               has_gate_proj=False, activation_type=2, w13_zp=torch.from_numpy(z13), w2_zp=torch.from_numpy(z2))
    assert "wf=3 zp=1" in eng.engine.describe()
    x = torch.eye(H, dtype=tdt)
    for e in range(E):
        ids = np.full((H, 1), e, np.int32)
        tw = np.ones((H, 1), np.float32)
        # the planner's own choice, every tile height of the tile kernel ("pf" = -1), the 32 x 32 MFMA kernels (gemm_w4x.h 5,
        # gemm_w4e.h 6 with four and seven consumers; not at 32-k groups: their scale area holds two pairs per unit)
        plans = [(0, 0, 0), (-1, 32, 0), (-1, 64, 0), (-1, 128, 0)] + ([(5, 32, 0), (5, 64, 0), (6, 32, 0), (6, 64, 0), (6, 64, 7)] if g >= 64 else [])
        for pf, tiled, waves in plans:
            eng.engine.set_tuning(pf=pf, tiled=tiled, waves=waves, pd1=2 if waves == 7 else 0)
            for sign in (1.0, -1.0):
                out = eng.decode((x * sign).to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)).cpu().numpy()[:, :I]
                want = np.maximum(sign * wd[e].T, 0.0) ** 2                                   # out[j, i] = T(relu(+-W[e, i, j])^2)
                want = orc.bits_to_f32(orc.f32_to_bits(want.astype(np.float32), odt), odt)
                np.testing.assert_array_equal(out, want, err_msg=f"g={g} e={e} pf={pf} tiled={tiled} waves={waves} sign={sign} {eng.engine.describe()}")
        # few rows per expert: the streamer
        eng.engine.set_tuning(pf=0, tiled=0, waves=0, pd1=0)
        rows = np.array([0, 1, 7, H - 1])
        out = eng.decode(x[rows].to(DEV), torch.ones((4, 1), dtype=torch.float32, device=DEV),
                         torch.full((4, 1), e, dtype=torch.int32, device=DEV)).cpu().numpy()[:, :I]
        want = np.maximum(wd[e].T[rows], 0.0) ** 2
        np.testing.assert_array_equal(out, orc.bits_to_f32(orc.f32_to_bits(want.astype(np.float32), odt), odt), err_msg=f"streamer g={g} e={e}")


@pytest.mark.parametrize("M,E,K,H,I,g,dt,gated,prefill", [
    (1, 8, 2, 512, 256, 128, "bf16", True, False),       # one token: the streamer
    (32, 8, 2, 512, 384, 128, "bf16", True, False),      # decode batch
    (128, 8, 2, 512, 384, 64, "f16", True, False),       # ~32 rows per expert: tile kernels, two scale groups per unit
    (77, 4, 2, 256, 128, 32, "bf16", True, False),       # ragged, four groups per unit
    (33, 16, 4, 256, 128, 128, "bf16", False, False),    # relu2, non-gated
    (600, 4, 2, 256, 256, 256, "bf16", True, True),      # prefill-sized, one group per two units, activation-dtype output
    (2000, 6, 2, 384, 256, 128, "f16", True, True),      # ~670 rows per expert: the tall tiles
])
def test_zero_point_layers_vs_oracle(M, E, K, H, I, g, dt, gated, prefill):
    odt, tdt = (orc.BF16, torch.bfloat16) if dt == "bf16" else (orc.F16, torch.float16)
    rng = np.random.default_rng(7 + M)
    halves = 2 if gated else 1
    q13, s13b, z13 = _case(rng, E, halves * I, H, g, odt, wide=False)
    g2 = min(g, I)
    q2, s2b, z2 = _case(rng, E, H, I, g2, odt, wide=False)
    kw = {} if gated else dict(has_gate_proj=False, activation_type=2)
    if g != g2:
        pytest.skip("one group size for both GEMMs")
    eng = _eng(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=tdt, fmt="int4",
               w13_scale=bits_to_torch(s13b, odt), w2_scale=bits_to_torch(s2b, odt), group_n=1, group_k=g,
               w13_zp=torch.from_numpy(z13), w2_zp=torch.from_numpy(z2), **kw)
    d13 = orc.dequant_wna16(q13, s13b, _pack_zp(z13), 4, g, odt)
    d2 = orc.dequant_wna16(q2, s2b, _pack_zp(z2), 4, g2, odt)
    desc = orc.MoeDesc(E=E, H=H, I=I, act_dtype=odt, wfmt=orc.W_BF16 if dt == "bf16" else orc.W_F16, **({} if gated else dict(has_gate=False, activation=orc.ACT_RELU2)))
    gen = torch.Generator().manual_seed(M)
    a = (torch.randn((M, H), generator=gen) / 10).to(tdt)
    tw, ids = make_routing(M, E, K, seed=M, drop=0.05)
    ref = orc.moe(desc, d13, d2, torch_to_bits(a), ids, tw)
    assert np.abs(ref).max() > 0
    twd, idd = torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
    scale = float(np.abs(ref).max())
    if prefill:
        out = eng.prefill(a.to(DEV), twd, idd).float().cpu().numpy()
        np.testing.assert_allclose(out, ref, atol=4e-3 * scale, rtol=1.5e-2, err_msg=eng.engine.describe())
        if H % 128 == 0 and I % 128 == 0 and g >= 64:
            # ... and on the 256-row prefill kernel (gemm_prefill.h: the weights decoded ONCE per workgroup into a 16-bit image,
            # with the same decoder), whatever the planner chose above
            eng.engine.set_tuning(tiled=256, waves=8, pf=8)
            o8 = eng.prefill(a.to(DEV), twd, idd).float().cpu().numpy()
            assert "pf=8" in eng.engine.describe() and "tm=256" in eng.engine.describe(), eng.engine.describe()
            np.testing.assert_allclose(o8, ref, atol=4e-3 * scale, rtol=1.5e-2, err_msg=eng.engine.describe())
            eng.engine.set_tuning(tiled=0, waves=0, pf=0)
    else:
        out = eng.decode(a.to(DEV), twd, idd).cpu().numpy()
        np.testing.assert_allclose(out, ref, atol=ATOL * scale, rtol=RTOL, err_msg=eng.engine.describe())
    # the same numbers as the 16-bit engine on the expanded weights (what modular.py built until round 6): same dequantised
    # operands, fp32 accumulation in another order
    e16 = _eng(bits_to_torch(d13, odt), bits_to_torch(d2, odt), top_k=K, act_dtype=tdt, fmt="bf16" if dt == "bf16" else "fp16", **kw)
    o16 = (e16.prefill(a.to(DEV), twd, idd).float() if prefill else e16.decode(a.to(DEV), twd, idd)).cpu().numpy()
    np.testing.assert_allclose(out, o16, atol=(4e-3 if prefill else 1e-3) * scale, rtol=1.5e-2 if prefill else 5e-3)


def test_zero_point_arguments_are_checked():
    E, H, I, g = 2, 256, 128, 128
    rng = np.random.default_rng(3)
    q13, s13b, z13 = _case(rng, E, 2 * I, H, g, orc.BF16)
    q2, s2b, z2 = _case(rng, E, H, I, g, orc.BF16)
    kw = dict(top_k=2, act_dtype=torch.bfloat16, fmt="int4", w13_scale=bits_to_torch(s13b, orc.BF16),
              w2_scale=bits_to_torch(s2b, orc.BF16), group_n=1, group_k=g)
    with pytest.raises(ValueError):
        _eng(torch.from_numpy(q13), torch.from_numpy(q2), w13_zp=torch.from_numpy(z13), **kw)                    # half a pair
    with pytest.raises(ValueError):
        _eng(torch.from_numpy(q13), torch.from_numpy(q2), w13_zp=torch.from_numpy(z13[:, :4]), w2_zp=torch.from_numpy(z2), **kw)
    with pytest.raises(ValueError):
        _eng(torch.from_numpy(q13), torch.from_numpy(q2), int4_mode=2, **kw)                                     # the mode without zero points
    with pytest.raises(ValueError):
        _eng(torch.from_numpy(q13), torch.from_numpy(q2), int4_mode=1, w13_zp=torch.from_numpy(z13), w2_zp=torch.from_numpy(z2), **kw)
