"""Full-size parity for the BASELINE.json configurations that tests/test_gpu_moe.py does not already run at size
(configs[1], Mixtral bf16 M=32, lives there: test_mixtral_full_size_vs_oracle).  Every output row is compared
with the CPU oracle for configs[2] and configs[3]; for configs[4] (8192 tokens, 65 536 routed rows) the engine runs
the whole batch and a seeded 256-token subset is checked -- a token's result does not depend on the other tokens
(permutation equivariance is tested at full size in test_gpu_moe.py), so the subset is an unbiased sample of the
SAME launch: 256-row tiles, the multi-workgroup sort, the prefill kernels.

Weights are quantised on the GPU with bench.py's recipes (the bit-exactness of the quantisers themselves is
pinned elsewhere: tests/test_oracle_golden.py); the oracle consumes the same packed bytes and scales, so the
comparison isolates the expert arithmetic.  Tolerances as in test_gpu_moe.py: vs oracle atol 2e-3 x max|ref|,
rtol 1e-2; fp8-W8A8 atol 1e-2 x max|ref|, rtol 2e-2 (an fp8 rounding decision of the re-quantised intermediate can
flip on a last-bit difference of the bf16 value; the reference's own tolerance for this operator is 0.035)."""
import numpy as np
import pytest
import torch

import bench
from oracle import oracle as orc
from tests.helpers import make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _masters(E, H, I, seed):
    return bench.make_weights(E, seed, H, I, torch.device(DEV), "bf16")


def _np(t):
    return torch_to_bits(t) if t.dtype in (torch.bfloat16, torch.float16) else t.cpu().numpy()


def test_config2_mixtral_int4_g128_m128_full_size():
    """BASELINE.json configs[2]: Mixtral-8x7B uint4b8 g128 experts, decode batch 128 (tiled int4 kernels,
    896-tile weight panels), all 128 x 4096 outputs; routing ids from the GPU router equal the oracle's"""
    from lvllm_amd import ops
    E, K, H, I, M, g = 8, 2, 4096, 14336, 128, 128
    w13, w2 = _masters(E, H, I, 0)
    q13, s13 = bench.quantize_int4(w13, g)
    q2, s2 = bench.quantize_int4(w2, g)
    del w13, w2
    eng = ops.RoutedExpertsEngine(q13, q2, top_k=K, act_dtype=torch.bfloat16, fmt="int4", w13_scale=s13, w2_scale=s2,
                                  group_n=1, group_k=g)
    gen = torch.Generator().manual_seed(7)
    x = (torch.randn((M, H), generator=gen) / 10).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=gen)
    tw, ids = ops.topk_softmax(logits.to(DEV), K, True)
    otw, oids = orc.topk_softmax(logits.numpy(), K, renormalize=True)
    assert np.array_equal(ids.cpu().numpy(), oids), "routing ids differ from the oracle"
    assert np.array_equal(tw.cpu().numpy().view(np.uint32), otw.view(np.uint32))
    out = eng.decode(x.to(DEV), tw, ids).cpu().numpy()
    assert "tiled" in eng.engine.describe()
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=g)
    ref = orc.moe(d, _np(q13), _np(q2), torch_to_bits(x), oids, otw, s13=_np(s13), s2=_np(s2))
    np.testing.assert_allclose(out, ref, atol=2e-3 * float(np.abs(ref).max()), rtol=1e-2)


def _dsv3_rank_inputs(rank=3, ep=8):
    """BASELINE.json configs[3] as rank `rank` of 8 sees it: the global decode batch of 256 tokens routed by the
    DeepSeek-V3 router (sigmoid + bias, 8 groups, top-4 groups, top-8, renormalised, x2.5), ids of the other ranks'
    experts -> -1; records of all 256 tokens arrive (token-granular exchange, capacity = tokens per rank)."""
    E, K, H, M = 256, 8, 7168, 256
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn((M, H), generator=gen) / 10).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=gen)
    bias = torch.randn((E,), generator=gen) * 0.1
    tw, ids = orc.grouped_topk(logits.numpy(), K, 8, 4, bias=bias.numpy(), scoring=1, renormalize=True,
                               routed_scaling=2.5)
    per = E // ep
    groups = ids // per
    assert (np.array([len(set(r)) for r in groups]) <= 4).all(), "group-limited routing: a token visits <= 4 ranks"
    lids = np.where(groups == rank, ids - rank * per, -1).astype(np.int32)
    return x, logits, bias, tw, ids, lids


@pytest.mark.parametrize("mode", ["w8a16", "w8a8"])
def test_config3_dsv3_rank_slice_fp8_full_size(mode):
    """BASELINE.json configs[3], one EP rank at full size: 32 local fp8 experts of H=7168, I=2048, the 256-token
    global batch; GPU grouped top-k ids == oracle ids; every output row against the oracle"""
    from lvllm_amd import _clib, ops
    E_loc, K, H, I = 32, 8, 7168, 2048
    x, logits, bias, tw, ids, lids = _dsv3_rank_inputs()
    gw, gi = ops.grouped_topk(x.to(DEV), logits.to(DEV), K, True, 8, 4, "sigmoid", 2.5, bias.to(DEV))
    assert np.array_equal(gi.cpu().numpy(), ids), "grouped top-k ids differ from the oracle"
    assert np.array_equal(gw.cpu().numpy().view(np.uint32), tw.view(np.uint32))
    w13, w2 = _masters(E_loc, H, I, 96)
    q13, s13 = bench.quantize_fp8_block(w13)
    q2, s2 = bench.quantize_fp8_block(w2)
    del w13, w2
    a8 = mode == "w8a8"
    eng = ops.RoutedExpertsEngine(q13, q2, top_k=K, act_dtype=torch.bfloat16, fmt="fp8", w13_scale=s13, w2_scale=s2,
                                  group_n=128, group_k=128, fp8_mode=_clib.FP8_W8A8 if a8 else _clib.FP8_W8A16)
    eng.engine.set_tuning(valid_den=8)
    out = eng.decode(x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(lids).to(DEV)).cpu().numpy()
    d = orc.MoeDesc(E=E_loc, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128,
                    round_gemm1=a8, w8a8=a8)
    ref = orc.moe(d, _np(q13), _np(q2), torch_to_bits(x), lids, tw, s13=_np(s13), s2=_np(s2))
    scale = float(np.abs(ref).max())
    assert scale > 0 and (np.abs(ref).max(axis=1) > 0).sum() >= 100        # most tokens have an expert on this rank
    if a8:
        np.testing.assert_allclose(out, ref, atol=1e-2 * scale, rtol=2e-2)
    else:
        np.testing.assert_allclose(out, ref, atol=2e-3 * scale, rtol=1e-2)
    no_local = (lids < 0).all(axis=1)
    assert (out[no_local] == 0).all()                                      # rows without a local expert: exact zeros


@pytest.mark.parametrize("mode", ["w8a8", "w8a16", "bf16"])
def test_config4_glm45air_prefill_m8192_full_size(mode):
    """BASELINE.json configs[4]: GLM-4.5-Air shapes, prefill of 8192 tokens (65 536 routed rows, 128 experts);
    the whole batch runs, a seeded 256-token subset is compared with the oracle"""
    from lvllm_amd import _clib, ops
    E, K, H, I, M = 128, 8, 4096, 1408, 8192
    w13, w2 = _masters(E, H, I, 200)
    kw = {}
    if mode == "bf16":
        eng = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16, max_batch_size=M)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        oa = (torch_to_bits(w13), torch_to_bits(w2))
    else:
        q13, s13 = bench.quantize_fp8_block(w13)
        q2, s2 = bench.quantize_fp8_block(w2)
        a8 = mode == "w8a8"
        eng = ops.RoutedExpertsEngine(q13, q2, top_k=K, act_dtype=torch.bfloat16, fmt="fp8", w13_scale=s13,
                                      w2_scale=s2, group_n=128, group_k=128, max_batch_size=M,
                                      fp8_mode=_clib.FP8_W8A8 if a8 else _clib.FP8_W8A16)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128, round_gemm1=a8,
                        w8a8=a8)
        oa = (_np(q13), _np(q2))
        kw = dict(s13=_np(s13), s2=_np(s2))
    del w13, w2
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn((M, H), generator=gen) / 10).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed=6)
    xd, twd, idd = x.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
    y_dev = eng.prefill(xd, twd, idd).clone()
    for _ in range(6):          # the full-size step again and again: bit-identical (every kernel orders its DMA by counted waits only)
        assert torch.equal(eng.prefill(xd, twd, idd), y_dev), eng.engine.describe()
    y = y_dev.float().cpu().numpy()
    assert np.isfinite(y).all()
    sub = np.sort(np.random.default_rng(9).choice(M, 256, replace=False))
    ref = orc.moe(d, *oa, torch_to_bits(x[sub]), ids[sub], tw[sub], **kw)
    scale = float(np.abs(ref).max())
    # + one bf16 rounding of the output (gpu_prefill writes the activation dtype)
    if mode == "w8a8":
        np.testing.assert_allclose(y[sub], ref, atol=1e-2 * scale, rtol=2e-2)
    else:
        np.testing.assert_allclose(y[sub], ref, atol=3e-3 * scale, rtol=1.5e-2)
