"""Spill tier on the GPU (lvllm_amd/spill.py; SURVEY 8 f4, round-4 verdict "missing" 3): experts parked in pinned host
memory and streamed through 2 x LVLLM_GPU_PREFETCH_WINDOW device slots give the resident engine's results -- against the
oracle at the operator's tolerance, and against the resident engine of the same weights (only the fp32 order of the
per-group partial sums differs)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ATOL, RTOL = 2e-3, 1e-2


def _case(M, E, K, H, I, seed, skew=0.0, drop=0.0):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed, skew=skew, drop=drop)
    return a, w13, w2, tw, ids


@pytest.mark.parametrize("M,E,K,window,skew", [(300, 10, 2, 2, 0.0), (64, 7, 3, 3, 1.0), (1, 5, 2, 1, 0.0), (2000, 9, 2, 2, 0.5)])
def test_spill_tier_bf16_matches_oracle_and_resident_engine(M, E, K, window, skew):
    from lvllm_amd import ops
    from lvllm_amd.spill import HostResidentExperts
    H, I = 256, 128
    a, w13, w2, tw, ids = _case(M, E, K, H, I, seed=3 + M, skew=skew, drop=0.05)
    sp = HostResidentExperts(w13, w2, top_k=K, act_dtype=torch.bfloat16, window=window)
    assert sp.slots == 2 * window and len(sp.images) == E and sp.host_bytes() == E * sp.expert_bytes
    ad, twd, idd = a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV)
    out = sp.decode(ad, twd, idd)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids, tw)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)
    hit = len({int(e) for e in ids.reshape(-1) if e >= 0})
    lp = sp.last_pass
    assert lp["experts_with_rows"] == hit and lp["groups"] == -(-hit // window) and lp["bytes_h2d"] <= hit * sp.expert_bytes
    res = ops.RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16)
    want = res.decode(ad, twd, idd)
    # (the two plan differently -- the spill pass sees ~window / E of the rows per launch -- so kernels and rounding points
    #  differ inside the operator's tolerance: e.g. the resident prefill-sized step rounds its GEMM2 partial rows to bf16)
    torch.testing.assert_close(out, want, atol=ATOL * float(want.abs().max()), rtol=RTOL)
    # the activation-dtype surface (gpu_prefill) and a second pass on new routing (slots are re-filled as needed)
    pre = sp.prefill(ad, twd, idd)
    assert pre.dtype == torch.bfloat16
    torch.testing.assert_close(pre.float(), out, atol=2.0 ** -7 * float(out.abs().max()), rtol=2.0 ** -7)
    tw2, ids2 = make_routing(M, E, K, seed=99 + M)
    out2 = sp.decode(ad, torch.from_numpy(tw2).to(DEV), torch.from_numpy(ids2).to(DEV))
    ref2 = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids2, tw2)
    np.testing.assert_allclose(out2.cpu().numpy(), ref2, atol=ATOL, rtol=RTOL)
    sp.close()


def test_spill_tier_int4_and_window_from_environment(monkeypatch):
    """the quantised formats' images carry their scales; the window defaults to LVLLM_GPU_PREFETCH_WINDOW"""
    from lvllm_amd.spill import HostResidentExperts
    monkeypatch.setenv("LVLLM_GPU_PREFETCH_WINDOW", "2")
    M, E, K, H, I, g = 150, 6, 2, 256, 128, 128
    a, w13, w2, tw, ids = _case(M, E, K, H, I, seed=21)
    q13, s13 = orc.quant_int4(torch_to_bits(w13), orc.BF16, g)
    q2, s2 = orc.quant_int4(torch_to_bits(w2), orc.BF16, g)
    sp = HostResidentExperts(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=torch.bfloat16, fmt="int4",
                             w13_scale=bits_to_torch(s13, orc.BF16), w2_scale=bits_to_torch(s2, orc.BF16), group_n=1, group_k=g)
    assert sp.window == 2 and sp.slots == 4
    out = sp.decode(a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV))
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=g)
    ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=ATOL, rtol=RTOL)
    with pytest.raises(ValueError):
        HostResidentExperts(w13.to(DEV), w2.to(DEV), top_k=K, act_dtype=torch.bfloat16)
    sp.close()
