"""GPU: lvllm_amd/layer.py on the real kernels -- router (+ owned gate) -> EPLB id map -> shared slots -> engine ->
post-processing, against the same chain evaluated with the CPU oracle."""
import numpy as np
import pytest
import torch

from lvllm_amd import eplb
from lvllm_amd import shared_experts as se
from lvllm_amd.layer import RoutedExpertsLayer, RoutingConfig
from oracle import oracle as orc
from tests.helpers import torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
E, K, H, I = 16, 4, 512, 256


def _weights(n, seed):
    g = torch.Generator().manual_seed(seed)
    return ((torch.randn((n, 2 * I, H), generator=g) / 4).to(torch.bfloat16),
            (torch.randn((n, H, I), generator=g) / 4).to(torch.bfloat16))


def _engine(w13, w2, top_k, max_num_seqs=64):
    from lvllm_amd.ops import RoutedExpertsEngine
    return RoutedExpertsEngine(w13.to(DEV), w2.to(DEV), top_k=top_k, act_dtype=torch.bfloat16, max_num_seqs=max_num_seqs)


def _moe(w13, w2, x, ids, tw):
    d = orc.MoeDesc(E=w13.shape[0], H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    return orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids, tw)


@pytest.mark.parametrize("M", [1, 40, 200])                     # decode convention up to 64 tokens, prefill above
def test_layer_softmax_routing_scaling_and_both_output_conventions(M):
    w13, w2 = _weights(E, 1)
    g = torch.Generator().manual_seed(M)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=g)
    layer = RoutedExpertsLayer(_engine(w13, w2, K), RoutingConfig(K, E, routed_scaling_factor=2.5), check_nan_in_output=True)
    out = layer.forward(x.to(DEV), logits.to(DEV))
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == (M, H)
    tw, ids = orc.topk_softmax(logits.numpy(), K)
    ref = 2.5 * _moe(w13, w2, x, ids, tw)
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-2 * scale, rtol=2e-2)


def test_layer_owned_gate_grouped_routing():
    w13, w2 = _weights(E, 2)
    g = torch.Generator().manual_seed(5)
    M = 24
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    gate = (torch.randn((E, H), generator=g) / 8).to(torch.bfloat16)
    bias = torch.randn(E, generator=g) * 0.1
    rc = RoutingConfig(K, E, scoring_func="sigmoid", use_grouped_topk=True, num_expert_group=4, topk_group=2,
                       e_score_correction_bias=bias.to(DEV), routed_scaling_factor=2.5, apply_routed_scaling_in_router=True)
    layer = RoutedExpertsLayer(_engine(w13, w2, K), rc, gate_weight=gate.to(DEV))
    out = layer.forward(x.to(DEV))
    # routing parity of the fused gate projection is tests/test_gpu_router.py's subject (its fp32 summation order
    # differs from the oracle's, so a near-tie may legitimately flip): here the layer's own routing feeds the
    # reference, and must agree with the oracle's on (nearly) every token
    tw_g, ids_g = layer.select_experts(x.to(DEV), None)
    logits = orc.router_logits(torch_to_bits(x), orc.BF16, torch_to_bits(gate), orc.BF16)
    tw, ids = orc.grouped_topk(logits, K, 4, 2, bias=bias.numpy(), scoring=1, renormalize=True, routed_scaling=2.5)
    same = (np.sort(ids_g.cpu().numpy(), axis=1) == np.sort(ids, axis=1)).all(axis=1)
    assert same.mean() >= 0.9, same
    np.testing.assert_allclose(tw_g.cpu().numpy()[same].sum(axis=1), tw[same].sum(axis=1), rtol=1e-4)
    ref = _moe(w13, w2, x, ids_g.cpu().numpy(), tw_g.cpu().numpy())
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-2 * scale, rtol=2e-2)


def test_layer_with_eplb_maps_and_folded_shared_expert():
    red, n_sh, M = 8, 2, 48
    P = E + red
    w13, w2 = _weights(E, 3)
    g = torch.Generator().manual_seed(6)
    s13 = (torch.randn((2 * n_sh * I, H), generator=g) / 4).to(torch.bfloat16)
    s2 = (torch.randn((H, n_sh * I), generator=g) / 4).to(torch.bfloat16)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    logits = torch.randn((M, E), generator=g)
    c13, c2, _, _ = se.split_shared_expert(s13, s2, n_sh)
    st = eplb.EplbState(1, E, red, device=DEV)
    p2l = st.physical_to_logical_map[0]
    eng = _engine(se.append_shared_experts(w13[p2l], c13), se.append_shared_experts(w2[p2l], c2), K + n_sh)
    slots = se.SharedExpertSlots(P, n_sh, K, max_num_tokens=64, device=DEV)
    layer = RoutedExpertsLayer(eng, RoutingConfig(K, E), eplb_state=st.layer_state(0), shared_slots=slots)
    out = layer.forward(x.to(DEV), logits.to(DEV))
    tw, ids = orc.topk_softmax(logits.numpy(), K)
    routed = _moe(w13, w2, x, ids, tw)
    shared = orc.moe(orc.MoeDesc(E=1, H=H, I=n_sh * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                     torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((M, 1), np.int32), np.ones((M, 1), np.float32))
    ref = routed + shared
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, atol=2e-2 * scale, rtol=2e-2)
    # every routed slot was recorded on a physical expert that holds the right logical expert
    load = st.expert_load_pass[0].cpu().numpy()
    assert load.sum() == M * K
    want = np.bincount(ids.reshape(-1), minlength=E)
    got = np.zeros(E, np.int64)
    np.add.at(got, p2l.numpy(), load)
    np.testing.assert_array_equal(got, want)
