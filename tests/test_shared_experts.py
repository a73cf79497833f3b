"""Shared experts folded into the routed grouped GEMM (lvllm_amd/shared_experts.py) on the CPU:
the split is exact algebra (checked through the oracle: E + n experts with the extended slots == routed experts +
the shared expert computed as ONE expert of intermediate size n*I), for 16-bit, fp8-block and int4 weights; the slot
buffers behave as the reference's (`init_aiter_topK_meta_data` / `inject_shared_expert_weights`,
experts/rocm_aiter_moe.py:61-158), incl. the EP token-ownership sentinel."""
import numpy as np
import pytest
import torch

from lvllm_amd import shared_experts as se
from oracle import oracle as orc
from tests.helpers import bits_to_torch, make_routing, torch_to_bits

E, K, H, I, N, M = 6, 2, 256, 128, 3, 21


def _master(seed=3):
    g = torch.Generator().manual_seed(seed)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 4).to(torch.bfloat16)
    s13 = (torch.randn((2 * N * I, H), generator=g) / 4).to(torch.bfloat16)        # one shared expert, size N*I
    s2 = (torch.randn((H, N * I), generator=g) / 4).to(torch.bfloat16)
    x = (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed=seed)
    return w13, w2, s13, s2, x, tw, ids


def _extended(tw, ids, score=1.0, gate=None):
    slots = se.SharedExpertSlots(E, N, K, shared_experts_score=score, max_num_tokens=64)
    w, i = slots.inject(torch.from_numpy(tw), torch.from_numpy(ids), gate)
    assert tuple(w.shape) == (M, K + N) and w.is_contiguous() and i.is_contiguous() and i.dtype == torch.int32
    return w.numpy().copy(), i.numpy().copy()


def test_split_is_exact_algebra_bf16():
    w13, w2, s13, s2, x, tw, ids = _master()
    c13, c2, _, _ = se.split_shared_expert(s13, s2, N)
    assert tuple(c13.shape) == (N, 2 * I, H) and tuple(c2.shape) == (N, H, I)
    # chunk c holds gate rows [c*I, (c+1)*I) then up rows [N*I + c*I, ...), and the matching columns of down
    assert torch.equal(c13[1, :I], s13[I:2 * I]) and torch.equal(c13[1, I:], s13[N * I + I:N * I + 2 * I])
    assert torch.equal(c2[2], s2[:, 2 * I:3 * I])
    f13, f2 = se.append_shared_experts(w13, c13), se.append_shared_experts(w2, c2)
    etw, eids = _extended(tw, ids)
    d = orc.MoeDesc(E=E + N, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    fused = orc.moe(d, torch_to_bits(f13), torch_to_bits(f2), torch_to_bits(x), eids, etw)
    routed = orc.moe(orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(w13),
                     torch_to_bits(w2), torch_to_bits(x), ids, tw)
    one = orc.moe(orc.MoeDesc(E=1, H=H, I=N * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                  torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((M, 1), np.int32), np.ones((M, 1), np.float32))
    np.testing.assert_allclose(fused, routed + one, atol=2e-5 * np.abs(one).max(), rtol=1e-5)
    # and against plain torch fp32 (the dense MLP the reference would run): the bf16-intermediate tolerance
    xf = x.float()
    g, u = xf @ s13[:N * I].float().T, xf @ s13[N * I:].float().T
    mlp = (torch.nn.functional.silu(g) * u).to(torch.bfloat16).float() @ s2.float().T
    np.testing.assert_allclose(fused - routed, mlp.numpy(), atol=2e-3 * float(mlp.abs().max()), rtol=1e-2)


def test_gated_shared_expert_and_score():
    w13, w2, s13, s2, x, tw, ids = _master(seed=4)
    c13, c2, _, _ = se.split_shared_expert(s13, s2, N)
    f13, f2 = se.append_shared_experts(w13, c13), se.append_shared_experts(w2, c2)
    gate = torch.sigmoid(torch.randn((M, 1), generator=torch.Generator().manual_seed(1))).expand(M, N).contiguous()
    etw, eids = _extended(tw, ids, gate=gate)
    np.testing.assert_array_equal(etw[:, K:], gate.numpy())
    d = orc.MoeDesc(E=E + N, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    fused = orc.moe(d, torch_to_bits(f13), torch_to_bits(f2), torch_to_bits(x), eids, etw)
    routed = orc.moe(orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(w13),
                     torch_to_bits(w2), torch_to_bits(x), ids, tw)
    one = orc.moe(orc.MoeDesc(E=1, H=H, I=N * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                  torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((M, 1), np.int32),
                  gate[:, :1].numpy().astype(np.float32))
    np.testing.assert_allclose(fused, routed + one, atol=2e-5 * np.abs(one).max(), rtol=1e-5)


@pytest.mark.parametrize("fmt", ["fp8", "int4"])
def test_split_carries_block_and_group_scales(fmt):
    _, _, s13, s2, x, _, _ = _master(seed=5)
    ids1, ones = np.zeros((M, 1), np.int32), np.ones((M, 1), np.float32)
    idsn = np.tile(np.arange(N, dtype=np.int32), (M, 1))
    onesn = np.ones((M, N), np.float32)
    if fmt == "fp8":
        q13, sc13 = orc.quant_fp8_block(s13[None].float().numpy(), 128, 128)
        q2, sc2 = orc.quant_fp8_block(s2[None].float().numpy(), 128, 128)
        kw = dict(wfmt=orc.W_FP8, groupN=128, groupK=128)
        to_t = torch.from_numpy
    else:
        q13, sc13 = orc.quant_int4(torch_to_bits(s13[None]), orc.BF16, 64)
        q2, sc2 = orc.quant_int4(torch_to_bits(s2[None]), orc.BF16, 64)
        kw = dict(wfmt=orc.W_INT4, groupN=1, groupK=64)
        to_t = torch.from_numpy
    c13, c2, cs13, cs2 = se.split_shared_expert(to_t(q13[0]), to_t(q2[0]), N, w13_scale=to_t(sc13[0]),
                                                w2_scale=to_t(sc2[0]))
    assert c13.shape[0] == N and cs13.shape[0] == N and cs2.shape[0] == N
    one = orc.moe(orc.MoeDesc(E=1, H=H, I=N * I, act_dtype=orc.BF16, **kw), q13, q2, torch_to_bits(x), ids1, ones,
                  s13=sc13, s2=sc2)
    split = orc.moe(orc.MoeDesc(E=N, H=H, I=I, act_dtype=orc.BF16, **kw), c13.numpy(), c2.numpy(), torch_to_bits(x),
                    idsn, onesn, s13=cs13.numpy(), s2=cs2.numpy())
    np.testing.assert_allclose(split, one, atol=2e-5 * np.abs(one).max(), rtol=1e-5)


def test_split_rejects_indivisible_shapes_and_mismatched_experts():
    with pytest.raises(ValueError):
        se.split_shared_expert(torch.zeros((2 * 100, 8)), torch.zeros((8, 100)), 3)
    with pytest.raises(ValueError):
        se.split_shared_expert(torch.zeros((2, 6, 8)), torch.zeros((8, 6)), 2)
    with pytest.raises(ValueError):
        se.append_shared_experts(torch.zeros((4, 8, 16)), torch.zeros((1, 8, 32)))
    with pytest.raises(ValueError):
        se.append_shared_experts(torch.zeros((4, 8, 16)), torch.zeros((1, 8, 16), dtype=torch.bfloat16))
    c13, c2, _, _ = se.split_shared_expert(torch.arange(12.).reshape(6, 2), torch.arange(6.).reshape(2, 3), 3,
                                           has_gate_proj=False)
    assert tuple(c13.shape) == (3, 2, 2) and tuple(c2.shape) == (3, 2, 1)


def test_slot_buffers_follow_the_reference_layout():
    """rocm_aiter_moe.py:61-110: ids n_routed .. n_routed+n-1 and the score in the shared columns; under EP token i
    belongs to rank i % ep_size, the others (and the sentinel column) carry the id n_routed + n_shared"""
    s = se.SharedExpertSlots(8, 2, 3, shared_experts_score=0.5, max_num_tokens=10)
    assert s.width == 5
    w, ids = s.inject(torch.full((4, 3), 0.25), torch.tensor([[1, 2, 3]] * 4, dtype=torch.int32))
    assert ids.tolist() == [[1, 2, 3, 8, 9]] * 4 and w.tolist() == [[0.25, 0.25, 0.25, 0.5, 0.5]] * 4
    ep = se.SharedExpertSlots(8, 2, 3, ep_rank=1, ep_size=4, is_ep=True, max_num_tokens=10)
    assert ep.width == 6 and ep.fake_id == 10
    _, ids = ep.inject(torch.zeros((9, 3)), torch.zeros((9, 3), dtype=torch.int32))
    for i in range(9):
        assert ids[i, 3:].tolist() == ([8, 9, 10] if i % 4 == 1 else [10, 10, 10])
    # the expert map of an EP rank: routed map, then the shared experts behind its local experts, then the sentinel
    n_loc, emap = orc.expert_map(4, 1, 8, 0)
    full = np.concatenate([emap, se.shared_expert_map_tail(n_loc, 2).numpy()])
    local = orc.map_ids(ids.numpy(), full)
    assert local[1, 3:].tolist() == [n_loc, n_loc + 1, -1] and local[0, 3:].tolist() == [-1, -1, -1]
    with pytest.raises(ValueError):
        s.inject(torch.zeros((11, 3)), torch.zeros((11, 3), dtype=torch.int32))
    with pytest.raises(ValueError):
        s.inject(torch.zeros((2, 4)), torch.zeros((2, 4), dtype=torch.int32))
    with pytest.raises(ValueError):
        se.SharedExpertSlots(8, 0, 2)


def test_expert_map_with_fused_shared_experts_equals_the_reference_run():
    """ops.determine_expert_map(..., num_fused_shared_experts=n): the first E + n entries are the reference's
    extended expert_map, and (map >= 0) is the reference's expert_mask, whose last entry masks the sentinel id
    (goldens: tests/golden/make_golden_eplb.py runs expert_map_manager.py:22-113 with return_expert_mask=True)"""
    import os
    from lvllm_amd import ops
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "eplb.npz"))
    assert int(gold["n_sm"]) >= 50
    for ci in range(int(gold["n_sm"])):
        E_, ep, r, n_sh, strat, n_loc = gold[f"sm{ci}_meta"].tolist()
        local, emap = ops.determine_expert_map(ep, r, E_, "linear" if strat == 0 else "round_robin", n_sh)
        assert local == n_loc and emap.dtype == torch.int32 and emap.numel() == E_ + n_sh + 1
        np.testing.assert_array_equal(emap[:-1].numpy(), gold[f"sm{ci}_map"])
        np.testing.assert_array_equal((emap >= 0).to(torch.int32).numpy(), gold[f"sm{ci}_mask"])
        np.testing.assert_array_equal(emap[E_:].numpy(), se.shared_expert_map_tail(n_loc, n_sh).numpy())
