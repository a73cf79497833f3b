"""Control logic of lvllm_amd/layer.py (SURVEY 8 rows a4 / a10 and the chaining of a1-a3, f2, f4) on the CPU: the
operator namespace and the engine are replaced by doubles built on the CPU oracle (tests may do that), so what is under
test is the ORDER and the plumbing -- routing -> EPLB map -> shared slots -> EP map -> decode / prefill -> post.
The same composition on the real kernels: tests/test_zz2_gpu_layer.py."""
import types

import numpy as np
import pytest
import torch

from lvllm_amd import eplb
from lvllm_amd import shared_experts as se
from lvllm_amd.layer import RoutedExpertsLayer, RoutingConfig
from oracle import oracle as orc
from tests.helpers import torch_to_bits

SC = {"softmax": 0, "sigmoid": 1}


def _oracle_ops(calls):
    def topk_softmax(logits, K, renorm, bias=None, scoring="softmax", scale=1.0):
        calls.append("topk")
        w, i = orc.topk_softmax(logits.numpy(), K, bias=None if bias is None else bias.numpy(), scoring=SC[scoring],
                                renormalize=renorm, routed_scaling=scale)
        return torch.from_numpy(w), torch.from_numpy(i)

    def grouped_topk(h, logits, K, renorm, ng, tg, scoring="softmax", scale=1.0, bias=None):
        calls.append("grouped")
        w, i = orc.grouped_topk(logits.numpy(), K, ng, tg, bias=None if bias is None else bias.numpy(),
                                scoring=SC[scoring], renormalize=renorm, routed_scaling=scale)
        return torch.from_numpy(w), torch.from_numpy(i)

    def router_topk(h, gw, K, renorm, *, scoring_func="softmax", num_expert_group=0, topk_group=0,
                    routed_scaling_factor=1.0, e_score_correction_bias=None):
        calls.append("router_gemm")
        logits = torch.from_numpy(orc.router_logits(torch_to_bits(h), orc.BF16, torch_to_bits(gw), orc.BF16))
        if num_expert_group:
            return grouped_topk(h, logits, K, renorm, num_expert_group, topk_group, scoring_func, routed_scaling_factor,
                                e_score_correction_bias)
        return topk_softmax(logits, K, renorm, e_score_correction_bias, scoring_func, routed_scaling_factor)

    def eplb_map(ids, load, l2p, cnt, rec=None, nu=None):
        calls.append("eplb")
        phys, new = orc.eplb_map_record(ids.numpy(), l2p.numpy(), cnt.numpy(), None if load is None else load.numpy(),
                                        True if rec is None else bool(rec.item()), None if nu is None else int(nu.item()))
        if load is not None:
            load.copy_(torch.from_numpy(new))
        return torch.from_numpy(phys)

    def to_local(ids, emap):
        calls.append("ep_map")
        return torch.from_numpy(orc.map_ids(ids.numpy(), emap.numpy()))
    return types.SimpleNamespace(topk_softmax=topk_softmax, grouped_topk=grouped_topk, router_topk=router_topk,
                                 eplb_map_to_physical_and_record=eplb_map, global_to_local_expert_ids=to_local)


class _Engine:
    """RoutedExpertsEngine double: same methods, the CPU oracle inside"""

    def __init__(self, w13, w2, calls, max_num_seqs=16):
        self.w13, self.w2, self.calls = w13, w2, calls
        E, twoI, H = w13.shape
        self.d = orc.MoeDesc(E=E, H=H, I=twoI // 2, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        self.cfg = types.SimpleNamespace(max_num_seqs=max_num_seqs, expert_num=E)

    def _run(self, x, tw, ids):
        return torch.from_numpy(orc.moe(self.d, torch_to_bits(self.w13), torch_to_bits(self.w2), torch_to_bits(x),
                                        ids.numpy(), tw.numpy()))

    def decode(self, x, tw, ids, out=None):
        self.calls.append("decode")
        y = self._run(x, tw, ids)
        out.copy_(y)
        return out

    def prefill(self, x, tw, ids):
        self.calls.append("prefill")
        return self._run(x, tw, ids).to(x.dtype)


E, K, H, I = 8, 2, 64, 32


def _weights(n, seed=1):
    g = torch.Generator().manual_seed(seed)
    return ((torch.randn((n, 2 * I, H), generator=g) / 4).to(torch.bfloat16),
            (torch.randn((n, H, I), generator=g) / 4).to(torch.bfloat16))


def _x(M, seed=2):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((M, H), generator=g) / 2).to(torch.bfloat16), torch.randn((M, E), generator=g)


def _ref(w13, w2, x, logits, scale=1.0, **kw):
    tw, ids = orc.topk_softmax(logits.numpy(), K, **kw)
    y = orc.moe(orc.MoeDesc(E=w13.shape[0], H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(w13),
                torch_to_bits(w2), torch_to_bits(x), ids, tw)
    return (torch.from_numpy(y).to(torch.bfloat16) * scale).float().numpy()


def test_decode_and_prefill_branches_and_post_processing():
    calls = []
    w13, w2 = _weights(E)
    layer = RoutedExpertsLayer(_Engine(w13, w2, calls, max_num_seqs=16), RoutingConfig(K, E, routed_scaling_factor=2.5),
                               ops=_oracle_ops(calls))
    x, logits = _x(9)
    out = layer.forward(x, logits)
    assert calls == ["topk", "decode"] and out.dtype == torch.bfloat16
    np.testing.assert_array_equal(out.float().numpy(), _ref(w13, w2, x, logits, 2.5))
    buf = layer._decode_out
    assert tuple(buf.shape) == (16, H) and buf.dtype == torch.float32
    calls.clear()
    x2, logits2 = _x(40, seed=3)
    out2 = layer.forward(x2, logits2)                                       # > max_num_seqs: the prefill convention
    assert calls == ["topk", "prefill"]
    np.testing.assert_array_equal(out2.float().numpy(), _ref(w13, w2, x2, logits2, 2.5))
    layer.forward(x, logits)
    assert layer._decode_out is buf, "the decode buffer is allocated once (cpu_decode contract: one shared buffer)"
    assert layer.forward(x[:0], logits[:0]).shape == (0, H)
    with pytest.raises(ValueError):
        layer.forward(x, None)
    with pytest.raises(ValueError):
        layer.forward(x, logits[:3])


def test_nan_scrub_follows_the_reference_switch():
    calls = []
    w13, w2 = _weights(E)
    w2[0, 0, 0] = float("nan")
    x, logits = _x(6)
    logits[:, 0] += 10.0                                                    # everybody routes to the poisoned expert
    for M_cap in (16, 2):                                                   # decode branch, prefill branch
        dirty = RoutedExpertsLayer(_Engine(w13, w2, calls, M_cap), RoutingConfig(K, E), ops=_oracle_ops(calls)).forward(x, logits)
        clean = RoutedExpertsLayer(_Engine(w13, w2, calls, M_cap), RoutingConfig(K, E), check_nan_in_output=True,
                                   ops=_oracle_ops(calls)).forward(x, logits)
        assert torch.isnan(dirty[:, 0]).all() and torch.isfinite(clean).all()
        assert torch.equal(clean[:, 1:], dirty[:, 1:]) and (clean[:, 0] == 0).all()


def test_owned_gate_grouped_routing_and_scaling_in_the_router():
    calls = []
    w13, w2 = _weights(E)
    g = torch.Generator().manual_seed(9)
    gate = (torch.randn((E, H), generator=g) / 8).to(torch.bfloat16)
    bias = torch.randn(E, generator=g) * 0.1
    rc = RoutingConfig(K, E, renormalize=True, scoring_func="sigmoid", use_grouped_topk=True, num_expert_group=4,
                       topk_group=2, e_score_correction_bias=bias, routed_scaling_factor=2.5,
                       apply_routed_scaling_in_router=True)
    layer = RoutedExpertsLayer(_Engine(w13, w2, calls), rc, gate_weight=gate, ops=_oracle_ops(calls))
    x, _ = _x(5)
    out = layer.forward(x)
    assert calls == ["router_gemm", "grouped", "decode"]
    logits = orc.router_logits(torch_to_bits(x), orc.BF16, torch_to_bits(gate), orc.BF16)
    tw, ids = orc.grouped_topk(logits, K, 4, 2, bias=bias.numpy(), scoring=1, renormalize=True, routed_scaling=2.5)
    y = orc.moe(orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(w13), torch_to_bits(w2),
                torch_to_bits(x), ids, tw)
    np.testing.assert_array_equal(out.float().numpy(), torch.from_numpy(y).to(torch.bfloat16).float().numpy())
    with pytest.raises(ValueError):
        RoutedExpertsLayer(_Engine(w13, w2, calls), RoutingConfig(K, E, use_grouped_topk=True), ops=_oracle_ops(calls))


def test_eplb_shared_experts_and_ep_map_chain_in_the_reference_order():
    """logical ids -> physical ids (EPLB) -> + shared slots -> local ids (EP map with the shared tail): the engine
    double holds THIS rank's physical experts + the shared chunks; summed over ranks == one rank with everything"""
    calls = []
    red, n_sh, ep = 4, 2, 2
    P = E + red
    w13, w2 = _weights(E)
    g = torch.Generator().manual_seed(8)                                   # one shared expert of size n_sh * I
    s13 = (torch.randn((2 * n_sh * I, H), generator=g) / 4).to(torch.bfloat16)
    s2 = (torch.randn((H, n_sh * I), generator=g) / 4).to(torch.bfloat16)
    c13, c2, _, _ = se.split_shared_expert(s13, s2, n_sh)
    st = eplb.EplbState(1, E, red)
    p2l = st.physical_to_logical_map[0]
    per = P // ep
    x, logits = _x(11)
    total = torch.zeros((11, H))
    for rank in range(ep):
        mine = p2l[rank * per:(rank + 1) * per]
        e13 = se.append_shared_experts(w13[mine], c13)
        e2 = se.append_shared_experts(w2[mine], c2)
        n_loc, emap = orc.expert_map(ep, rank, P, 0)
        full_map = torch.cat([torch.from_numpy(emap), se.shared_expert_map_tail(n_loc, n_sh)])
        slots = se.SharedExpertSlots(P, n_sh, K, ep_rank=rank, ep_size=ep, is_ep=True, max_num_tokens=32)
        calls.clear()
        layer = RoutedExpertsLayer(_Engine(e13, e2, calls), RoutingConfig(K, E), expert_map=full_map,
                                   eplb_state=st.layer_state(0), shared_slots=slots, ops=_oracle_ops(calls))
        total += layer.forward(x, logits).float()
        assert calls == ["topk", "eplb", "ep_map", "decode"]
    tw, ids = orc.topk_softmax(logits.numpy(), K)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    routed = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids, tw)
    shared = orc.moe(orc.MoeDesc(E=1, H=H, I=n_sh * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                     torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((11, 1), np.int32), np.ones((11, 1), np.float32))
    want = routed + shared
    np.testing.assert_allclose(total.numpy(), want, atol=0.03 * np.abs(want).max(), rtol=2e-2)      # two bf16 roundings
    # both ranks recorded the same tokens: the load view counts every routed slot twice, shared slots never
    assert int(st.expert_load_pass.sum()) == 2 * 11 * K


@pytest.mark.parametrize("gated", [False, True])
def test_folded_shared_expert_is_not_scaled_by_the_routed_scaling_factor(gated):
    """DeepSeek-V3 style: routed_scaling_factor 2.5 applied to the layer OUTPUT, one folded shared expert.  The
    reference compensates with a shared slot weight of 1/rsf (fused_moe/layer.py:306-318): out = rsf * routed +
    1 * shared (x sigmoid gate for the Qwen2-MoE form)."""
    calls = []
    rsf, n_sh, M = 2.5, 2, 9
    w13, w2 = _weights(E)
    g = torch.Generator().manual_seed(8)
    s13 = (torch.randn((2 * n_sh * I, H), generator=g) / 4).to(torch.bfloat16)
    s2 = (torch.randn((H, n_sh * I), generator=g) / 4).to(torch.bfloat16)
    gate_w = (torch.randn((1, H), generator=g) / 4).to(torch.bfloat16) if gated else None
    c13, c2, _, _ = se.split_shared_expert(s13, s2, n_sh)
    e13, e2 = se.append_shared_experts(w13, c13), se.append_shared_experts(w2, c2)
    slots = se.SharedExpertSlots(E, n_sh, K, max_num_tokens=32)
    layer = RoutedExpertsLayer(_Engine(e13, e2, calls), RoutingConfig(K, E, routed_scaling_factor=rsf),
                               shared_slots=slots, shared_gate_weight=gate_w, ops=_oracle_ops(calls))
    x, logits = _x(M)
    got = layer.forward(x, logits).float().numpy()
    tw, ids = orc.topk_softmax(logits.numpy(), K)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
    routed = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(x), ids, tw)
    sw = np.ones((M, 1), np.float32)
    if gated:
        sw = torch.sigmoid(torch.nn.functional.linear(x, gate_w).float()).numpy()
    shared = orc.moe(orc.MoeDesc(E=1, H=H, I=n_sh * I, act_dtype=orc.BF16, wfmt=orc.W_BF16), torch_to_bits(s13[None]),
                     torch_to_bits(s2[None]), torch_to_bits(x), np.zeros((M, 1), np.int32), sw)
    want = rsf * routed + shared
    np.testing.assert_allclose(got, want, atol=0.03 * np.abs(want).max(), rtol=2e-2)
    wrong = rsf * (routed + shared)                      # what an uncompensated slot weight of 1.0 gives
    assert np.abs(got - wrong).max() > 10 * np.abs(got - want).max()


# ------------------------------------------------------------------------------------------ the layer over EP + EPLB
def _ep_layer_worker(rank, world, port, q):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lvllm_amd.ep import ExpertParallelExperts
        from tests.helpers import TorchEpKernels
        calls = []
        red = 4
        P, per = E + red, (E + red) // world
        w13, w2 = _weights(E)
        st = eplb.EplbState(1, E, red, window_size=2, step_interval=1)
        mine = torch.tensor(st.local_logical_ids(0))
        l13, l2 = w13[mine].contiguous(), w2[mine].contiguous()
        st.expert_stores = [eplb.TensorExpertStore([l13, l2])]

        class _Live:                                              # engine double over the tensors the exchange maintains
            cfg = types.SimpleNamespace(max_num_seqs=64, expert_num=per)

            def decode(self, x, tw, ids, out=None):
                if x.shape[0] == 0:
                    return torch.zeros((0, H))
                d = orc.MoeDesc(E=per, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
                return torch.from_numpy(orc.moe(d, torch_to_bits(l13), torch_to_bits(l2), torch_to_bits(x), ids.numpy(),
                                                tw.numpy()))
        eng = _Live()
        ep = ExpertParallelExperts(lambda rows, lids, ws, dt: eng.decode(rows.contiguous(), ws.contiguous(), lids.contiguous()).to(dt),
                                   P, H, mode="a2a", kernels=TorchEpKernels)
        layer = RoutedExpertsLayer(eng, RoutingConfig(K, E, routed_scaling_factor=2.0), eplb_state=st.layer_state(0),
                                   expert_parallel=ep, ops=_oracle_ops(calls))
        x, logits = _x(10, seed=30 + rank)
        logits[:, 2] += 2.0                                        # a hot expert on every rank
        want = _ref(w13, w2, x, logits, 2.0)
        ok = bool(np.allclose(layer.forward(x, logits).float().numpy(), want, atol=0.02 * np.abs(want).max(), rtol=2e-2))
        before = st.physical_to_logical_map.clone()
        ok = ok and st.step() and not torch.equal(before, st.physical_to_logical_map)      # rearranged
        ok = ok and bool(np.allclose(layer.forward(x, logits).float().numpy(), want, atol=0.02 * np.abs(want).max(), rtol=2e-2))
        q.put((rank, (ok, calls[:2], int(st.logical_replica_count[0, 2]))))
    finally:
        dist.destroy_process_group()


def test_layer_over_expert_parallel_all_to_all_and_eplb():
    from tests.test_eplb import _spawn
    res = _spawn(_ep_layer_worker, 2)
    assert all(ok for ok, _, _ in res.values()), res
    assert all(c == ["topk", "eplb"] for _, c, _ in res.values())
    assert all(n >= 2 for _, _, n in res.values()), "the hot expert was not replicated"
    with pytest.raises(ValueError):
        RoutedExpertsLayer(_Engine(*_weights(E), []), RoutingConfig(K, E), expert_parallel=object(),
                           expert_map=torch.zeros(E, dtype=torch.int32), ops=_oracle_ops([]))
