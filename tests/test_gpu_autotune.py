"""First-call micro-autotune of a step's plan (tuning key "autotune", lkm_api.hip run_chunk_tuned):
the first eager call of a step shape times the candidate plans on the caller's inputs and keeps the fastest; every
candidate is a parity-tested plan, so the result stays inside the oracle tolerance whichever wins; captures that follow
the warm-up call replay the chosen plan; a forced plan or autotune = 0 gives the thresholds of pick_cfg back."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, make_routing, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(fmt, M, E, K, H, I, seed):
    from lvllm_amd.ops import RoutedExpertsEngine
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((M, H), generator=g) / 10).to(torch.bfloat16)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 10).to(torch.bfloat16)
    tw, ids = make_routing(M, E, K, seed)
    if fmt == "int4":
        q13, s13 = orc.quant_int4(torch_to_bits(w13), orc.BF16, 128)
        q2, s2 = orc.quant_int4(torch_to_bits(w2), orc.BF16, 128)
        eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=torch.bfloat16, fmt="int4",
                                  w13_scale=bits_to_torch(s13, orc.BF16), w2_scale=bits_to_torch(s2, orc.BF16), group_n=1, group_k=128)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=128)
        ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
    else:
        eng = RoutedExpertsEngine(w13, w2, top_k=K, act_dtype=torch.bfloat16)
        d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_BF16)
        ref = orc.moe(d, torch_to_bits(w13), torch_to_bits(w2), torch_to_bits(a), ids, tw)
    return eng, a.to(DEV), torch.from_numpy(tw).to(DEV), torch.from_numpy(ids).to(DEV), ref


@pytest.mark.parametrize("fmt,M", [("int4", 128), ("int4", 40), ("bf16", 48)])
def test_autotune_picks_a_tested_plan_and_graphs_replay_it(fmt, M):
    E, K, H, I = 8, 2, 512, 384
    eng, a, tw, ids, ref = _case(fmt, M, E, K, H, I, seed=M)
    base = eng.decode(a, tw, ids).clone()
    assert "autotuned" not in eng.engine.describe()
    eng.engine.set_tuning(autotune=1)
    out = eng.decode(a, tw, ids).clone()                     # first eager call of the shape: times the candidates
    d1 = eng.engine.describe()
    assert "autotuned:" in d1, d1
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-3, rtol=1e-2)
    out2 = eng.decode(a, tw, ids)                            # the remembered plan: same bits as the call that chose it
    assert torch.equal(out2, out) and eng.engine.describe() == d1
    # captured after the warm-up call: the replay runs the chosen plan (nothing is timed while capturing)
    xs = a.clone()
    o = torch.empty((M, H), dtype=torch.float32, device=DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.decode(xs, tw, ids, out=o)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.decode(xs, tw, ids, out=o)
    o.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(o, out)
    # a shape first seen while capturing is not tuned (the default plan runs), a forced plan is not tuned either
    eng.engine.set_tuning(tiled=64)
    eng.decode(a, tw, ids)
    assert "autotuned" not in eng.engine.describe()
    eng.engine.set_tuning(tiled=0, autotune=0)
    again = eng.decode(a, tw, ids)
    assert "autotuned" not in eng.engine.describe() and torch.equal(again, base)


def test_autotune_robustness_and_the_plan_agreement_calls():
    """ADVICE r4 + verdict item 8: an in-place caller is never tuned (the candidates would run repeatedly on its own
    buffers), changing a planning knob forgets the remembered plans, the remembered plans can be read and replaced by
    candidate index (what ep.agree_tuned_plans does across a group), and an expert-parallel engine refuses autotune = 1."""
    from lvllm_amd._clib import LkmError
    from lvllm_amd.ops import RoutedExpertsEngine
    M, E, K, H, I = 128, 8, 2, 512, 384
    eng, a, tw, ids, ref = _case("int4", M, E, K, H, I, seed=77)
    eng.engine.set_tuning(autotune=1)
    inplace = a.clone()
    eng.forward_rows(inplace, tw, ids, out=inplace, out_dtype=torch.bfloat16)          # out aliases the token rows
    assert eng.engine.tuned_plans() == [] and "autotuned" not in eng.engine.describe()
    np.testing.assert_allclose(inplace.float().cpu().numpy(), ref, atol=2e-2 * float(np.abs(ref).max()), rtol=2e-2)
    out = eng.decode(a, tw, ids).clone()
    plans = eng.engine.tuned_plans()
    assert len(plans) == 1 and plans[0][1] >= 0, plans
    key, idx = plans[0]
    # replace the choice by candidate 0 (= the default plan), as a group agreement would: the next call runs it
    eng.engine.set_tuned_plan(key, 0)
    assert eng.engine.tuned_plans() == [(key, 0)]
    dflt = eng.decode(a, tw, ids).clone()
    assert "autotuned: default" in eng.engine.describe(), eng.engine.describe()
    np.testing.assert_allclose(dflt.cpu().numpy(), ref, atol=2e-3, rtol=1e-2)
    with pytest.raises(LkmError):
        eng.engine.set_tuned_plan(key, 99)                     # outside the shape's candidate list
    with pytest.raises(LkmError):
        eng.engine.set_tuned_plan(key + 1, 0)                  # a shape that was never tuned here
    eng.engine.set_tuning(nt2=1)                               # a planning knob: remembered plans are for the old knobs
    assert eng.engine.tuned_plans() == []
    eng.engine.set_tuning(nt2=0, autotune=0)
    # expert-parallel engine (num_processes > 1): 1 is refused loudly, 2 = "the host agrees on the plans" is accepted
    g = torch.Generator().manual_seed(5)
    w13 = (torch.randn((4, 2 * 128, 256), generator=g) / 10).to(torch.bfloat16)
    w2 = (torch.randn((4, 256, 128), generator=g) / 10).to(torch.bfloat16)
    ep_eng = RoutedExpertsEngine(w13, w2, top_k=2, act_dtype=torch.bfloat16, num_processes=2, process_id=0)
    with pytest.raises(LkmError, match="agree"):
        ep_eng.engine.set_tuning(autotune=1)
    ep_eng.engine.set_tuning(autotune=2)
    ep_eng.engine.set_tuning(autotune=0)
