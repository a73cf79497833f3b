"""GPU parity: router + scatter kernels, through the C ABI (ctypes), vs the CPU oracle (bit-exact)
and vs the reference's golden vectors."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, load_golden
from tests.test_oracle_golden import _assert_ids_equal_modulo_exact_ties

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from lvllm_amd import ops
    return ops


def _np_logits(t: torch.Tensor):
    t = t.cpu().contiguous()
    if t.dtype == torch.float32:
        return t.numpy(), orc.F32
    code = orc.BF16 if t.dtype == torch.bfloat16 else orc.F16
    return t.view(torch.int16).numpy().view(np.uint16), code


def test_topk_golden_bit_exact_vs_oracle_and_reference():
    ops = _ops()
    for i, c in load_golden("topk.npz"):
        m, e, k, renorm, scoring, dt, has_bias = [int(v) for v in c["meta"]]
        logits = bits_to_torch(c["logits"], dt).to(DEV)
        bias = torch.from_numpy(c["bias"]).to(DEV) if has_bias else None
        w, ids = ops.topk_softmax(logits, k, bool(renorm), bias, "softmax" if scoring == 0 else "sigmoid")
        ow, oids = orc.topk_softmax(c["logits"], k, dt=dt, bias=c.get("bias"), scoring=scoring,
                                    renormalize=bool(renorm))
        np.testing.assert_array_equal(ids.cpu().numpy(), oids, err_msg=f"case {i}")      # bit-exact ids
        np.testing.assert_array_equal(w.cpu().numpy().view(np.uint32), ow.view(np.uint32),
                                      err_msg=f"case {i}: weights not bit-identical to the oracle")
        lf = orc.bits_to_f32(c["logits"], dt) if dt != orc.F32 else c["logits"]
        _assert_ids_equal_modulo_exact_ties(ids.cpu().numpy(), c["ids"], lf, c.get("bias"), i)
        np.testing.assert_allclose(w.cpu().numpy(), c["w"], atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("E,K", [(8, 2), (60, 4), (128, 8), (256, 8), (384, 8), (512, 22)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_topk_large_random_bit_exact(E, K, dtype):
    ops = _ops()
    torch.manual_seed(E * 131 + K)
    M = 777
    logits = (torch.randn((M, E)) * 3).to(dtype)
    bias = torch.randn(E)
    for scoring in ("softmax", "sigmoid"):
        for b in (None, bias):
            for renorm, rsf in ((True, 1.0), (False, 2.5)):
                w, ids = ops.topk_softmax(logits.to(DEV), K, renorm, None if b is None else b.to(DEV),
                                          scoring, rsf)
                ln, code = _np_logits(logits)
                ow, oids = orc.topk_softmax(ln, K, dt=code, bias=None if b is None else b.numpy(),
                                            scoring=0 if scoring == "softmax" else 1,
                                            renormalize=renorm, routed_scaling=rsf)
                np.testing.assert_array_equal(ids.cpu().numpy(), oids)
                np.testing.assert_array_equal(w.cpu().numpy().view(np.uint32), ow.view(np.uint32))


def test_topk_nan_inf_rows():
    ops = _ops()
    for bad in (float("nan"), float("inf")):
        for scoring in ("softmax", "sigmoid"):
            torch.manual_seed(0)
            g = torch.randn((4, 8))
            g[1:, :] = bad
            w, ids, tei = ops.fused_topk(torch.empty((4, 16), device=DEV), g.to(DEV), 4, False,
                                         scoring_func=scoring)
            for r in range(1, 4):
                assert ids[r].tolist() == [0, 1, 2, 3]
                assert torch.isfinite(w[r]).all()
            assert tei.tolist() == [[k * 4 + m for k in range(4)] for m in range(4)]


def test_topk_errors():
    ops = _ops()
    g = torch.randn((4, 8), device=DEV)
    with pytest.raises(ValueError):
        ops.topk_softmax(g, 2, True, None, "tanh")
    with pytest.raises(Exception):
        ops.topk_softmax(g, 9, True)          # K > E
    with pytest.raises(ValueError):
        ops.topk_softmax(g.cpu(), 2, True)    # no CPU path


def test_grouped_topk_bit_exact_vs_oracle_and_reference_sets():
    ops = _ops()
    for i, c in load_golden("grouped_topk.npz"):
        m, e, k, ng, tg, renorm, scoring, has_bias = [int(v) for v in c["meta"]]
        logits = torch.from_numpy(c["logits"]).to(DEV)
        bias = torch.from_numpy(c["bias"]).to(DEV) if has_bias else None
        w, ids = ops.grouped_topk(torch.empty((m, 0), device=DEV), logits, k, bool(renorm), ng, tg,
                                  "softmax" if scoring == 0 else "sigmoid", float(c["rsf"]), bias)
        ow, oids = orc.grouped_topk(c["logits"], k, ng, tg, bias=c.get("bias"), scoring=scoring,
                                    renormalize=bool(renorm), routed_scaling=float(c["rsf"]))
        np.testing.assert_array_equal(ids.cpu().numpy(), oids, err_msg=f"case {i}")
        np.testing.assert_array_equal(w.cpu().numpy().view(np.uint32), ow.view(np.uint32))
        wn, idn = w.cpu().numpy(), ids.cpu().numpy()
        for r in range(m):
            o1, o2 = np.argsort(idn[r], kind="stable"), np.argsort(c["ids"][r], kind="stable")
            np.testing.assert_array_equal(idn[r][o1], c["ids"][r][o2])
            np.testing.assert_allclose(wn[r][o1], c["w"][r][o2], rtol=3e-6, atol=1e-7)


def test_map_expert_ids_and_expert_map():
    ops = _ops()
    for i, c in load_golden("expert_map.npz"):
        ep, r, E, strat, nloc = [int(v) for v in c["meta"]]
        n, emap = ops.determine_expert_map(ep, r, E, "linear" if strat == 0 else "round_robin")
        assert n == nloc
        if emap is not None:
            np.testing.assert_array_equal(emap.numpy(), c["map"])
    rng = np.random.default_rng(0)
    E = 256
    _, emap = orc.expert_map(8, 3, E, 0)
    ids = rng.integers(-2, E, size=(500, 8)).astype(np.int32)
    got = ops.global_to_local_expert_ids(torch.from_numpy(ids).to(DEV), torch.from_numpy(emap))
    np.testing.assert_array_equal(got.cpu().numpy(), orc.map_ids(ids, emap))


@pytest.mark.parametrize("n,E", [(0, 4), (1, 1), (64, 8), (1000, 7), (1024, 128), (1025, 256),
                                 (5000, 512), (70000, 128)])
def test_sort_slots_exact(n, E):
    ops = _ops()
    rng = np.random.default_rng(n + E)
    ids = rng.integers(-1, E + 1, size=n).astype(np.int32)      # -1 and E are "skip"
    counts, offsets, sorted_slot, pos = ops.sort_slots(torch.from_numpy(ids).to(DEV), E)
    oc, oo, os_, op = orc.sort_slots(ids, E)
    np.testing.assert_array_equal(counts.cpu().numpy(), oc)
    np.testing.assert_array_equal(offsets.cpu().numpy(), oo)
    np.testing.assert_array_equal(sorted_slot.cpu().numpy(), os_)
    np.testing.assert_array_equal(pos.cpu().numpy(), op)
