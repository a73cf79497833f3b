"""The CPU oracle pinned against golden vectors produced by the REFERENCE's own python oracles
(tests/golden/make_golden.py).  Tolerances are the reference's (cited per test)."""
import math

import numpy as np

from oracle import oracle as orc
from tests.helpers import load_golden


def test_expf_accuracy_and_edges():
    xs = np.concatenate([np.linspace(-103, 88, 20001), np.linspace(-2, 2, 4001)]).astype(np.float32)
    worst = 0.0
    for x in xs:
        got, ref = orc.expf(float(x)), math.exp(float(x))
        if ref > 1e-37:                      # normal range: relative error in ulps
            worst = max(worst, abs(got - ref) / (ref * 2 ** -23))
    assert worst < 1.6, worst
    assert orc.expf(0.0) == 1.0
    assert orc.expf(-200.0) == 0.0 and orc.expf(100.0) == float("inf")
    assert math.isnan(orc.expf(float("nan")))
    assert orc.expf(-100.0) > 0.0            # subnormal, not flushed


def _assert_ids_equal_modulo_exact_ties(ids, ref_ids, logits, bias, case):
    """ids must equal the reference's, except where two experts have EXACTLY equal logits (and
    bias): there torch.topk's order is unspecified while the reference kernel -- and we -- take
    the lowest index first (topk_softmax_kernels.cu:515-522, :537)."""
    for r in np.where((ids != ref_ids).any(axis=1))[0]:
        key = logits[r] if bias is None else None
        assert key is not None, f"case {case} row {r}: id mismatch with bias {ids[r]} vs {ref_ids[r]}"
        for a, b in zip(ids[r], ref_ids[r]):
            assert key[a] == key[b], f"case {case} row {r}: {ids[r]} vs {ref_ids[r]} (not a tie)"
        # among tied experts ours must be ascending
        for j in range(len(ids[r]) - 1):
            if key[ids[r][j]] == key[ids[r][j + 1]]:
                assert ids[r][j] < ids[r][j + 1]


def test_topk_vs_reference_torch_topk():
    """ids atol=0, weights atol=rtol=1e-2: tests/kernels/moe/test_fused_topk.py:86-89."""
    n = 0
    for i, c in load_golden("topk.npz"):
        m, e, k, renorm, scoring, dt, has_bias = [int(v) for v in c["meta"]]
        w, ids = orc.topk_softmax(c["logits"], k, dt=dt, bias=c.get("bias"), scoring=scoring,
                                  renormalize=bool(renorm))
        _assert_ids_equal_modulo_exact_ties(ids, c["ids"], orc.bits_to_f32(c["logits"], dt)
                                            if dt != orc.F32 else c["logits"], c.get("bias"), i)
        np.testing.assert_allclose(w, c["w"], atol=1e-2, rtol=1e-2)
        # our own tighter bound: fp32 softmax restatement
        np.testing.assert_allclose(w, c["w"], atol=2e-6, rtol=2e-5)
        n += 1
    assert n > 100


def test_topk_nan_inf_rows_give_unique_ids():
    """test_fused_topk.py:158-214: poisoned rows -> unique ids, finite weights."""
    for bad in (float("nan"), float("inf")):
        for scoring in (0, 1):
            for e, k in ((6, 3), (8, 4), (16, 4)):
                rng = np.random.default_rng(0)
                logits = rng.standard_normal((4, e)).astype(np.float32)
                logits[1:, :] = bad
                w, ids = orc.topk_softmax(logits, k, scoring=scoring, renormalize=False)
                for r in range(1, 4):
                    assert len(set(ids[r].tolist())) == k
                    assert np.isfinite(w[r]).all()
                    assert ids[r].tolist() == list(range(k))   # index tie-break: [0..k-1]


def test_grouped_topk_vs_reference():
    """ids as sets (torch.topk(sorted=False) leaves order open), weights 1e-6 rel."""
    n = 0
    for i, c in load_golden("grouped_topk.npz"):
        m, e, k, ng, tg, renorm, scoring, has_bias = [int(v) for v in c["meta"]]
        w, ids = orc.grouped_topk(c["logits"], k, ng, tg, bias=c.get("bias"), scoring=scoring,
                                  renormalize=bool(renorm), routed_scaling=float(c["rsf"]))
        for r in range(m):
            o1, o2 = np.argsort(ids[r], kind="stable"), np.argsort(c["ids"][r], kind="stable")
            np.testing.assert_array_equal(ids[r][o1], c["ids"][r][o2], err_msg=f"case {i} row {r}")
            np.testing.assert_allclose(w[r][o1], c["w"][r][o2], rtol=3e-6, atol=1e-7)
        n += 1
    assert n > 100


def test_expert_map_vs_reference():
    for i, c in load_golden("expert_map.npz"):
        ep, r, E, strat, nloc = [int(v) for v in c["meta"]]
        n, emap = orc.expert_map(ep, r, E, strat)
        assert n == nloc
        np.testing.assert_array_equal(emap, c["map"])


def test_sort_matches_stable_argsort():
    """torch.sort(stable=True) order of tests/kernels/moe/test_moe_permute_unpermute.py:52-55."""
    rng = np.random.default_rng(1)
    for n, E in ((0, 4), (1, 1), (64, 8), (1000, 7), (4096, 256)):
        ids = rng.integers(-1, E + 1, size=n).astype(np.int32)     # includes -1 and E (invalid)
        counts, offsets, sorted_slot, pos = orc.sort_slots(ids, E)
        valid = (ids >= 0) & (ids < E)
        order = np.argsort(np.where(valid, ids, E + 1), kind="stable")[: valid.sum()]
        np.testing.assert_array_equal(sorted_slot[: valid.sum()], order)
        assert (sorted_slot[valid.sum():] == -1).all()
        np.testing.assert_array_equal(counts, np.bincount(ids[valid], minlength=E))
        np.testing.assert_array_equal(offsets, np.concatenate([[0], np.cumsum(counts)]))
        inv = np.full(n, -1, np.int32)
        inv[order] = np.arange(valid.sum())
        np.testing.assert_array_equal(pos, inv)


def _dense_desc(c, round_gemm1):
    m, n, k, e, topk, dt, act = [int(v) for v in c["meta"]]
    return orc.MoeDesc(E=e, H=k, I=n, activation=orc.ACT_SILU if act == 0 else orc.ACT_SWIGLUOAI,
                       act_dtype=dt, wfmt=orc.W_BF16 if dt == orc.BF16 else orc.W_F16,
                       round_gemm1=round_gemm1), dt


def test_moe_dense_vs_reference_cpu_oracle():
    """ref_fused_moe; default tolerances tests/kernels/allclose_default.py:8-9
    (bf16: atol 1e-3 rtol 1.6e-2; fp16: atol 1e-3 rtol 1e-3)."""
    for i, c in load_golden("moe_dense.npz"):
        d, dt = _dense_desc(c, False)
        out = orc.moe(d, c["w1"], c["w2"], c["a"], c["ids"], c["tw"])
        ref = orc.bits_to_f32(c["out_cpu"], dt)
        got = orc.bits_to_f32(orc.f32_to_bits(out, dt), dt)
        rtol = 1.6e-2 if dt == orc.BF16 else 1e-3
        np.testing.assert_allclose(got, ref, atol=1e-3, rtol=rtol, err_msg=f"case {i}")


def test_moe_dense_vs_reference_gpu_operator_oracle():
    """torch_experts; atol=2e-2 rtol=0: tests/kernels/moe/test_moe.py:233-234."""
    n = 0
    for i, c in load_golden("moe_dense.npz"):
        if "out_gpu" not in c:
            continue
        for rg in (True, False):        # both rounding conventions sit inside the reference's tolerance
            d, dt = _dense_desc(c, rg)
            out = orc.moe(d, c["w1"], c["w2"], c["a"], c["ids"], c["tw"])
            ref = orc.bits_to_f32(c["out_gpu"], dt)
            np.testing.assert_allclose(out, ref, atol=2e-2, rtol=0, err_msg=f"case {i} rg={rg}")
        n += 1
    assert n >= 4


def test_int4_quantiser_and_moe_vs_reference():
    """quantize_weights(uint4b8) packing + scales bit-exact; MoE on the dequantised weights within
    the reference's atol=2e-2 (test_moe.py:565-693)."""
    for i, c in load_golden("moe_int4.npz"):
        m, n, k, e, topk, g, dt = [int(v) for v in c["meta"]]
        q1, s1 = orc.quant_int4(c["w1"], dt, g)
        q2, s2 = orc.quant_int4(c["w2"], dt, g)
        np.testing.assert_array_equal(s1, c["s1"])
        np.testing.assert_array_equal(s2, c["s2"])
        np.testing.assert_array_equal(q1, c["q1"])
        np.testing.assert_array_equal(q2, c["q2"])
        d = orc.MoeDesc(E=e, H=k, I=n, act_dtype=dt, wfmt=orc.W_INT4, groupN=1, groupK=g)
        out = orc.moe(d, c["q1"], c["q2"], c["a"], c["ids"], c["tw"], s13=c["s1"], s2=c["s2"])
        ref = orc.bits_to_f32(c["out"], dt)
        rtol = 1.6e-2 if dt == orc.BF16 else 1e-3
        np.testing.assert_allclose(orc.bits_to_f32(orc.f32_to_bits(out, dt), dt), ref, atol=1e-3,
                                   rtol=rtol, err_msg=f"case {i}")
        if "ref1" in c:     # dequantised weights themselves, bit-exact
            dq = orc.MoeDesc(E=e, H=k, I=n, act_dtype=dt, wfmt=orc.W_BF16 if dt == orc.BF16 else orc.W_F16)
            out2 = orc.moe(dq, c["ref1"], c["ref2"], c["a"], c["ids"], c["tw"])
            np.testing.assert_array_equal(out2, out)


def test_wna16_zero_point_and_8bit_dequant_vs_reference():
    """The has_zp x weight_bits grid of test_fused_moe_wn16 (tests/kernels/moe/test_moe.py:565-693): the oracle's
    dequantisation ((q - zp) * s, fused_moe.py:207-276) equals quantize_weights' w_ref BIT FOR BIT for uint4 + zp,
    uint4b8, uint8 + zp and uint8b128, and the MoE on those weights is inside the test's atol 2e-2 of torch_moe."""
    seen = set()
    for i, c in load_golden("moe_wna16.npz"):
        m, n, k, e, topk, g, has_zp, bits = [int(v) for v in c["meta"]]
        d1 = orc.dequant_wna16(c["q1"], c["s1"], c["z1"] if has_zp else None, bits, g, orc.BF16)
        d2 = orc.dequant_wna16(c["q2"], c["s2"], c["z2"] if has_zp else None, bits, g, orc.BF16)
        if "ref1" in c:
            np.testing.assert_array_equal(d1, c["ref1"], err_msg=f"case {i}")
            np.testing.assert_array_equal(d2, c["ref2"], err_msg=f"case {i}")
            seen.add((has_zp, bits))
        if not has_zp and bits == 4:      # the symmetric 4-bit case is also the native packed format's dequantisation
            np.testing.assert_array_equal(d1, orc.dequant_rows(orc.W_INT4, orc.BF16, c["q1"], c["s1"], k, g))
        out = orc.moe(orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16), d1, d2, c["a"], c["ids"], c["tw"])
        np.testing.assert_allclose(out, orc.bits_to_f32(c["out"], orc.BF16), atol=2e-2, rtol=0, err_msg=f"case {i}")
    assert seen == {(0, 4), (1, 4), (0, 8), (1, 8)}


def test_int4_scale_on_partial_sums_stays_inside_the_reference_tolerance():
    """Checker for a candidate kernel mode (DESIGN_history.md 6): group scale applied to fp32 partial sums = the weight (q-8)*s
    kept unrounded.  Not the reference's semantics, but inside its int4 tolerance (atol 2e-2, test_moe.py:565-693)
    against the reference's own golden outputs, and close to the rounded-weight result."""
    n = 0
    for i, c in load_golden("moe_int4.npz"):
        m, nn, k, e, topk, g, dt = [int(v) for v in c["meta"]]
        kw = dict(E=e, H=k, I=nn, act_dtype=dt, wfmt=orc.W_INT4, groupN=1, groupK=g)
        a = orc.moe(orc.MoeDesc(**kw), c["q1"], c["q2"], c["a"], c["ids"], c["tw"], s13=c["s1"], s2=c["s2"])
        b = orc.moe(orc.MoeDesc(int4_unrounded=True, **kw), c["q1"], c["q2"], c["a"], c["ids"], c["tw"], s13=c["s1"],
                    s2=c["s2"])
        ref = orc.bits_to_f32(c["out"], dt)
        np.testing.assert_allclose(b, ref, atol=2e-2, rtol=0, err_msg=f"case {i}")
        assert np.abs(a - b).max() <= 5e-3 * max(1.0, np.abs(a).max())
        assert not np.array_equal(a, b)
        n += 1
    assert n > 0


def test_fp4_dequant_and_moe_vs_reference():
    """MXFP4 (dq_mxfp4_torch) and NVFP4 (dequantize_nvfp4_to_dtype) dequantisation bit-exact; MoE on
    those weights vs the reference's CPU oracle with its own tolerance (allclose_default.py:8-9)."""
    seen = set()
    for i, c in load_golden("moe_fp4.npz"):
        m, n, k, e, topk, fmt, dt = [int(v) for v in c["meta"]]
        wf, g = (orc.W_MXFP4, 32) if fmt == 0 else (orc.W_NVFP4, 16)
        gs1, gs2 = (c.get("gs1"), c.get("gs2")) if fmt == 1 else (None, None)
        if "w1" in c:
            np.testing.assert_array_equal(orc.dequant_rows(wf, dt, c["q1"], c["s1"], k, g, gs=gs1), c["w1"])
            np.testing.assert_array_equal(orc.dequant_rows(wf, dt, c["q2"], c["s2"], n, g, gs=gs2), c["w2"])
            seen.add((fmt, dt))
        d = orc.MoeDesc(E=e, H=k, I=n, act_dtype=dt, wfmt=wf, groupN=1, groupK=g)
        out = orc.moe(d, c["q1"], c["q2"], c["a"], c["ids"], c["tw"], s13=c["s1"], s2=c["s2"], gs13=gs1, gs2=gs2)
        ref = orc.bits_to_f32(c["out"], dt)
        rtol = 1.6e-2 if dt == orc.BF16 else 1e-3
        np.testing.assert_allclose(orc.bits_to_f32(orc.f32_to_bits(out, dt), dt), ref,
                                   atol=1e-3 * max(1.0, float(np.abs(ref).max())), rtol=rtol, err_msg=f"case {i}")
    assert seen == {(0, orc.BF16), (0, orc.F16), (1, orc.BF16), (1, orc.F16)}


def test_fp8_block_w8a8_vs_reference():
    """torch_w8a8_block_fp8_moe; tol 0.035: tests/kernels/moe/test_block_fp8.py:205-207."""
    for i, c in load_golden("moe_fp8_block.npz"):
        m, n, k, e, topk = [int(v) for v in c["meta"]]
        d = orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128,
                        round_gemm1=True, w8a8=True)
        out = orc.moe(d, c["w1"], c["w2"], c["a"], c["ids"], c["tw"], s13=c["w1s"], s2=c["w2s"])
        ref = orc.bits_to_f32(c["out"], orc.BF16)
        np.testing.assert_allclose(out, ref, atol=0.035, rtol=0.035, err_msg=f"case {i}")
        # fp8 W8A16 (lk_moe semantics, unpinned): must agree with W8A8 up to activation-quant noise
        d16 = orc.MoeDesc(E=e, H=k, I=n, act_dtype=orc.BF16, wfmt=orc.W_FP8, groupN=128, groupK=128)
        out16 = orc.moe(d16, c["w1"], c["w2"], c["a"], c["ids"], c["tw"], s13=c["w1s"], s2=c["w2s"])
        np.testing.assert_allclose(out16, ref, atol=0.035, rtol=0.035)


def test_router_logits_vs_f_linear_fp32():
    """the router-GEMM oracle vs the reference's own baseline `F.linear(a.float(), b.float())`
    (tests/kernels/test_fp32_router_gemm.py:35-37), tolerance ATOL_FP32 = 2e-4 (:24)."""
    import torch
    for (M, H, E, xdt, wdt) in ((5, 3072, 256, torch.bfloat16, torch.float32), (32, 6144, 128, torch.bfloat16, torch.bfloat16),
                                (3, 1024, 8, torch.float16, torch.float16)):
        g = torch.Generator().manual_seed(M)
        a = (torch.randn((M, H), generator=g) / 4).to(xdt)
        b = (torch.randn((E, H), generator=g) / 8).to(wdt)
        ref = torch.nn.functional.linear(a.float(), b.float()).numpy()
        xd = orc.BF16 if xdt == torch.bfloat16 else orc.F16
        ab = a.view(torch.int16).numpy().view(np.uint16)
        bb, wd = (b.numpy(), orc.F32) if wdt == torch.float32 else (b.view(torch.int16).numpy().view(np.uint16), xd)
        out = orc.router_logits(ab, xd, bb, wd)
        np.testing.assert_allclose(out, ref, atol=2e-4, rtol=0)


def test_operator_forms_of_the_scatter_step_match_the_reference_goldens():
    """oracle.moe_align_block_size / moe_permute / moe_unpermute against the vectors the reference's own golden functions
    produced (tests/golden/make_golden_moe_ops.py): index outputs exact, the weighted sum within one bf16 ulp of the
    reference's torch sum (its fp32 accumulation order is torch's, ours slot order)."""
    from pathlib import Path
    g = np.load(Path(__file__).resolve().parent / "golden" / "moe_ops.npz")
    for i in range(int(g["n_align"])):
        bs, E, pad = (int(v) for v in g[f"align{i}_args"])
        s, e, total = orc.moe_align_block_size(g[f"align{i}_ids"], bs, E, None, bool(pad))
        gs, ge = g[f"align{i}_sorted"], g[f"align{i}_experts"]
        n, nb = min(s.size, gs.size), min(e.size, ge.size)
        assert total == int(g[f"align{i}_post"][0])
        np.testing.assert_array_equal(s[:n], gs[:n])
        np.testing.assert_array_equal(e[:nb], ge[:nb])
    for i in range(int(g["n_alignm"])):
        emap = g[f"alignm{i}_map"]
        s, e, total = orc.moe_align_block_size(g[f"alignm{i}_ids"], 64, emap.size, emap)
        assert total == int(g[f"alignm{i}_post"][0])
        n, nb = min(s.size, g[f"alignm{i}_sorted"].size), min(e.size, g[f"alignm{i}_experts"].size)
        np.testing.assert_array_equal(s[:n], g[f"alignm{i}_sorted"][:n])
        np.testing.assert_array_equal(e[:nb], g[f"alignm{i}_experts"][:nb])
    for i in range(int(g["n_perm"])):
        E, n_local, ep, rank = (int(v) for v in g[f"perm{i}_args"])
        emap = g[f"perm{i}_map"] if ep != 1 else None
        first, inv, perm = orc.moe_permute(g[f"perm{i}_ids"], E, n_local, emap)
        np.testing.assert_array_equal(first, g[f"perm{i}_first"])
        np.testing.assert_array_equal(inv, g[f"perm{i}_inv"].reshape(-1))
        np.testing.assert_array_equal(perm, g[f"perm{i}_perm"])
        out = orc.moe_unpermute(g[f"perm{i}_res0"].view(np.uint16), orc.BF16, g[f"perm{i}_tw"], inv, int(g[f"perm{i}_nvalid"]))
        a, b = orc.bits_to_f32(out, orc.BF16), orc.bits_to_f32(g[f"perm{i}_gold4"].view(np.uint16), orc.BF16)
        assert (np.abs(a - b) <= np.maximum(np.abs(b) * 2.0 ** -7, 2.0 ** -20)).all()
