"""GPU side of expert-parallel load balancing (include/lkm_eplb.h), through the C ABI:

* `eplb_map_record_kernel` bit-exact (ids AND counters) against the CPU restatement of the reference's
  kernel (oracle.eplb_map_record; base_router.py:24-97), incl. the reference's own test setup
  (tests/kernels/moe/test_routing.py:155-188: identity map, int64 maps, bool switch);
* expert images: export -> import moves an expert between slots and between engines bit for bit, for every
  weight format; a rearranged engine computes what the plain engine computes.

(Named test_zz3_* so that it runs after the parity suites of the hot path proper and after the less risky new suites.)
"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from tests.helpers import bits_to_torch, torch_to_bits

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops():
    from lvllm_amd import ops
    return ops


def _maps(E, P, seed, ranks=8):
    from lvllm_amd import eplb
    rng = np.random.default_rng(seed)
    w = rng.random((1, E)).astype(np.float32) ** 4
    p2l = eplb.rebalance_experts(w, P, 1, 1, ranks)
    l2p, cnt = eplb.compute_logical_maps(p2l, E, max_slots=P - E + 1)
    return p2l[0].numpy(), l2p[0].numpy().astype(np.int32), cnt[0].numpy().astype(np.int32)


@pytest.mark.parametrize("M,K,E,P", [(1, 8, 128, 144), (32, 2, 8, 16), (257, 6, 64, 72), (8192, 8, 256, 288),
                                     (700, 4, 2048, 2304)])          # the last: more physical experts than the LDS histogram holds
def test_map_record_bit_exact(M, K, E, P):
    ops = _ops()
    _, l2p, cnt = _maps(E, P, seed=M + E)
    rng = np.random.default_rng(M * 7 + K)
    ids = rng.integers(-1, E + 1, size=(M, K)).astype(np.int32)      # -1 and one id past the end included
    hot = int(np.argmax(cnt))
    ids[rng.random((M, K)) < 0.3] = hot                              # a hot expert: contended counters
    base = rng.integers(0, 1000, size=P).astype(np.int32)
    t_ids, t_l2p, t_cnt = (torch.from_numpy(a).to(DEV) for a in (ids, l2p, cnt))
    for enabled, unpadded in [(1, None), (1, max(0, M - 3)), (0, None), (1, 0)]:
        load = torch.from_numpy(base.copy()).to(DEV)
        sw = torch.tensor(enabled, dtype=torch.int32, device=DEV)
        nu = None if unpadded is None else torch.tensor(unpadded, dtype=torch.int32, device=DEV)
        got = ops.eplb_map_to_physical_and_record(t_ids, load, t_l2p, t_cnt, sw, nu)
        want, want_load = orc.eplb_map_record(ids, l2p, cnt, base, bool(enabled), unpadded)
        assert got.dtype == t_ids.dtype
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        np.testing.assert_array_equal(load.cpu().numpy(), want_load)
    # map only (no counters)
    got = ops.eplb_map_to_physical_and_record(t_ids, None, t_l2p, t_cnt)
    np.testing.assert_array_equal(got.cpu().numpy(), orc.eplb_map_record(ids, l2p, cnt)[0])


def test_map_record_reference_test_setup_and_dtypes():
    """identity map / one replica / int64 maps / bool switch / int64 ids, as the reference's test builds them"""
    ops = _ops()
    E = 64
    rng = np.random.default_rng(2)
    ids = torch.from_numpy(rng.integers(0, E, size=(33, 6))).to(DEV)                     # int64
    load = torch.zeros(E, dtype=torch.int32, device=DEV)
    l2p = torch.arange(E, dtype=torch.int64, device=DEV).unsqueeze(-1)
    cnt = torch.ones(E, dtype=torch.int64, device=DEV)
    out = ops.eplb_map_to_physical_and_record(ids, load, l2p, cnt, torch.ones((), dtype=torch.bool, device=DEV),
                                              torch.tensor(33, dtype=torch.int32, device=DEV))
    assert out.dtype == torch.int64 and torch.equal(out, ids)
    np.testing.assert_array_equal(load.cpu().numpy(), np.bincount(ids.cpu().numpy().reshape(-1), minlength=E))
    with pytest.raises(ValueError):
        ops.eplb_map_to_physical_and_record(ids, load.to(torch.int64), l2p, cnt)
    with pytest.raises(ValueError):
        ops.eplb_map_to_physical_and_record(ids.cpu(), load, l2p, cnt)
    empty = torch.empty((0, 6), dtype=torch.int32, device=DEV)
    assert ops.eplb_map_to_physical_and_record(empty, load, l2p, cnt).numel() == 0


def test_map_record_in_place_and_in_a_graph():
    """out may alias the ids (C ABI); inside a captured graph the device-side switch stays live"""
    from lvllm_amd import _clib
    lib = _clib.lib()
    E, P, M, K = 16, 24, 50, 4
    _, l2p, cnt = _maps(E, P, seed=9)
    rng = np.random.default_rng(4)
    ids = rng.integers(0, E, size=(M, K)).astype(np.int32)
    want, want_load = orc.eplb_map_record(ids, l2p, cnt, np.zeros(P, np.int32))
    t_l2p, t_cnt = torch.from_numpy(l2p).to(DEV), torch.from_numpy(cnt).to(DEV)
    buf = torch.from_numpy(ids.copy()).to(DEV)
    load = torch.zeros(P, dtype=torch.int32, device=DEV)
    sw = torch.ones((), dtype=torch.int32, device=DEV)
    vp = lambda t: C.c_void_p(t.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    _clib.check(lib.lkm_eplb_map_record(C.c_void_p(s or None), vp(buf), M * K, K, vp(t_l2p), vp(t_cnt), E, l2p.shape[1],
                                        vp(load), P, vp(sw), None, vp(buf)))
    np.testing.assert_array_equal(buf.cpu().numpy(), want)
    np.testing.assert_array_equal(load.cpu().numpy(), want_load)
    # graph: ids -> out, counters accumulate per replay while the switch is on
    src = torch.from_numpy(ids).to(DEV)
    out = torch.empty_like(src)
    load.zero_()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        _clib.check(lib.lkm_eplb_map_record(C.c_void_p(torch.cuda.current_stream().cuda_stream), vp(src), M * K, K,
                                            vp(t_l2p), vp(t_cnt), E, l2p.shape[1], vp(load), P, vp(sw), None, vp(out)))
    g.replay(); g.replay()
    sw.zero_()
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    np.testing.assert_array_equal(load.cpu().numpy(), 2 * want_load)


def test_map_record_errors_are_loud():
    from lvllm_amd import _clib
    lib = _clib.lib()
    t = torch.zeros(8, dtype=torch.int32, device=DEV)
    vp = lambda x: C.c_void_p(x.data_ptr())
    with pytest.raises(_clib.LkmError):
        _clib.check(lib.lkm_eplb_map_record(None, vp(t), 8, 0, vp(t), vp(t), 4, 1, None, 0, None, None, vp(t)))
    with pytest.raises(_clib.LkmError):
        _clib.check(lib.lkm_eplb_map_record(None, vp(t), 8, 2, None, vp(t), 4, 1, None, 0, None, None, vp(t)))
    with pytest.raises(_clib.LkmError):
        _clib.check(lib.lkm_eplb_map_record(None, vp(t), 8, 2, vp(t), vp(t), 4, 1, vp(t), 0, None, None, vp(t)))


# ------------------------------------------------------------------------------------------ expert images
def _engine(fmt, E, H, I, K, seed):
    """engine of E experts in `fmt` from seeded master weights -> (engine, per-expert constructor inputs)"""
    from lvllm_amd.ops import RoutedExpertsEngine
    g = torch.Generator().manual_seed(seed)
    dt = torch.float16 if fmt == "f16" else torch.bfloat16
    odt = orc.F16 if fmt == "f16" else orc.BF16
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(dt)
    w2 = (torch.randn((E, H, I), generator=g) / 4).to(dt)
    kw = {}
    if fmt in ("bf16", "f16"):
        parts = dict(w13=w13, w2=w2)
    elif fmt == "int4":
        q13, s13 = orc.quant_int4(torch_to_bits(w13), odt, 64)
        q2, s2 = orc.quant_int4(torch_to_bits(w2), odt, 64)
        parts = dict(w13=torch.from_numpy(q13), w2=torch.from_numpy(q2), w13_scale=bits_to_torch(s13, odt),
                     w2_scale=bits_to_torch(s2, odt))
        kw = dict(fmt="int4", group_n=1, group_k=64)
    elif fmt in ("fp8", "fp8a8"):
        q13, s13 = orc.quant_fp8_block(w13.float().numpy(), 128, 128)
        q2, s2 = orc.quant_fp8_block(w2.float().numpy(), 128, 128)
        parts = dict(w13=torch.from_numpy(q13), w2=torch.from_numpy(q2), w13_scale=torch.from_numpy(s13),
                     w2_scale=torch.from_numpy(s2))
        kw = dict(fmt="fp8", group_n=128, group_k=128, fp8_mode=1 if fmt == "fp8a8" else 0)
    else:                                                                # mxfp4 / nvfp4: random codes and scales
        r = np.random.default_rng(seed)
        gk = 32 if fmt == "mxfp4" else 16
        lo, hi = (117, 121) if fmt == "mxfp4" else (0x30, 0x40)
        parts = dict(w13=torch.from_numpy(r.integers(0, 256, (E, 2 * I, H // 2), dtype=np.uint8)),
                     w2=torch.from_numpy(r.integers(0, 256, (E, H, I // 2), dtype=np.uint8)),
                     w13_scale=torch.from_numpy(r.integers(lo, hi, (E, 2 * I, H // gk), dtype=np.uint8)),
                     w2_scale=torch.from_numpy(r.integers(lo, hi, (E, H, I // gk), dtype=np.uint8)))
        kw = dict(fmt=fmt, group_n=1, group_k=gk)
        if fmt == "nvfp4":
            parts["w13_global_scale"] = torch.from_numpy(r.random(E).astype(np.float32) + 0.5)
            parts["w2_global_scale"] = torch.from_numpy(r.random(E).astype(np.float32) + 0.5)
    eng = RoutedExpertsEngine(**{k: v.to(DEV) for k, v in parts.items()}, top_k=K, act_dtype=dt, **kw)
    return eng, parts, kw, dt


def _rows(M, H, dt, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn((M, H), generator=g) / 2).to(dt).to(DEV)


def _decode_all_on(eng, x, expert):
    M = x.size(0)
    ids = torch.full((M, 1), expert, dtype=torch.int32, device=DEV)
    tw = torch.ones((M, 1), dtype=torch.float32, device=DEV)
    return eng.decode(x, tw, ids).clone()


@pytest.mark.parametrize("fmt", ["bf16", "f16", "int4", "fp8", "fp8a8", "mxfp4", "nvfp4"])
def test_expert_image_moves_an_expert_bit_for_bit(fmt):
    from lvllm_amd.eplb import EngineExpertStore
    E, H, I, K, M = 4, 256, 128, 1, 19
    eng, _, _, dt = _engine(fmt, E, H, I, K, seed=31)
    other, _, _, _ = _engine(fmt, E, H, I, K, seed=32)                   # same configuration, other weights
    st, st2 = EngineExpertStore(eng), EngineExpertStore(other)
    assert st.num_local == E and st.expert_nbytes == st2.expert_nbytes and st.expert_nbytes % 16 == 0
    assert st.expert_nbytes * E >= eng.engine.weight_bytes() - 64        # the images cover the weight buffers
    x = _rows(M, H, dt, 5)
    before = [_decode_all_on(eng, x, e) for e in range(E)]
    assert not torch.equal(before[1], before[2])
    img = torch.empty(st.expert_nbytes, dtype=torch.uint8, device=DEV)
    # within one engine: slot 1 -> slot 2; the others are untouched
    st.export_expert(1, img)
    st.import_expert(2, img)
    after = [_decode_all_on(eng, x, e) for e in range(E)]
    assert torch.equal(after[2], before[1])
    for e in (0, 1, 3):
        assert torch.equal(after[e], before[e])
    # across engines: expert 3 of `eng` into slot 0 of `other`
    st.export_expert(3, img)
    st2.import_expert(0, img)
    assert torch.equal(_decode_all_on(other, x, 0), before[3])
    # export after import returns the image that was imported
    img2 = torch.empty_like(img)
    st2.export_expert(0, img2)
    assert torch.equal(img, img2)
    with pytest.raises(Exception):
        st.export_expert(E, img)
    with pytest.raises(ValueError):
        st.export_expert(0, img[:-16])


def test_rearranged_engine_computes_what_the_plain_engine_computes():
    """one rank, 8 logical experts in 12 physical slots: skewed load -> policy -> local moves of expert images ->
    maps; routing logical ids through the id map into the rearranged engine == the plain 8-expert engine"""
    from lvllm_amd import eplb, ops
    from lvllm_amd.ops import RoutedExpertsEngine
    E, red, H, I, K, M = 8, 4, 256, 128, 2, 64
    P = E + red
    g = torch.Generator().manual_seed(77)
    w13 = (torch.randn((E, 2 * I, H), generator=g) / 4).to(torch.bfloat16)
    w2 = (torch.randn((E, H, I), generator=g) / 4).to(torch.bfloat16)
    plain = RoutedExpertsEngine(w13.to(DEV), w2.to(DEV), top_k=K, act_dtype=torch.bfloat16)
    st = eplb.EplbState(1, E, red, window_size=2, step_interval=2, device=DEV)
    init = st.physical_to_logical_map[0]
    phys = RoutedExpertsEngine(w13[init].to(DEV), w2[init].to(DEV), top_k=K, act_dtype=torch.bfloat16)
    st.expert_stores = [eplb.EngineExpertStore(phys)]
    x = _rows(M, H, torch.bfloat16, 8)
    logits = torch.randn((M, E), generator=g)
    logits[:, 3] += 3.0                                                   # expert 3 is hot
    tw, ids = ops.topk_softmax(logits.to(DEV), K, True)
    want = plain.decode(x, tw, ids).cpu().numpy()

    def routed():
        ls = st.layer_state(0)
        pid = ops.eplb_map_to_physical_and_record(ids, ls.expert_load_view, ls.logical_to_physical_map,
                                                  ls.logical_replica_count, ls.should_record_tensor)
        return pid, phys.decode(x, tw, pid).cpu().numpy()

    pid0, out0 = routed()
    # same weights, same rows; the rows of an expert may be tiled differently (replicas split them), so the fp32
    # summation order -- and with it a bf16 rounding of the intermediate -- may differ: the suite's tolerance
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(out0, want, atol=2e-3 * scale, rtol=1e-2)
    cnt_before = st.logical_replica_count[0].cpu().numpy().copy()
    assert not st.step()
    routed()
    assert st.step()                                                      # second step: rearrangement ran
    cnt_after = st.logical_replica_count[0].cpu().numpy()
    assert cnt_after[3] == cnt_after.max() and cnt_after[3] >= 2 and cnt_after.sum() == P
    assert not np.array_equal(cnt_before, cnt_after) or not torch.equal(init, st.physical_to_logical_map[0])
    pid1, out1 = routed()
    np.testing.assert_allclose(out1, want, atol=2e-3 * scale, rtol=1e-2)
    # the physical ids name slots that hold the right logical expert
    p2l = st.physical_to_logical_map[0].numpy()
    np.testing.assert_array_equal(p2l[pid1.cpu().numpy()], ids.cpu().numpy())
