"""The device code of the EPLB id-map / load-recording kernel (lvllm_amd/csrc/eplb_kernel.inc -- the SAME source
hipcc compiles for gfx950) executed on the CPU under a thread-per-lane emulation (tests/emu/hip_cpu_emu.h: one
std::thread per lane, a std::barrier for __syncthreads, a static array for the LDS histogram) and compared bit for
bit with the CPU restatement of the reference's kernel.  This checks the indexing, the hash, the LDS-histogram path,
its barriers and the flush -- everything but the hardware; the real launch is tests/test_zz3_gpu_eplb.py."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from lvllm_amd import eplb
from oracle import oracle as orc

EMU = Path(__file__).resolve().parent / "emu"


@pytest.fixture(scope="module")
def emu():
    out = EMU / "_build"
    out.mkdir(exist_ok=True)
    lib = out / "libemu_eplb.so"
    srcs = [EMU / "emu_eplb.cpp", EMU / "hip_cpu_emu.h", EMU.parents[1] / "lvllm_amd" / "csrc" / "eplb_kernel.inc"]
    if not lib.exists() or lib.stat().st_mtime < max(s.stat().st_mtime for s in srcs):
        r = subprocess.run(["g++", "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-Wall", "-o", str(lib),
                            str(EMU / "emu_eplb.cpp")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    dll = C.CDLL(str(lib))
    dll.emu_eplb_map_record.restype = C.c_int
    dll.emu_eplb_map_record.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    return dll


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _maps(E, P, seed, ranks=8):
    rng = np.random.default_rng(seed)
    w = rng.random((1, E)).astype(np.float32) ** 4
    p2l = eplb.rebalance_experts(w, P, 1, 1, ranks)
    l2p, cnt = eplb.compute_logical_maps(p2l, E, max_slots=P - E + 1)
    return np.ascontiguousarray(l2p[0].numpy().astype(np.int32)), np.ascontiguousarray(cnt[0].numpy().astype(np.int32))


@pytest.mark.parametrize("M,K,E,P", [(1, 8, 128, 144), (32, 2, 8, 16), (257, 6, 64, 72), (700, 3, 256, 288)])
@pytest.mark.parametrize("variant", [1, 0])                    # LDS histogram, global atomics
def test_emulated_kernel_is_bit_exact(emu, M, K, E, P, variant):
    l2p, cnt = _maps(E, P, seed=M + E)
    rng = np.random.default_rng(M * 7 + K)
    ids = rng.integers(-1, E + 1, size=(M, K)).astype(np.int32)
    ids[rng.random((M, K)) < 0.3] = int(np.argmax(cnt))           # a hot expert: contended counters
    base = rng.integers(0, 1000, size=P).astype(np.int32)
    for enabled, unpadded in [(1, None), (1, max(0, M - 3)), (0, None), (1, 0)]:
        load, out = base.copy(), np.full((M, K), 12345, np.int32)
        sw = np.array([enabled], np.int32)
        nu = None if unpadded is None else np.array([unpadded], np.int32)
        assert emu.emu_eplb_map_record(_p(ids), M * K, K, _p(l2p), _p(cnt), E, l2p.shape[1], _p(load), P, _p(sw), _p(nu),
                                       _p(out), variant) == 0
        want, want_load = orc.eplb_map_record(ids, l2p, cnt, base, bool(enabled), unpadded)
        np.testing.assert_array_equal(out, want)
        np.testing.assert_array_equal(load, want_load)


def test_emulated_kernel_map_only_aliasing_and_variant_choice(emu):
    E, P, M, K = 16, 24, 300, 4
    l2p, cnt = _maps(E, P, seed=9)
    rng = np.random.default_rng(4)
    ids = rng.integers(0, E, size=(M, K)).astype(np.int32)
    want, _ = orc.eplb_map_record(ids, l2p, cnt)
    # no counters, no switch; out aliases the ids
    buf = ids.copy()
    emu.emu_eplb_map_record(_p(buf), M * K, K, _p(l2p), _p(cnt), E, l2p.shape[1], None, 0, None, None, _p(buf), -1)
    np.testing.assert_array_equal(buf, want)
    # more physical experts than the LDS histogram holds: the wrapper's rule picks the global-atomics variant
    E2, P2 = 2048, 2304
    l2p2, cnt2 = _maps(E2, P2, seed=3)
    ids2 = rng.integers(0, E2, size=(500, 2)).astype(np.int32)
    load, out = np.zeros(P2, np.int32), np.empty((500, 2), np.int32)
    emu.emu_eplb_map_record(_p(ids2), 1000, 2, _p(l2p2), _p(cnt2), E2, l2p2.shape[1], _p(load), P2, None, None, _p(out), -1)
    w2, wl2 = orc.eplb_map_record(ids2, l2p2, cnt2, np.zeros(P2, np.int32))
    np.testing.assert_array_equal(out, w2)
    np.testing.assert_array_equal(load, wl2)
    # an inconsistent map (replica count above the map width) stays in bounds, as the restatement documents
    bad_cnt = cnt.copy()
    bad_cnt[0] = 100
    out = np.empty((M, K), np.int32)
    emu.emu_eplb_map_record(_p(ids), M * K, K, _p(l2p), _p(bad_cnt), E, l2p.shape[1], None, 0, None, None, _p(out), -1)
    np.testing.assert_array_equal(out, orc.eplb_map_record(ids, l2p, bad_cnt)[0])
