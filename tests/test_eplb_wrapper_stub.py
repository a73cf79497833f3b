"""CPU check of the host-side marshalling of `ops.eplb_map_to_physical_and_record` (argument order, pointer /
size / dtype conversions) and of the GPU test's own logic: liblkm's entry point is replaced by a stub that reads the
raw pointers it is handed and computes the result with the CPU restatement, then the GPU parity test's body runs on
CPU tensors.  (The HIP kernel itself is only exercised by tests/test_zz3_gpu_eplb.py on an MI355X.)"""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc


class _StubLib:
    def __init__(self):
        self.calls = 0

    @staticmethod
    def _arr(ptr, n, dtype=np.int32):
        if ptr is None or (hasattr(ptr, "value") and ptr.value is None):
            return None
        addr = ptr.value if hasattr(ptr, "value") else int(ptr)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_int32)), shape=(n,))

    def lkm_eplb_map_record(self, stream, ids, numel, top_k, l2p, cnt, E, R, load, P, rec, nu, out):
        self.calls += 1
        assert numel % top_k == 0
        ids_a = self._arr(ids, numel).reshape(numel // top_k, top_k)
        l2p_a, cnt_a = self._arr(l2p, E * R).reshape(E, R), self._arr(cnt, E)
        load_a = self._arr(load, P) if P else None
        rec_a, nu_a = self._arr(rec, 1), self._arr(nu, 1)
        phys, new_load = orc.eplb_map_record(ids_a, l2p_a, cnt_a, load_a, True if rec_a is None else bool(rec_a[0]),
                                             None if nu_a is None else int(nu_a[0]))
        self._arr(out, numel)[:] = phys.reshape(-1)
        if load_a is not None:
            load_a[:] = new_load
        return 0


@pytest.fixture
def stubbed(monkeypatch):
    from lvllm_amd import _clib, ops
    import tests.test_zz3_gpu_eplb as gpu_tests
    stub = _StubLib()
    monkeypatch.setattr(_clib, "lib", lambda: stub)
    monkeypatch.setattr(ops, "_need_cuda", lambda *a: None)
    monkeypatch.setattr(ops, "_stream", lambda t: None)
    monkeypatch.setattr(gpu_tests, "DEV", "cpu")
    return stub, gpu_tests, ops


@pytest.mark.parametrize("M,K,E,P", [(1, 8, 128, 144), (32, 2, 8, 16), (257, 6, 64, 72)])
def test_wrapper_marshalling_through_a_stub_library(stubbed, M, K, E, P):
    stub, gpu_tests, _ = stubbed
    gpu_tests.test_map_record_bit_exact(M, K, E, P)
    assert stub.calls == 5


def test_wrapper_accepts_the_reference_dtypes(stubbed):
    stub, _, ops = stubbed
    E = 64
    rng = np.random.default_rng(2)
    ids = torch.from_numpy(rng.integers(0, E, size=(33, 6)))                              # int64
    load = torch.zeros(E, dtype=torch.int32)
    out = ops.eplb_map_to_physical_and_record(ids, load, torch.arange(E, dtype=torch.int64).unsqueeze(-1),
                                              torch.ones(E, dtype=torch.int64), torch.ones((), dtype=torch.bool),
                                              torch.tensor(33, dtype=torch.int32))
    assert out.dtype == torch.int64 and torch.equal(out, ids)
    np.testing.assert_array_equal(load.numpy(), np.bincount(ids.numpy().reshape(-1), minlength=E))
    with pytest.raises(ValueError):
        ops.eplb_map_to_physical_and_record(ids, load.to(torch.int64), torch.arange(E).unsqueeze(-1), torch.ones(E))
    with pytest.raises(ValueError):
        ops.eplb_map_to_physical_and_record(ids.reshape(-1), load, torch.arange(E).unsqueeze(-1), torch.ones(E))
    assert ops.eplb_map_to_physical_and_record(torch.empty((0, 6), dtype=torch.int32), load,
                                               torch.arange(E).unsqueeze(-1), torch.ones(E)).numel() == 0
