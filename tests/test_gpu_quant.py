"""The activation quantiser of the W8A8 path (per_token_group_quant_fp8 with groups of 128, fp8_utils.py:533-660; spec
tests/kernels/quant_utils.py:157-180) against its arithmetic in IEEE fp32 on the CPU: scale = max(amax, 1e-10) / 448,
q = clamp(x / scale, +-448) -> e4m3fn (round to nearest even).  BIT-EXACT: bytes and scales.  The kernel divides with a
shared refined reciprocal and two remainder corrections per element instead of the general division sequence; this is the
test that says the two agree on every element (millions of random values, all-zero groups, values at the clamp, tiny
and huge magnitudes, ragged column counts, strided rows)."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(x32: np.ndarray):
    rows, cols = x32.shape
    kb = -(-cols // 128)
    pad = np.zeros((rows, kb * 128), np.float32)
    pad[:, :cols] = x32
    g = pad.reshape(rows, kb, 128)
    amax = np.maximum(np.abs(g).max(axis=2), np.float32(1e-10)).astype(np.float32)
    s = (amax / np.float32(448.0)).astype(np.float32)
    q = np.clip((g / s[:, :, None]).astype(np.float32), np.float32(-448.0), np.float32(448.0))
    return orc.f32_to_fp8(q.reshape(rows, kb * 128)[:, :cols].copy()), s


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,cols,kind", [(4096, 4096, "randn"), (777, 1408, "randn"), (64, 136, "randn"), (3, 8, "randn"),
                                            (2048, 2048, "wide"), (512, 1024, "edge"), (40000, 512, "randn")])
def test_per_token_group_quant_is_bit_exact(rows, cols, kind, dtype):
    from lvllm_amd import ops
    g = torch.Generator().manual_seed(rows + cols)
    if kind == "randn":
        x = torch.randn((rows, cols), generator=g) * (torch.rand((rows, 1), generator=g) * 4 + 0.01)
    elif kind == "wide":     # magnitudes over the whole range of the dtype, row by row
        x = torch.randn((rows, cols), generator=g) * torch.exp2(torch.randint(-14 if dtype == torch.float16 else -60, 15, (rows, 1), generator=g).float())
    else:                    # zero groups, single spikes, values equal to +-amax, denormal-sized groups
        x = torch.randn((rows, cols), generator=g)
        x[::7] = 0.0
        x[1::7, ::128] = 1000.0
        x[2::7] = torch.sign(x[2::7]) * 3.0
        x[3::7] *= 1e-12 if dtype == torch.bfloat16 else 1e-6
    x = x.clamp(-6.0e4, 6.0e4).to(dtype)                    # (finite in fp16: an infinite input has no defined quantisation)
    ld = cols + 24                                            # rows strided like the engine's intermediate
    buf = torch.zeros((rows, ld), dtype=dtype)
    buf[:, :cols] = x
    xd = buf.to(DEV)[:, :cols]
    q, s = ops.per_token_group_quant_fp8(xd)
    q_ref, s_ref = _ref(x.float().numpy())
    np.testing.assert_array_equal(s.cpu().numpy().view(np.int32), s_ref.view(np.int32))
    got = q.cpu().numpy()
    bad = np.flatnonzero(got.ravel() != q_ref.ravel())
    assert bad.size == 0, (bad.size, got.ravel()[bad[:5]], q_ref.ravel()[bad[:5]], x.float().numpy().ravel()[bad[:5]])


def test_wna16_expand_is_bit_exact_vs_oracle_and_reference_w_ref():
    """lkm_wna16_expand = T((q - zp) * s) (fused_moe.py:207-276) on the reference test's packings: equal to the oracle bit
    for bit on every golden case, and to quantize_weights' w_ref where the golden file carries it."""
    from lvllm_amd import ops
    from tests.helpers import bits_to_torch, load_golden
    n_ref = 0
    for i, c in load_golden("moe_wna16.npz"):
        m, n, k, e, topk, g, has_zp, bits = [int(v) for v in c["meta"]]
        for w in ("1", "2"):
            q, s_ = torch.from_numpy(c["q" + w]).to(DEV), bits_to_torch(c["s" + w], orc.BF16).to(DEV)
            z = torch.from_numpy(c["z" + w]).to(DEV) if has_zp else None
            got = ops.wna16_expand(q, s_, z, bits, g).cpu().view(torch.int16).numpy().view(np.uint16)
            np.testing.assert_array_equal(got, orc.dequant_wna16(c["q" + w], c["s" + w], c["z" + w] if has_zp else None,
                                                                 bits, g, orc.BF16), err_msg=f"case {i} w{w}")
            if "ref" + w in c:
                np.testing.assert_array_equal(got, c["ref" + w], err_msg=f"case {i} w{w} vs w_ref")
                n_ref += 1
    assert n_ref == 8
    # fp16 scales, group 32, odd sizes of the expert count
    g_ = torch.Generator().manual_seed(9)
    for bits in (4, 8):
        E, N, K, grp = 3, 64, 96, 32
        q = torch.randint(0, 256, (E, N, K // 2 if bits == 4 else K), generator=g_, dtype=torch.uint8)
        s_ = (torch.rand((E, N, K // grp), generator=g_) * 0.02 + 1e-3).to(torch.float16)
        z = torch.randint(0, 256, (E, N // 2 if bits == 4 else N, K // grp), generator=g_, dtype=torch.uint8)
        got = ops.wna16_expand(q.to(DEV), s_.to(DEV), z.to(DEV), bits, grp).cpu().view(torch.int16).numpy().view(np.uint16)
        want = orc.dequant_wna16(q.numpy(), s_.view(torch.int16).numpy().view(np.uint16), z.numpy(), bits, grp, orc.F16)
        np.testing.assert_array_equal(got, want)
