"""lkm_create hands the caller's tensors over without doubling the footprint (the reference frees its
tensors right after the constructor returns, routed_experts.py:1420-1432):

  * host sources go through ONE staging buffer of <= LKM_STAGE_BYTES in chunks of whole experts; the image they produce
    is the image the device-pointer path produces (same decode bits), whatever the chunking;
  * engines of DeepSeek-V3 size (fp8 block weights, 11.3 GB = 10.5 GiB per layer) are created until the GPU is nearly full; the
    next one fails with LKM_E_NOMEM, leaves nothing behind and the engines that exist keep working;
  * with less free memory than TWICE an image, a host-sourced engine still fits (image + one staging chunk).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GiB = 1 << 30


def _rand_fp8(shape, g):
    b = torch.randint(0, 0x78, shape, generator=g, dtype=torch.uint8)          # finite e4m3 magnitudes
    return b | (torch.randint(0, 2, shape, generator=g, dtype=torch.uint8) << 7)


def _case(fmt, E, H, I, g):
    """random weights of one format on the HOST: (w13, w2, constructor keywords)"""
    from lvllm_amd import _clib
    if fmt == "bf16":
        return (torch.randn((E, 2 * I, H), generator=g) / 8).to(torch.bfloat16), \
               (torch.randn((E, H, I), generator=g) / 8).to(torch.bfloat16), dict(fmt="bf16")
    if fmt in ("int4", "int4fast"):
        gk = 32 if fmt == "int4" else 128
        kw = dict(fmt="int4", group_n=1, group_k=gk,
                  w13_scale=(torch.rand((E, 2 * I, H // gk), generator=g) / 64 + 1e-3).to(torch.bfloat16),
                  w2_scale=(torch.rand((E, H, I // gk), generator=g) / 64 + 1e-3).to(torch.bfloat16))
        if fmt == "int4fast":
            kw["int4_mode"] = _clib.INT4_FAST
        return torch.randint(0, 256, (E, 2 * I, H // 2), generator=g, dtype=torch.uint8), \
            torch.randint(0, 256, (E, H, I // 2), generator=g, dtype=torch.uint8), kw
    if fmt in ("fp8", "fp8a8"):
        kw = dict(fmt="fp8", group_n=128, group_k=128,
                  w13_scale=torch.rand((E, -(-2 * I // 128), -(-H // 128)), generator=g) / 256 + 1e-4,
                  w2_scale=torch.rand((E, -(-H // 128), -(-I // 128)), generator=g) / 256 + 1e-4)
        if fmt == "fp8a8":
            kw["fp8_mode"] = _clib.FP8_W8A8
        return _rand_fp8((E, 2 * I, H), g), _rand_fp8((E, H, I), g), kw
    gk = 32 if fmt == "mxfp4" else 16
    lo, hi = (118, 124) if fmt == "mxfp4" else (0x20, 0x38)                     # E8M0 / e4m3 scale bytes
    kw = dict(fmt=fmt, group_n=1, group_k=gk,
              w13_scale=torch.randint(lo, hi, (E, 2 * I, H // gk), generator=g, dtype=torch.uint8),
              w2_scale=torch.randint(lo, hi, (E, H, I // gk), generator=g, dtype=torch.uint8))
    if fmt == "nvfp4":
        kw.update(w13_global_scale=torch.rand((E,), generator=g) + 0.5, w2_global_scale=torch.rand((E,), generator=g) + 0.5)
    return torch.randint(0, 256, (E, 2 * I, H // 2), generator=g, dtype=torch.uint8), \
        torch.randint(0, 256, (E, H, I // 2), generator=g, dtype=torch.uint8), kw


def _to(kw, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}


@pytest.mark.parametrize("fmt", ["bf16", "int4", "int4fast", "fp8", "fp8a8", "mxfp4", "nvfp4"])
def test_host_chunks_build_the_image_the_device_path_builds(fmt, monkeypatch):
    from lvllm_amd.ops import RoutedExpertsEngine
    E, K, H, I, M = 5, 2, 512, 384, 24
    g = torch.Generator().manual_seed(11)
    w13, w2, kw = _case(fmt, E, H, I, g)
    x = (torch.randn((M, H), generator=g) / 4).to(torch.bfloat16).to(DEV)
    ids = torch.stack([torch.randperm(E, generator=g)[:K] for _ in range(M)]).to(torch.int32).to(DEV)
    tw = torch.rand((M, K), generator=g).to(DEV)
    common = dict(top_k=K, act_dtype=torch.bfloat16, max_num_seqs=64, max_batch_size=64)
    outs = {}
    for name, stage in (("device", None), ("host, one expert per chunk", "1"), ("host, two experts per chunk", None),
                        ("host, one chunk", str(1 << 28))):
        if stage is None and name.startswith("host"):
            # (the chunk never splits an expert: 2 x the largest per-expert source + 1 byte -> 2 experts per chunk)
            stage = str(2 * max(w13[0].numel() * w13.element_size(), w2[0].numel() * w2.element_size()) + 1)
        if stage is None:
            monkeypatch.delenv("LKM_STAGE_BYTES", raising=False)
            eng = RoutedExpertsEngine(w13.to(DEV), w2.to(DEV), **common, **_to(kw, DEV))
        else:
            monkeypatch.setenv("LKM_STAGE_BYTES", stage)
            eng = RoutedExpertsEngine(w13, w2, **common, **kw)
        outs[name] = eng.decode(x, tw, ids).cpu().numpy()
        del eng
    ref = outs.pop("device")
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0
    for name, o in outs.items():
        np.testing.assert_array_equal(o, ref, err_msg=f"{fmt}: {name}")
