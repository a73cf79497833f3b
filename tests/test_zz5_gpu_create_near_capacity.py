"""Engine creation near HBM capacity (runs last: it fills the GPU).  Engines of DeepSeek-V3 size (fp8 block weights,
11.3 GB = 10.5 GiB per layer) are created until the GPU is nearly full; the next one fails with LKM_E_NOMEM, leaves nothing
behind and the engines that exist keep working; with less free memory than TWICE an image a host-sourced engine still
fits (image + one staging chunk: tests/test_gpu_create_streaming.py checks the chunks build the same image)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GiB = 1 << 30


def _free():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_engines_of_deepseek_v3_size_until_the_gpu_is_full(monkeypatch):
    """BASELINE.json configs[3]'s layer (256 experts, 7168 x 2048, fp8 block 128 x 128: 11.3 GB of weights) created
    again and again from ONE set of device tensors until fewer than 12 GB are free, then with < 8 GB free once more."""
    from lvllm_amd import _clib
    from lvllm_amd.ops import RoutedExpertsEngine
    monkeypatch.delenv("LKM_STAGE_BYTES", raising=False)
    E, K, H, I = 256, 8, 7168, 2048
    torch.cuda.empty_cache()
    g = torch.Generator(device=DEV).manual_seed(5)
    w13 = torch.randint(0, 0x70, (E, 2 * I, H), generator=g, dtype=torch.uint8, device=DEV)
    w2 = torch.randint(0, 0x70, (E, H, I), generator=g, dtype=torch.uint8, device=DEV)
    kw = dict(top_k=K, act_dtype=torch.bfloat16, fmt="fp8", group_n=128, group_k=128, max_num_seqs=64, max_batch_size=64,
              w13_scale=torch.rand((E, 2 * I // 128, H // 128), generator=g, device=DEV) / 512 + 1e-4,
              w2_scale=torch.rand((E, H // 128, I // 128), generator=g, device=DEV) / 512 + 1e-4)
    x = (torch.randn((16, H), generator=g, device=DEV) / 4).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(E, generator=g, device=DEV)[:K] for _ in range(16)]).to(torch.int32)
    tw = torch.rand((16, K), generator=g, device=DEV)
    start = _free()
    engines = [RoutedExpertsEngine(w13, w2, **kw)]
    per_engine = start - _free()
    assert 10.4 * GiB < per_engine < 11.5 * GiB, per_engine / GiB          # the image and its scratch, nothing staged
    first = engines[0].decode(x, tw, ids).cpu().numpy()
    assert np.isfinite(first).all() and np.abs(first).max() > 0
    while _free() >= per_engine + 1 * GiB:
        engines.append(RoutedExpertsEngine(w13, w2, **kw))
    assert len(engines) >= 15                                                # (288 GB: 22-23 of them next to the source)
    filler = None
    if _free() > 8 * GiB:
        filler = torch.empty((_free() - 6 * GiB,), dtype=torch.uint8, device=DEV)
    before = _free()
    assert before < 8 * GiB
    with pytest.raises(_clib.LkmError) as ei:
        RoutedExpertsEngine(w13, w2, **kw)
    assert ei.value.code == _clib.E_NOMEM, ei.value
    assert _free() > before - (64 << 20)                                     # the failed attempt left nothing behind (the runtime may
                                                                             # hand back deferred frees of its own: more is fine)
    # every engine created so far holds the same image and still answers with the same bits
    for eng in (engines[0], engines[-1]):
        np.testing.assert_array_equal(eng.decode(x, tw, ids).cpu().numpy(), first)

    # a host-sourced engine fits where image + one staging chunk fit (2.8 GB image, < 2 x that free)
    del engines[-1], eng
    torch.cuda.empty_cache()
    Es = 64
    h13, h2 = w13[:Es].cpu(), w2[:Es].cpu()
    hkw = dict(kw, w13_scale=kw["w13_scale"][:Es].cpu(), w2_scale=kw["w2_scale"][:Es].cpu())
    image = per_engine * Es // E
    room = image + image // 2                                               # 1.5 x the image: too little for image + whole source
    filler2 = torch.empty((max(_free() - room, 1),), dtype=torch.uint8, device=DEV)
    assert _free() < 2 * image
    small = RoutedExpertsEngine(h13, h2, **hkw)
    ids_s = torch.stack([torch.randperm(Es, generator=g, device=DEV)[:K] for _ in range(16)]).to(torch.int32)
    out_small = small.decode(x, tw, ids_s).cpu().numpy()
    del filler2, small
    torch.cuda.empty_cache()
    dev_small = RoutedExpertsEngine(w13[:Es], w2[:Es], **dict(kw, w13_scale=kw["w13_scale"][:Es], w2_scale=kw["w2_scale"][:Es]))
    np.testing.assert_array_equal(out_small, dev_small.decode(x, tw, ids_s).cpu().numpy())
    del dev_small, engines, filler
    torch.cuda.empty_cache()
    assert start - _free() < 256 << 20                                       # destroying the engines returned the memory
