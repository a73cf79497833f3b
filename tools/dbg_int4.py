import sys; sys.path.insert(0, '.')
import numpy as np, torch
from oracle import oracle as orc
from tests.helpers import bits_to_torch, torch_to_bits, make_routing
from tests.test_gpu_moe import _rand_case, _eng, _run_decode
for (M,E,K,H,I,g) in ((160,4,2,256,256,64),):
  for mode in ("normal", "same_tokens", "same_wrows"):
    a, w13, w2, tw, ids = _rand_case(M, E, K, H, I, torch.bfloat16, seed=21)
    if mode == "same_tokens": a = a[:1].repeat(M, 1).contiguous()
    if mode == "same_wrows":
        w13 = w13[:, :1, :].repeat(1, 2*I, 1).contiguous(); w2 = w2[:, :1, :].repeat(1, H, 1).contiguous()
    q13, s13 = orc.quant_int4(torch_to_bits(w13), orc.BF16, g)
    q2, s2 = orc.quant_int4(torch_to_bits(w2), orc.BF16, g)
    eng = _eng(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=torch.bfloat16, fmt="int4",
               w13_scale=bits_to_torch(s13, orc.BF16), w2_scale=bits_to_torch(s2, orc.BF16), group_n=1, group_k=g)
    d = orc.MoeDesc(E=E, H=H, I=I, act_dtype=orc.BF16, wfmt=orc.W_INT4, groupN=1, groupK=g)
    ref = orc.moe(d, q13, q2, torch_to_bits(a), ids, tw, s13=s13, s2=s2)
    for cfg in (dict(tiled=64, waves=8, tmask=1), dict(tiled=64, waves=8, tmask=2), dict(tiled=64, waves=4, tmask=3)):
        eng.engine.set_tuning(**cfg)
        outs = [_run_decode(eng, a, tw, ids) for _ in range(3)]
        out = outs[0]
        bad = np.abs(out-ref) > (2e-3 + 1e-2*np.abs(ref))
        rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
        print(mode, cfg, "bad", int(bad.sum()), "rows", len(rows), "cols", len(cols), "maxerr", float(np.abs(out-ref).max()), "refmax", float(np.abs(ref).max()), "determ", all(np.array_equal(outs[0], o) for o in outs[1:]))
