import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import bits_to_torch
from oracle import oracle as orc
from lvllm_amd.ops import RoutedExpertsEngine
DEV = torch.device("cuda", 0)
E, H, I, K, g = 2, 256, 128, 1, 128
odt, tdt = orc.BF16, torch.bfloat16
for special in (True, False):
    rng = np.random.default_rng(11 + g)
    q13 = rng.integers(0, 256, (E, I, H // 2), dtype=np.uint8)
    if special:
        q13[0, 0, :8] = np.arange(0, 256, 32, dtype=np.uint8) + np.arange(8, dtype=np.uint8)
    s13 = (rng.uniform(0.004, 0.03, (E, I, H // g)) * rng.choice([1.0, 37.0, 0.25], (E, I, H // g))).astype(np.float32)
    s13b = orc.f32_to_bits(s13, odt)
    wd = orc.bits_to_f32(orc.dequant_rows(orc.W_INT4, odt, q13, s13b, H, g), odt)
    q2 = np.full((E, H, I // 2), 0x88, np.uint8)
    for r in range(min(H, I)):
        q2[:, r, r // 2] = 0x88 + (1 << (4 * (r & 1)))
    s2b = orc.f32_to_bits(np.ones((E, H, max(1, I // g)), np.float32), odt)
    eng = RoutedExpertsEngine(torch.from_numpy(q13), torch.from_numpy(q2), top_k=K, act_dtype=tdt, fmt="int4",
               w13_scale=bits_to_torch(s13b, odt), w2_scale=bits_to_torch(s2b, odt), group_n=1, group_k=g,
               has_gate_proj=False, activation_type=2)
    x = torch.eye(H, dtype=tdt)
    for e in range(E):
        ids = torch.full((H, 1), e, dtype=torch.int32)
        tw = torch.ones((H, 1))
        for tiled in (-1, 64, 128):
            eng.engine.set_tuning(tiled=tiled)
            for sign in (1.0, -1.0):
                out = eng.decode((x * sign).to(DEV), tw.to(DEV), ids.to(DEV)).cpu().numpy()[:, :I]
                want = np.maximum(sign * wd[e].T, 0.0) ** 2
                want = orc.bits_to_f32(orc.f32_to_bits(want.astype(np.float32), odt), odt)
                bad = np.argwhere(out != want)
                print(f"special={special} e={e} tiled={tiled} sign={sign}: {len(bad)} bad", flush=True)
                for (j, i) in bad[:20]:
                    b = q13[e, i, j // 2]; v = (b >> (4 * (j & 1))) & 15
                    print(f"   k={j} row={i} code={v} scale={s13[e, i, j // g]:.5f} got={out[j, i]:.5f} want={want[j, i]:.5f} sqrt_got={np.sqrt(out[j,i]):.4f}")
