"""Layer tiers: the reference's predicates over a grid of configurations
(tests/golden/residency.json, produced by running vllm/envs.py's own functions) and the HBM planner
against the engine's real footprint."""
import json

import pytest
import torch

from lvllm_amd.residency import TierConfig, expert_layer_bytes, parse_layer_list, plan_hbm
from tests.helpers import GOLDEN


def test_tier_predicates_match_reference():
    rows = json.loads((GOLDEN / "residency.json").read_text())
    assert len(rows) == 168
    for r in rows:
        cfg = TierConfig(feature_enabled=r["on"], resident_layers=r["spec"], gpu_prefill_min_batch_size=r["thr"])
        assert cfg.is_gpu_resident_layer(r["name"]) == r["resident"], r
        assert cfg.is_gpu_prefill_layer(r["name"]) == r["prefill"], r
        assert cfg.is_engine_layer(r["name"]) == r["cpu"], r


def test_parser_and_threshold():
    assert parse_layer_list("0-5,7") == {0, 1, 2, 3, 4, 5, 7}
    assert parse_layer_list(" 3 , 9-8, x, 12-12,") == {3, 12}
    assert parse_layer_list(None) == set() and parse_layer_list("a-b,7-") == set()
    cfg = TierConfig(True, "0-1", 256)
    assert cfg.should_use_gpu_prefill("model.layers.3.mlp.experts", 256)
    assert not cfg.should_use_gpu_prefill("model.layers.3.mlp.experts", 255)
    assert not cfg.should_use_gpu_prefill("model.layers.3.mlp.experts", 4096, graph_capturing=True)
    assert not cfg.should_use_gpu_prefill("model.layers.1.mlp.experts", 4096)        # resident layer
    with pytest.raises(ValueError):
        cfg.is_gpu_resident_layer("model.layers.3.experts.7")
    env = {"LVLLM_MOE_NUMA_ENABLED": "1", "LVLLM_GPU_RESIDENT_MOE_LAYERS": "2", "LVLLM_GPU_PREFILL_MIN_BATCH_SIZE": "64"}
    assert TierConfig.from_env(env) == TierConfig(True, "2", 64)


def test_hbm_plan_for_the_survey_models():
    # Mixtral-8x7B bf16: 32 layers x 8 experts x 3 x 4096 x 14336 x 2 B = 90.2 GB: one GPU
    p = plan_hbm(num_layers=32, num_experts=8, hidden=4096, intermediate=14336, fmt="bf16")
    assert p.fits and p.min_ep_size == 1 and abs(p.per_gpu_expert_bytes - 90.2e9) < 0.2e9
    # DeepSeek-V3 fp8: 58 MoE layers x 256 experts x 3 x 7168 x 2048 = 654 GB: needs EP >= 4 of 8 GPUs
    p = plan_hbm(num_layers=58, num_experts=256, hidden=7168, intermediate=2048, fmt="fp8", ep_size=8,
                 dense_bytes_per_gpu=20 * 10**9, kv_cache_bytes_per_gpu=60 * 10**9)
    assert p.fits and p.min_ep_size == 4
    p1 = plan_hbm(num_layers=58, num_experts=256, hidden=7168, intermediate=2048, fmt="fp8", ep_size=1)
    assert not p1.fits and p1.headroom_bytes < 0
    # 4-bit formats: 4.25 / 4.5 bits per weight with their scales
    b = expert_layer_bytes(8, 4096, 14336, "mxfp4")
    assert abs(b / (8 * 3 * 4096 * 14336) * 8 - 4.25) < 0.01
    b = expert_layer_bytes(8, 4096, 14336, "nvfp4")
    assert abs(b / (8 * 3 * 4096 * 14336) * 8 - 4.5) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bf16", "mxfp4", "int4"])
def test_planner_footprint_equals_engine_weight_bytes(fmt):
    from bench import build_engine
    from lvllm_amd import ops
    E, H, I = 4, 1024, 1408            # I = 1408: rows padded to 64, K padded to the unit
    dev = torch.device("cuda", 0)
    eng = build_engine(ops, dict(fmt=fmt, K=2, g=128, H=H, I=I), E, 0, dev)[0]
    assert eng.engine.weight_bytes() == expert_layer_bytes(E, H, I, fmt, 128)
