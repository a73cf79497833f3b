"""The in-tree operator surface (lvllm_amd/modular.py, SURVEY 8b secondary boundary): names, argument order and the
host-side behaviour of `LkmExperts` against the reference's `FusedMoEExpertsModular`
(vllm/model_executor/layers/fused_moe/modular_kernel.py:762-975).  The device arithmetic is covered by
tests/test_gpu_moe.py::test_modular_experts_*."""
import ast
import inspect
from pathlib import Path

import pytest
import torch

from lvllm_amd.modular import LkmExperts, LkmQuant, _NoOpReduce

REF = Path("/root/reference/vllm/model_executor/layers/fused_moe/modular_kernel.py")

# modular_kernel.py:922-939
APPLY_PARAMS = ["self", "output", "hidden_states", "w1", "w2", "topk_weights", "topk_ids", "activation",
                "global_num_experts", "expert_map", "a1q_scale", "a2_scale", "workspace13", "workspace2",
                "expert_tokens_meta", "apply_router_weight_on_input"]
# modular_kernel.py:822-832
WORKSPACE_PARAMS = ["self", "M", "N", "K", "topk", "global_num_experts", "local_num_experts", "expert_tokens_meta",
                    "activation"]


def _params(fn):
    return list(inspect.signature(fn).parameters)


def test_method_surface_matches_the_reference_interface():
    assert _params(LkmExperts.apply) == APPLY_PARAMS
    assert _params(LkmExperts.workspace_shapes) == WORKSPACE_PARAMS
    assert _params(LkmExperts.moe_problem_size) == ["self", "a1", "w1", "w2", "topk_ids"]
    for name in ("is_monolithic", "activation_format", "finalize_weight_and_reduce_impl", "workspace_dtype",
                 "adjust_N_for_activation", "_supports_current_device", "_supports_no_act_and_mul",
                 "_supports_activation", "_supports_quant_scheme", "_supports_parallel_config"):
        assert callable(getattr(LkmExperts, name)), name
    assert LkmExperts.is_monolithic() is False


@pytest.mark.skipif(not REF.exists(), reason="reference tree not present")
def test_method_surface_against_the_reference_source():
    """the same comparison read off the reference's source (abstract methods and their parameter names)"""
    tree = ast.parse(REF.read_text())
    cls = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.ClassDef)}
    mod, base = cls["FusedMoEExpertsModular"], cls["FusedMoEExperts"]
    fns = {f.name: f for c in (base, mod) for f in c.body if isinstance(f, ast.FunctionDef)}
    for name in ("apply", "workspace_shapes", "moe_problem_size", "workspace_dtype", "finalize_weight_and_reduce_impl"):
        want = [a.arg for a in fns[name].args.args]
        assert _params(getattr(LkmExperts, name)) == want, name
    abstract = [f.name for c in (base, mod) for f in c.body if isinstance(f, ast.FunctionDef)
                and any(getattr(d, "id", getattr(d, "attr", "")) == "abstractmethod" for d in f.decorator_list)]
    missing = [n for n in abstract if not hasattr(LkmExperts, n)]
    assert not missing, missing


def test_shapes_and_reducer():
    ex = LkmExperts()
    assert ex.workspace_shapes(33, 28672, 4096, 2, 8, 8, None, "silu") == ((0,), (0,), (33, 4096))
    assert ex.workspace_dtype(torch.bfloat16) == torch.bfloat16
    assert LkmExperts.adjust_N_for_activation(28672, "silu") == 14336
    assert LkmExperts.adjust_N_for_activation(1024, "relu2_no_mul") == 1024
    a1, w1, w2 = torch.empty(5, 64), torch.empty(8, 256, 64), torch.empty(8, 64, 128)
    assert ex.moe_problem_size(a1, w1, w2, torch.zeros(5, 2, dtype=torch.int32)) == (8, 5, 256, 64, 2)
    red = ex.finalize_weight_and_reduce_impl()
    y = torch.randn(5, 64)
    assert red.apply(None, y, None, None, False) is y
    out = torch.empty(5, 64)
    assert red.apply(out, y, None, None, False) is out and torch.equal(out, y)
    assert _NoOpReduce() == _NoOpReduce()


def test_support_predicates():
    assert LkmExperts._supports_activation("silu") and LkmExperts._supports_activation("swigluoai")
    assert LkmExperts._supports_activation("relu2_no_mul") and not LkmExperts._supports_activation("gelu")
    assert LkmExperts._supports_quant_scheme(None, None)
    assert LkmExperts._supports_quant_scheme("kFp8Static128BlockSym", "kFp8Dynamic128Sym")
    assert LkmExperts._supports_quant_scheme("kMxfp4Static", None)
    assert not LkmExperts._supports_quant_scheme("kFp8StaticTensorSym", "kFp8StaticTensorSym")
    assert not LkmExperts._supports_quant_scheme(None, "kFp8Dynamic128Sym")
    assert LkmExperts._supports_no_act_and_mul() and LkmExperts._supports_parallel_config(None)
    assert LkmExperts._supports_current_device() in (True, False)      # False without an MI355X, never raises


def test_quant_from_reference_style_config():
    class QC:       # the attributes of FusedMoEQuantConfig that matter (fused_moe/config.py)
        w1_scale = torch.ones(2, 2, 1)
        w2_scale = torch.ones(2, 1, 2)
        block_shape = [128, 128]
        use_fp8_w8a8 = True
    q = LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert (q.fmt, q.group_n, q.group_k) == ("fp8", 128, 128)
    QC.block_shape = [64, 64]
    with pytest.raises(ValueError):
        LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert LkmQuant.from_vllm(None, torch.float16).fmt == "fp16"


def test_weight_only_integer_configs_never_fall_into_the_symmetric_4bit_decoder():
    """a config with zero points (or 8-bit weights) must not be mapped to uint4b8 -- it takes the
    expanded format (exact (q - zp) * s), or raises; test grid of tests/kernels/moe/test_moe.py:565-693."""
    class QC:
        w1_scale = torch.ones(2, 4, 1)
        w2_scale = torch.ones(2, 2, 2)
        w1_zp = w2_zp = w1_bias = w2_bias = None
        block_shape = [0, 128]
        use_int4_w4a16 = True
        use_int8_w8a16 = False
    q = LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert (q.fmt, q.group_k, q.weight_bits) == ("int4", 128, 4)          # uint4b8: native
    QC.w1_zp, QC.w2_zp = torch.zeros(2, 2, 1, dtype=torch.uint8), torch.zeros(2, 1, 2, dtype=torch.uint8)
    q = LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert q.fmt == "wna16" and q.weight_bits == 4 and q.w1_zp is QC.w1_zp and q.w2_zp is QC.w2_zp
    QC.w2_zp = None
    with pytest.raises(ValueError):                                        # half a zero-point pair
        LkmQuant.from_vllm(QC(), torch.bfloat16)
    QC.w1_zp = None
    QC.use_int4_w4a16, QC.use_int8_w8a16 = False, True
    q = LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert q.fmt == "wna16" and q.weight_bits == 8 and q.w1_zp is None      # uint8b128
    QC.w1_bias = torch.zeros(2, 4)
    with pytest.raises(ValueError):
        LkmQuant.from_vllm(QC(), torch.bfloat16)
    QC.w1_bias, QC.block_shape = None, None
    with pytest.raises(ValueError):
        LkmQuant.from_vllm(QC(), torch.bfloat16)
    assert LkmExperts._supports_quant_scheme("uint4b8", None) and LkmExperts._supports_quant_scheme("uint4", None)
    assert LkmExperts._supports_quant_scheme("uint8b128", None) and not LkmExperts._supports_quant_scheme("int8", "int8")


def test_zero_point_layers_choose_the_native_mode_by_shape_and_unpack_the_reference_layout():
    """4-bit experts with zero points run the engine's zero-point mode (LkmConfig.int4_mode = LKM_INT4_ZP) when lkm_create takes
    their shape -- group 32 / 64 / 128 / a multiple of 128, both K multiples of 128 and of the group -- and are expanded to 16 bits
    otherwise (LKM_WNA16_EXPAND=1 forces that); the packed zero points [E, R / 2, K / g] (low nibble = even row,
    tests/kernels/moe/test_moe.py:634-641) become one byte per (row, group)."""
    import numpy as np
    from lvllm_amd.modular import _unpack_zp4
    rng = np.random.default_rng(0)
    z = rng.integers(0, 16, (3, 8, 5), dtype=np.uint8)
    packed = torch.from_numpy((z[:, 0::2] | (z[:, 1::2] << 4)).astype(np.uint8))
    assert np.array_equal(_unpack_zp4(packed).numpy(), z) and _unpack_zp4(packed).dtype == torch.uint8
    w1, w2 = torch.zeros(2, 64, 128, dtype=torch.uint8), torch.zeros(2, 256, 64, dtype=torch.uint8)      # K = 256 / 128

    def ok(group, a=w1, b=w2):
        return LkmExperts._native_zp_ok(a, b, LkmQuant("wna16", None, None, 1, group, weight_bits=4))
    assert ok(32) and ok(64) and ok(128)
    assert not ok(256)                                     # does not divide GEMM2's K = 128
    assert ok(256, b=torch.zeros(2, 256, 128, dtype=torch.uint8))
    assert not ok(48) and not ok(16)
    assert not ok(64, a=torch.zeros(2, 64, 96, dtype=torch.uint8))       # K = 192: not a multiple of 128
    import os
    os.environ["LKM_WNA16_EXPAND"] = "1"
    try:
        assert not ok(128)
    finally:
        del os.environ["LKM_WNA16_EXPAND"]


def test_errors_are_loud_on_the_host_side():
    ex = LkmExperts()
    x = torch.zeros(4, 64, dtype=torch.bfloat16)
    ids = torch.zeros(4, 2, dtype=torch.int32)
    tw = torch.ones(4, 2)
    w1, w2 = torch.zeros(2, 64, 64, dtype=torch.bfloat16), torch.zeros(2, 64, 32, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ex.apply(torch.empty_like(x), x, w1, w2, tw, ids, "silu", 2, None, None, None, None, None, None, False)


def test_prepare_finalize_surface_matches_the_reference_interface():
    """FusedMoEPrepareAndFinalizeModular (modular_kernel.py:257-418): prepare / finalize parameter names; the
    abstract facts of its base (:201-239).  Behaviour over 2 and 4 ranks: tests/test_ep_gloo.py."""
    from lvllm_amd.modular import LkmPrepareAndFinalize
    assert _params(LkmPrepareAndFinalize.prepare) == ["self", "a1", "topk_weights", "topk_ids", "num_experts",
                                                      "expert_map", "apply_router_weight_on_input", "quant_config",
                                                      "defer_input_quant"]
    assert _params(LkmPrepareAndFinalize.finalize) == ["self", "output", "fused_expert_output", "topk_weights",
                                                       "topk_ids", "apply_router_weight_on_input",
                                                       "weight_and_reduce_impl"]
    if REF.exists():
        tree = ast.parse(REF.read_text())
        cls = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.ClassDef)}
        for cname in ("FusedMoEPrepareAndFinalize", "FusedMoEPrepareAndFinalizeModular"):
            for f in cls[cname].body:
                if isinstance(f, ast.FunctionDef) and any(
                        getattr(d, "id", getattr(d, "attr", "")) == "abstractmethod" for d in f.decorator_list):
                    assert hasattr(LkmPrepareAndFinalize, f.name), f.name
                    if f.name in ("prepare", "finalize"):
                        assert _params(getattr(LkmPrepareAndFinalize, f.name)) == [a.arg for a in f.args.args]
