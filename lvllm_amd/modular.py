"""The reference's in-tree expert-operator surface over the MI355X engine (secondary boundary).

GPU-resident MoE layers of LvLLM do not go through `lk_moe`; they run a `FusedMoEExpertsModular`
(vllm/model_executor/layers/fused_moe/modular_kernel.py:762-975) between a prepare/finalize pair.  `LkmExperts`
has that class's method surface -- same names, argument order and meaning -- so that it can stand where
`AiterExperts` (experts/rocm_aiter_moe.py:422-572, the reference's own ROCm implementation and the model for the
choices below) stands today:

  * activation format Standard; inputs arrive unquantised (`expects_unquantized_inputs`), the engine quantises
    activations itself for fp8-W8A8;
  * workspaces are managed inside the engine: `workspace_shapes` -> ((0,), (0,), (M, K))  (rocm_aiter_moe.py:501-516);
  * routing weights are applied and the top-k slots reduced inside `apply`, so
    `finalize_weight_and_reduce_impl()` is the no-op reducer (topk_weight_and_reduce.py:44-77);
  * `expert_map` (global -> local id, -1 = not on this rank) is applied on the device
    (routed_experts.py:1332-1342) -- the expert-parallel form of the same operator.

vLLM is not importable next to this repository's tests (zmq, msgspec, ... are absent), so the class is
duck-typed; `bind_vllm_base()` returns a real `mk.FusedMoEExpertsModular` subclass when vLLM is present.
No CPU path: CPU tensors raise, a missing liblkm.so raises at import of `lvllm_amd.ops`.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Any

import torch

from . import _clib, ops

# vllm/model_executor/layers/fused_moe/activation.py:16-35 (MoEActivation values) -> MOEConfigV2.activation_type
# (routed_experts.py:160-164: 0 silu-gated, 1 swigluoai, 2 relu2 without gate)
_ACTIVATIONS = {"silu": (0, True), "swigluoai": (1, True), "relu2_no_mul": (2, False)}


def _act_name(activation: Any) -> str:
    """MoEActivation member or its string value"""
    return str(getattr(activation, "value", activation))


@dataclass
class LkmQuant:
    """What the engine needs of a FusedMoEQuantConfig (fused_moe/config.py): the weight format, its scales and
    group / block shape.  `from_vllm` reads the same facts off the reference's object."""
    fmt: str = "bf16"                    # bf16 | fp16 | int4 | fp8 | mxfp4 | nvfp4 | wna16 (expanded at hand-off)
    w1_scale: torch.Tensor | None = None
    w2_scale: torch.Tensor | None = None
    group_n: int = 0
    group_k: int = 0
    w1_global_scale: torch.Tensor | None = None
    w2_global_scale: torch.Tensor | None = None
    fp8_mode: int = _clib.FP8_W8A16      # FP8_W8A8: activations quantised 1x128 on the fly (in-tree block-fp8)
    # fmt "wna16": weight-only integers the packed uint4b8 format cannot hold -- zero points and / or 8 bits
    # (fused_moe.py:207-276).  Expanded once to T((q - zp) * s) by lkm_wna16_expand; the 16-bit engine runs on them.
    weight_bits: int = 4
    w1_zp: torch.Tensor | None = None
    w2_zp: torch.Tensor | None = None

    @staticmethod
    def from_vllm(qc: Any, act_dtype: torch.dtype) -> "LkmQuant":
        if qc is None or getattr(qc, "w1_scale", None) is None:
            return LkmQuant(fmt="bf16" if act_dtype == torch.bfloat16 else "fp16")
        block = getattr(qc, "block_shape", None)
        if getattr(qc, "use_fp8_w8a8", False) or getattr(qc, "use_fp8_w8a16", False):
            if not block or list(block) != [128, 128]:
                raise ValueError(f"fp8 experts: only 128x128 block scales are supported, got {block}")
            return LkmQuant("fp8", qc.w1_scale, qc.w2_scale, 128, 128,
                            fp8_mode=_clib.FP8_W8A8 if getattr(qc, "use_fp8_w8a8", False) else _clib.FP8_W8A16)
        int4, int8 = getattr(qc, "use_int4_w4a16", False), getattr(qc, "use_int8_w8a16", False)
        if int4 or int8:
            for name in ("w1_bias", "w2_bias"):
                if getattr(qc, name, None) is not None:
                    raise ValueError(f"weight-only integer experts with {name} are not supported by the MI355X engine")
            z1, z2 = getattr(qc, "w1_zp", None), getattr(qc, "w2_zp", None)
            if (z1 is None) != (z2 is None):
                raise ValueError("zero points must be given for both w1 and w2 or for neither")
            if not block or int(block[1]) <= 0:
                raise ValueError(f"weight-only integer experts need block_shape [0, group], got {block}")
            group = int(block[1])
            if int4 and z1 is None:      # symmetric 4-bit (uint4b8): the engine's native packed format
                return LkmQuant("int4", qc.w1_scale, qc.w2_scale, 1, group)
            # zero points (uint4 / uint8) or symmetric 8-bit (uint8b128): NEVER the uint4b8 decoder -- (q - 8) * s
            # is not (q - zp) * s
            return LkmQuant("wna16", qc.w1_scale, qc.w2_scale, 1, group, weight_bits=4 if int4 else 8,
                            w1_zp=z1, w2_zp=z2)
        if getattr(qc, "use_mxfp4_w4a16", False):
            return LkmQuant("mxfp4", qc.w1_scale, qc.w2_scale, 1, 32)
        raise ValueError("quantisation scheme of this FusedMoEQuantConfig is not supported by the MI355X engine")


def _unpack_zp4(zp: torch.Tensor) -> torch.Tensor:
    """packed 4-bit zero points [E, R / 2, K / g] (low nibble = even row: tests/kernels/moe/test_moe.py:634-641) -> uint8
    [E, R, K / g], what lkm_create takes in the zero-point mode"""
    z = zp.view(torch.uint8)
    out = torch.empty((z.shape[0], z.shape[1] * 2, z.shape[2]), dtype=torch.uint8, device=z.device)
    out[:, 0::2] = z & 0xF
    out[:, 1::2] = z >> 4
    return out


class _NoOpReduce:
    """`TopKWeightAndReduceNoOP` (topk_weight_and_reduce.py:44-77): apply() already weighted and reduced."""

    def __eq__(self, other):
        return type(other).__name__ in ("_NoOpReduce", "TopKWeightAndReduceNoOP")

    def apply(self, output, fused_expert_output, topk_weights, topk_ids, apply_router_weight_on_input):
        if output is None or output is fused_expert_output:
            return fused_expert_output
        assert output.size() == fused_expert_output.size(), (output.size(), fused_expert_output.size())
        output.copy_(fused_expert_output, non_blocking=True)
        return output


class LkmExperts:
    """`FusedMoEExpertsModular`-shaped operator; one instance per MoE layer (it owns that layer's pre-shuffled
    weights in HBM, created from w1 / w2 on the first `apply`, like the engine behind `lk_moe.MOE_*`)."""

    def __init__(self, moe_config: Any = None, quant_config: Any = None, *, quant: LkmQuant | None = None,
                 max_num_tokens: int = 8192, max_num_seqs: int = 256):
        self.moe_config, self.quant_config = moe_config, quant_config
        self._quant = quant
        self._max_num_tokens, self._max_num_seqs = max_num_tokens, max_num_seqs
        self._engine: ops.RoutedExpertsEngine | None = None
        self._engine_key: tuple | None = None

    # ------------------------------------------------------------------ class-level facts (modular_kernel.py:508-664)
    @staticmethod
    def is_monolithic() -> bool:
        return False

    @staticmethod
    def activation_format():
        try:
            from vllm.model_executor.layers.fused_moe import modular_kernel as mk
            return mk.FusedMoEActivationFormat.Standard
        except Exception:
            return "standard"

    @property
    def expects_unquantized_inputs(self) -> bool:
        return True

    @staticmethod
    def _supports_current_device() -> bool:
        try:
            n, arch = _clib.device_info()
        except Exception:
            return False
        return n >= 1 and arch.startswith("gfx950")

    @staticmethod
    def _supports_no_act_and_mul() -> bool:
        return True                       # relu2_no_mul (non-gated experts)

    @staticmethod
    def _supports_activation(activation: Any) -> bool:
        return _act_name(activation) in _ACTIVATIONS

    @staticmethod
    def _supports_quant_scheme(weight_key: Any, activation_key: Any) -> bool:
        """keys are the reference's QuantKey objects (quant_utils.py) or None; matched by name so that this module
        imports without vLLM"""
        w = "none" if weight_key is None else str(weight_key).lower()
        a = "none" if activation_key is None else str(activation_key).lower()
        if w == "none":
            return a == "none"
        if "fp8" in w and "128" in w:                       # kFp8Static128BlockSym x (None | kFp8Dynamic128Sym)
            return a == "none" or ("fp8" in a and "128" in a)
        if "mxfp4" in w or "nvfp4" in w:
            return a == "none"                              # W4A16
        if "int4" in w or "uint4" in w or "int8" in w or "uint8" in w:
            # uint4b8 natively; zero-point uint4 and 8-bit weight-only (uint8b128 / uint8 + zp) expanded to 16 bits at
            # hand-off (LkmQuant fmt "wna16").  Integer ACTIVATIONS (w8a8 int8) are not supported.
            return a == "none"
        return False

    @staticmethod
    def _supports_parallel_config(moe_parallel_config: Any) -> bool:
        return True                       # TP splits the intermediate size, EP arrives as expert_map

    @staticmethod
    def _supports_batch_invariance() -> bool:
        return True                       # fixed-order reductions throughout

    # ------------------------------------------------------------------ shapes (modular_kernel.py:772-870)
    def moe_problem_size(self, a1: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, topk_ids: torch.Tensor):
        assert w1.dim() == 3 and w2.dim() == 3
        E, N, _ = w1.shape
        K = a1.size(-1)
        assert a1.dim() == 2 and topk_ids.size(0) == a1.size(0), f"{topk_ids.size(0)} != {a1.size(0)}"
        assert topk_ids.dim() == 2
        return E, a1.size(0), N, K, topk_ids.size(1)

    def workspace_dtype(self, act_dtype: torch.dtype) -> torch.dtype:
        return act_dtype

    def workspace_shapes(self, M: int, N: int, K: int, topk: int, global_num_experts: int, local_num_experts: int,
                         expert_tokens_meta: Any, activation: Any):
        # scratch lives in the engine's per-device arena
        return (0,), (0,), (M, K)

    @staticmethod
    def adjust_N_for_activation(N: int, activation: Any) -> int:
        name = _act_name(activation)
        gated = _ACTIVATIONS[name][1] if name in _ACTIVATIONS else not name.endswith("_no_mul")
        return N // 2 if gated else N

    def finalize_weight_and_reduce_impl(self):
        try:
            from vllm.model_executor.layers.fused_moe.topk_weight_and_reduce import TopKWeightAndReduceNoOP
            return TopKWeightAndReduceNoOP()
        except Exception:
            return _NoOpReduce()

    # ------------------------------------------------------------------ the operator (modular_kernel.py:922-975)
    def _engine_for(self, w1: torch.Tensor, w2: torch.Tensor, topk: int, act_dtype: torch.dtype, activation: Any):
        name = _act_name(activation)
        if name not in _ACTIVATIONS:
            raise ValueError(f"activation {name!r} is not supported by the MI355X expert engine "
                             f"(supported: {sorted(_ACTIVATIONS)})")
        key = (w1.data_ptr(), w2.data_ptr(), tuple(w1.shape), tuple(w2.shape), topk, act_dtype, name)
        if self._engine is None or self._engine_key != key:
            q = self._quant or LkmQuant.from_vllm(self.quant_config, act_dtype)
            act_type, gated = _ACTIVATIONS[name]
            fmt = q.fmt if q.fmt not in ("bf16", "fp16") else ("bf16" if act_dtype == torch.bfloat16 else "fp16")
            w1u = w1.view(torch.uint8) if w1.dtype == torch.float8_e4m3fn else w1
            w2u = w2.view(torch.uint8) if w2.dtype == torch.float8_e4m3fn else w2
            s1, s2, group_n, group_k = q.w1_scale, q.w2_scale, q.group_n, q.group_k
            zp1 = zp2 = None
            if fmt == "wna16":
                if q.w1_scale.dtype != act_dtype or q.w2_scale.dtype != act_dtype:
                    raise ValueError(f"wna16 scales are {q.w1_scale.dtype}, activations {act_dtype}: the in-tree "
                                     "operator dequantises to the activation dtype (fused_moe.py:270-276)")
                if q.weight_bits == 4 and q.w1_zp is not None and self._native_zp_ok(w1u, w2u, q):
                    # asymmetric uint4: the engine streams the packed 4-bit image and decodes T((q - zp) * s) in registers
                    # (LkmConfig.int4_mode = LKM_INT4_ZP); zero points unpacked to one byte per (row, group)
                    fmt, zp1, zp2 = "int4", _unpack_zp4(q.w1_zp), _unpack_zp4(q.w2_zp)
                    w1u, w2u = w1u.view(torch.uint8), w2u.view(torch.uint8)
            if fmt == "wna16":
                w1u = ops.wna16_expand(w1u.view(torch.uint8), q.w1_scale, q.w1_zp, q.weight_bits, q.group_k)
                w2u = ops.wna16_expand(w2u.view(torch.uint8), q.w2_scale, q.w2_zp, q.weight_bits, q.group_k)
                fmt, s1, s2, group_n, group_k = ("bf16" if act_dtype == torch.bfloat16 else "fp16"), None, None, 0, 0
            self._engine = ops.RoutedExpertsEngine(
                w1u, w2u, top_k=topk, act_dtype=act_dtype, fmt=fmt, w13_scale=s1, w2_scale=s2,
                group_n=group_n, group_k=group_k, has_gate_proj=gated, activation_type=act_type,
                max_num_seqs=self._max_num_seqs, max_batch_size=self._max_num_tokens, fp8_mode=q.fp8_mode,
                w13_global_scale=q.w1_global_scale, w2_global_scale=q.w2_global_scale, w13_zp=zp1, w2_zp=zp2)
            self._engine_key = key
        return self._engine

    @staticmethod
    def _native_zp_ok(w1u: torch.Tensor, w2u: torch.Tensor, q: LkmQuant) -> bool:
        """the shapes lkm_create takes in the zero-point mode (include/lkm.h: LkmConfig.groupK for int4); anything else is
        expanded to 16 bits at hand-off"""
        g = int(q.group_k)
        k1, k2 = w1u.shape[-1] * 2, w2u.shape[-1] * 2
        if not (g in (32, 64, 128) or (g > 128 and g % 128 == 0)):
            return False
        return k1 % g == 0 and k2 % g == 0 and k1 % 128 == 0 and k2 % 128 == 0 and os.environ.get("LKM_WNA16_EXPAND", "0") != "1"

    def invalidate(self) -> None:
        """Drop the cached engine: the next apply() copies and pre-shuffles w1 / w2 again.  lkm_create COPIES the
        weights, so an in-place update of the tensors (weight reload, an EPLB rearrangement of torch parameters
        through TensorExpertStore) is invisible to the engine until this is called; rearrangements through
        EngineExpertStore move the engine's own images and need no invalidation.  After the first apply() the
        caller may free w1 / w2 (as the reference does, routed_experts.py:1420-1432) -- keep passing tensors with
        the same data_ptr / shape as the cache key, or hold on to the engine."""
        self._engine = None
        self._engine_key = None

    def apply(self, output: torch.Tensor, hidden_states: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor,
              topk_weights: torch.Tensor, topk_ids: torch.Tensor, activation: Any, global_num_experts: int,
              expert_map: torch.Tensor | None, a1q_scale: torch.Tensor | None, a2_scale: torch.Tensor | None,
              workspace13: torch.Tensor | None, workspace2: torch.Tensor | None, expert_tokens_meta: Any,
              apply_router_weight_on_input: bool) -> None:
        """output [M, K] <- sum_k w[m,k] * W2[e] act(W13[e] x_m), weighted and reduced (fp32 output: the
        `cpu_decode` arithmetic; activation-dtype output: `gpu_prefill`)."""
        if not hidden_states.is_cuda:
            raise RuntimeError("LkmExperts.apply needs tensors on an MI355X (no CPU path)")
        if a1q_scale is not None:
            raise ValueError("LkmExperts expects unquantised activations (expects_unquantized_inputs); "
                             "got a1q_scale")
        if hidden_states.dim() != 2 or output.shape != hidden_states.shape:
            raise ValueError(f"standard activation format: hidden [M, K] and output [M, K], got "
                             f"{tuple(hidden_states.shape)} / {tuple(output.shape)}")
        M, topk = topk_ids.shape
        if apply_router_weight_on_input:
            # the caller already multiplied the inputs by the routing weight (only defined for top-1,
            # fused_moe.py:1700-1704)
            if topk != 1:
                raise ValueError("apply_router_weight_on_input is only supported for topk=1")
            topk_weights = torch.ones_like(topk_weights, dtype=torch.float32)
        eng = self._engine_for(w1, w2, topk, hidden_states.dtype, activation)
        ids = topk_ids if topk_ids.dtype == torch.int32 else topk_ids.to(torch.int32)
        if expert_map is not None:
            ids = ops.global_to_local_expert_ids(ids.contiguous(), expert_map.to(device=ids.device, dtype=torch.int32))
        tw = topk_weights.to(torch.float32)
        if M == 0:
            return

        def rowwise(t_, align):   # rows of the token-granular EP exchange arrive as row-strided views: used in place
            ok = t_.dim() == 2 and (t_.size(1) == 1 or t_.stride(1) == 1) and t_.stride(0) % align == 0
            return t_ if ok else t_.contiguous()
        x, ids, tw = rowwise(hidden_states, 8), rowwise(ids, 1), rowwise(tw, 1)
        den = getattr(expert_tokens_meta, "valid_den", None)
        if output.is_contiguous() and output.dtype in (torch.float32, hidden_states.dtype):
            eng.forward_rows(x, tw, ids, out=output, valid_den=den)
        else:
            output.copy_(eng.forward_rows(x, tw, ids, out_dtype=torch.float32 if output.dtype == torch.float32
                                          else hidden_states.dtype, valid_den=den))


class LkmTokensMeta:
    """what LkmPrepareAndFinalize.prepare hands to LkmExperts.apply in the `expert_tokens_meta` position"""
    __slots__ = ("valid_den", "expert_num_tokens", "expert_num_tokens_cpu")

    def __init__(self, valid_den: int):
        self.valid_den, self.expert_num_tokens, self.expert_num_tokens_cpu = valid_den, None, None


class LkmPrepareAndFinalize:
    """`FusedMoEPrepareAndFinalizeModular`-shaped pair (modular_kernel.py:180-418) for expert parallelism over the
    8 GPUs of one node: `prepare` is the token-granular fixed-capacity RCCL all-to-all of lvllm_amd/ep.py (tokens ->
    owning ranks, one collective, no host sync), `finalize` the reverse all-to-all and the fixed-order fp32 sum.
    Together with `LkmExperts`:

        a1q, a1q_scale, meta, ids_d, w_d = pf.prepare(a1, topk_weights, topk_ids, E, expert_map, False, None, True)
        experts.apply(fused, a1q, w1, w2, w_d, ids_d, activation, E, expert_map, None, None, None, None, None, False)
        pf.finalize(output, fused, topk_weights, topk_ids, False, experts.finalize_weight_and_reduce_impl())

    which is the call sequence of the reference's modular kernel (modular_kernel.py:1219-1420).  A dispatched row is
    one TOKEN with its top-k ids and weights (ids of experts on other ranks and of empty record slots are -1); the
    ids are GLOBAL because the experts apply `expert_map` themselves, as with the reference's all-to-all backends
    (deepep_ht_prepare_finalize.py:196-215).  Every rank must pass the same token count (or construct with a common
    `capacity_tokens`): see lvllm_amd/ep.py.  Above `fixed_max_tokens` tokens the equal-split exchange would pad
    prefill-sized batches; use `ExpertParallelExperts.forward` (ragged path) there."""

    def __init__(self, num_experts: int, hidden_size: int, group=None, kernels=None, transport=None,
                 capacity_tokens: int | None = None, fixed_max_tokens: int = 1024, pool_tag: str = ""):
        from .ep import ExpertParallelExperts
        # pool_tag: the exchange-buffer pool of the (device, group) this instance packs into; two micro-batches in
        # flight (dual-batch overlap) = two instances with different tags (lvllm_amd/ep.py: _pool)
        self._ep = ExpertParallelExperts(lambda *a: None, num_experts, hidden_size, group=group, mode="a2a",
                                         kernels=kernels, transport=transport, capacity_tokens=capacity_tokens,
                                         fixed_max_tokens=fixed_max_tokens, global_ids=True,
                                         return_dtype=torch.float32,   # the experts' rows return as they are
                                         pool_tag=pool_tag)
        self._shape: tuple[int, int] | None = None
        self._handle = None

    # ---- facts (modular_kernel.py:201-246)
    @property
    def activation_format(self):
        return LkmExperts.activation_format()

    def topk_indices_dtype(self):
        return torch.int32

    def max_num_tokens_per_rank(self):
        return self._ep.capacity_tokens

    def num_dispatchers(self) -> int:
        return self._ep.ep

    def output_is_reduced(self) -> bool:
        return True                       # finalize returns the complete sum for this rank's tokens

    def supports_async(self) -> bool:
        return False

    def post_init_setup(self, fused_experts) -> None:
        pass

    def on_commit(self) -> None:
        pass

    # ---- modular_kernel.py:264-299
    def prepare(self, a1: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor, num_experts: int,
                expert_map: torch.Tensor | None, apply_router_weight_on_input: bool, quant_config: Any,
                defer_input_quant: bool = False):
        if num_experts != self._ep.E:
            raise ValueError(f"prepare: {num_experts} experts, constructed for {self._ep.E}")
        if not defer_input_quant and quant_config is not None and getattr(quant_config, "quant_dtype", None) is not None:
            raise ValueError("LkmPrepareAndFinalize dispatches unquantised rows (the experts quantise); "
                             "call with defer_input_quant=True")
        M, K = topk_ids.shape
        cap = self._ep.capacity_for(M)
        if cap > self._ep.fixed_max_tokens:
            raise ValueError(f"prepare: {cap} tokens per rank exceed the equal-split exchange "
                             f"(fixed_max_tokens={self._ep.fixed_max_tokens}); use ExpertParallelExperts.forward")
        tw = topk_weights.to(torch.float32)
        if apply_router_weight_on_input:
            if K != 1:
                raise ValueError("apply_router_weight_on_input is only supported for topk=1")
            a1 = (a1.to(torch.float32) * tw).to(a1.dtype)
            tw = torch.ones_like(tw)
        if self._handle is not None:
            # one exchange in flight per instance: a second prepare would pack into the buffers the first finalize still
            # has to read (two micro-batches need two instances with different `pool_tag`s; the pool itself refuses a
            # second dispatch too, lvllm_amd/ep.py) -- fail loudly
            raise RuntimeError("LkmPrepareAndFinalize.prepare called again before the matching finalize "
                               "(supports_async() is False: one exchange in flight per instance)")
        rows, gids, ws, self._handle = self._ep.dispatch_fixed(a1.contiguous(), tw.contiguous(),
                                                               topk_ids.to(torch.int32).contiguous(), return_handle=True)
        self._shape = (M, K)
        # third value = expert_tokens_meta (modular_kernel.py:96-118 carries per-expert counts there; this exchange has
        # none on the host): it tells LkmExperts.apply how sparse the records are, so the launch plan is made for the
        # rows that exist (~1/ep of the ep x capacity slots carry a local id)
        return rows, None, LkmTokensMeta(valid_den=self._ep.ep) if self._ep.ep > 1 else None, gids, ws

    def reset(self) -> None:
        """gives up an exchange whose experts raised between prepare and finalize: the handle is dropped and the shared
        exchange pool is free again for every layer's instance (without it the next prepare -- of ANY layer on
        this device / group / pool tag -- fails with "already holds a dispatch")"""
        if self._handle is not None:
            self._ep.abandon_dispatch(self._handle)
            self._handle = None
        self._shape = None

    # ---- modular_kernel.py:354-376
    def finalize(self, output: torch.Tensor, fused_expert_output: torch.Tensor, topk_weights: torch.Tensor,
                 topk_ids: torch.Tensor, apply_router_weight_on_input: bool, weight_and_reduce_impl: Any) -> None:
        if weight_and_reduce_impl is not None and type(weight_and_reduce_impl).__name__ not in (
                "_NoOpReduce", "TopKWeightAndReduceNoOP", "TopKWeightAndReduceDelegate"):
            raise ValueError("LkmPrepareAndFinalize.finalize expects rows that are already weighted "
                             "(TopKWeightAndReduceNoOP), as LkmExperts produces them")
        M, K = topk_ids.shape
        if self._shape != (M, K):
            raise RuntimeError(f"finalize for [{M}, {K}] slots without the matching prepare ({self._shape})")
        y = fused_expert_output if fused_expert_output.dtype == torch.float32 else fused_expert_output.to(torch.float32)
        handle, self._handle = self._handle, None
        out = self._ep.combine_fixed(y.contiguous(), M, output.dtype if output.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32,
                                     handle=handle)
        if out.data_ptr() != output.data_ptr():
            output.copy_(out)


def _mk_module(mk=None):
    if mk is not None:
        return mk
    from vllm.model_executor.layers.fused_moe import modular_kernel
    return modular_kernel


def bind_vllm_base(mk=None):
    """`class LkmExpertsModular(LkmExperts, mk.FusedMoEExpertsModular)` for registration inside LvLLM
    (e.g. through the kernel-selection table of the fused-MoE layer, or @PluggableLayer.register_oot around RoutedExperts, custom_op.py:47-101).
    `mk`: the reference's modular_kernel module (default: imported from vLLM; raises ImportError where vLLM is not
    importable -- the GPU tests pass a copy of the reference module cut out of the reference tree)."""
    mk = _mk_module(mk)
    noop = getattr(mk, "TopKWeightAndReduceNoOP", None)

    class LkmExpertsModular(LkmExperts, mk.FusedMoEExpertsModular):  # type: ignore[misc]
        def __init__(self, moe_config, quant_config, max_num_tokens=None, num_dispatchers=None, **kw):
            mk.FusedMoEExpertsModular.__init__(self, moe_config, quant_config, max_num_tokens, num_dispatchers)
            LkmExperts.__init__(self, moe_config, quant_config, max_num_tokens=max_num_tokens or 8192, **kw)

        @staticmethod
        def activation_format():
            return mk.FusedMoEActivationFormat.Standard

        def finalize_weight_and_reduce_impl(self):
            if noop is not None:
                return noop()
            return LkmExperts.finalize_weight_and_reduce_impl(self)

    return LkmExpertsModular


def bind_vllm_prepare_finalize(mk=None):
    """`class LkmPrepareAndFinalizeModular(LkmPrepareAndFinalize, mk.FusedMoEPrepareAndFinalizeModular)` -- the same for
    the prepare / finalize half (modular_kernel.py:257-418).  Raises ImportError where vLLM is not importable."""
    mk = _mk_module(mk)

    class LkmPrepareAndFinalizeModular(LkmPrepareAndFinalize, mk.FusedMoEPrepareAndFinalizeModular):  # type: ignore[misc]
        @property
        def activation_format(self):
            return mk.FusedMoEActivationFormat.Standard

    return LkmPrepareAndFinalizeModular
