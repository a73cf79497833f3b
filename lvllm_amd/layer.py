"""One MoE layer's routed-expert step as a single object: the MI355X-native form of what the reference spreads
over `MoERunner._apply_quant_method` (vllm/model_executor/layers/fused_moe/runner/moe_runner.py:577-664), the
router (`router/base_router.py:204-260`: routing, EPLB mapping), `RoutedExperts.global_to_local_expert_ids`
(`routed_experts.py:1332-1342`), the three `_cpu_decode / _cpu_prefill / _gpu_prefill` callers (`:1840-1899`) and
the post-processing (`:1853-1854` NaN scrub + dtype, `moe_runner.py:391-408` routed scaling).  SURVEY 8 rows a4
and a10 plus the order in which a1/a2, f4 (EPLB), a3, f2 (shared experts) and a6/a8 are chained; it is what
`@PluggableLayer.register_oot(name="RoutedExperts")` (custom_op.py:47-101) would install as the whole-layer
replacement.

Dispatch, re-read for a tier where every expert is in HBM: the reference picks between a CPU decode path (the only
graph-capturable one), a host-pointer CPU prefill and a streamed GPU prefill.  Here both remaining paths are
on-device and capturable; the choice is only the output convention -- `decode` (fp32 rows into one shared
`[max_num_seqs, H]` buffer, the `cpu_decode` contract) up to `max_num_seqs` tokens, `prefill` (activation dtype)
above.  No host path is ever taken.

Host glue only: every tensor operation below is a HIP kernel of liblkm.so (through `lvllm_amd.ops`) or one of the
three elementwise torch calls the reference itself issues in its post-processing.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .eplb import EplbLayerState
from .shared_experts import SharedExpertSlots

__all__ = ["RoutingConfig", "RoutedExpertsLayer"]


@dataclass
class RoutingConfig:
    """what `create_fused_moe_router` (router/router_factory.py) is configured with"""
    top_k: int
    num_experts: int                                 # global LOGICAL routed experts
    renormalize: bool = True
    scoring_func: str = "softmax"
    use_grouped_topk: bool = False
    num_expert_group: int = 0
    topk_group: int = 0
    e_score_correction_bias: torch.Tensor | None = None
    routed_scaling_factor: float = 1.0
    apply_routed_scaling_in_router: bool = False     # grouped_topk_router.py:156-160 scales the weights itself


class RoutedExpertsLayer:
    def __init__(self, engine, routing: RoutingConfig, *, gate_weight: torch.Tensor | None = None,
                 expert_map: torch.Tensor | None = None, eplb_state: EplbLayerState | None = None,
                 shared_slots: SharedExpertSlots | None = None, shared_gate_weight: torch.Tensor | None = None,
                 check_nan_in_output: bool = False, max_num_seqs: int | None = None, expert_parallel=None,
                 ops=None):
        """engine: lvllm_amd.ops.RoutedExpertsEngine holding this rank's (physical, + shared) experts.
        gate_weight [E, H]: run the gate projection inside the router operator (f2); else forward() takes logits.
        expert_map int32 [P (+ shared + sentinel)]: global -> local ids of this EP rank, None without EP.
        eplb_state: logical -> physical maps + load counters of this layer (lvllm_amd.eplb.EplbState.layer_state).
        shared_slots: the shared experts' slot buffers when they are folded into the engine (shared_experts.py);
        shared_gate_weight [1, H]: Qwen2-MoE style sigmoid gate of the shared expert.
        expert_parallel: an lvllm_amd.ep.ExpertParallelExperts whose local_compute runs `engine`; the routed rows
        then travel by all-to-all (tokens stay on their rank) and forward() returns the COMPLETE routed output of
        this rank's tokens -- no reduction is left to the caller.  Exclusive with expert_map (the replicated-token
        form) and, for now, with folded shared experts.
        ops: the operator namespace (default lvllm_amd.ops, the HIP kernels); tests inject a CPU double."""
        if ops is None:
            from . import ops as _ops
            ops = _ops
        self.ops = ops
        self.engine = engine
        self.routing = routing
        self.gate_weight = gate_weight
        self.expert_map = expert_map
        self.eplb_state = eplb_state
        self.shared_slots = shared_slots
        self.shared_gate_weight = shared_gate_weight
        self.check_nan_in_output = check_nan_in_output
        self.max_num_seqs = int(max_num_seqs if max_num_seqs is not None else engine.cfg.max_num_seqs)
        if routing.use_grouped_topk and not (routing.num_expert_group > 0 and routing.topk_group > 0):
            raise ValueError("grouped top-k needs num_expert_group and topk_group")
        if shared_gate_weight is not None and shared_slots is None:
            raise ValueError("a shared-expert gate needs shared_slots")
        if expert_parallel is not None and (expert_map is not None or shared_slots is not None):
            raise ValueError("expert_parallel dispatches global ids itself: no expert_map; folded shared experts "
                             "are not dispatched (run them on the token's own rank)")
        self.expert_parallel = expert_parallel
        self._decode_out: torch.Tensor | None = None          # the shared fp32 buffer of the cpu_decode contract

    # ---- a1 / a2 (+ gate projection, f2)
    def select_experts(self, hidden_states: torch.Tensor, router_logits: torch.Tensor | None):
        r = self.routing
        scale = r.routed_scaling_factor if r.apply_routed_scaling_in_router else 1.0
        if self.gate_weight is not None:
            return self.ops.router_topk(hidden_states, self.gate_weight, r.top_k, r.renormalize,
                                        scoring_func=r.scoring_func,
                                        num_expert_group=r.num_expert_group if r.use_grouped_topk else 0,
                                        topk_group=r.topk_group if r.use_grouped_topk else 0,
                                        routed_scaling_factor=scale, e_score_correction_bias=r.e_score_correction_bias)
        if router_logits is None:
            raise ValueError("router_logits are required when the layer has no gate_weight")
        if router_logits.size(0) != hidden_states.size(0):
            raise ValueError("Number of tokens mismatch")
        if r.use_grouped_topk:
            return self.ops.grouped_topk(hidden_states, router_logits, r.top_k, r.renormalize, r.num_expert_group,
                                         r.topk_group, r.scoring_func, scale, r.e_score_correction_bias)
        return self.ops.topk_softmax(router_logits, r.top_k, r.renormalize, r.e_score_correction_bias, r.scoring_func,
                                     scale)

    def _decode_buffer(self, hidden_states: torch.Tensor) -> torch.Tensor:
        buf = self._decode_out
        if buf is None or buf.device != hidden_states.device or buf.size(1) != hidden_states.size(1):
            buf = torch.empty((self.max_num_seqs, hidden_states.size(1)), dtype=torch.float32,
                              device=hidden_states.device)
            self._decode_out = buf
        return buf

    # ---- the whole step
    def forward(self, hidden_states: torch.Tensor, router_logits: torch.Tensor | None = None) -> torch.Tensor:
        """[M, H] activations (+ [M, E] logits unless the layer owns the gate) -> [M, H] in the activation dtype:
        the routed (+ folded shared) experts' output of THIS rank, before the TP/EP reduction (with
        `expert_parallel`: the complete output for this rank's tokens)."""
        M = hidden_states.size(0)
        if M == 0:
            return torch.empty_like(hidden_states)
        r = self.routing
        if (self.gate_weight is None and router_logits is not None and self.eplb_state is None
                and self.shared_slots is None and self.expert_map is None and self.expert_parallel is None
                and M <= self.max_num_seqs and router_logits.size(0) == M and hasattr(self.engine, "forward_logits")):
            # plain decode step: routing + experts through lkm_forward_routed (router + scatter metadata in one launch;
            # one to four tokens of a many-expert layer: router inside GEMM1, two launches in all; same bits as
            # select_experts + decode)
            buf = self._decode_buffer(hidden_states)
            out, _, _ = self.engine.forward_logits(
                hidden_states, router_logits, r.top_k, r.renormalize, scoring_func=r.scoring_func,
                num_expert_group=r.num_expert_group if r.use_grouped_topk else 0,
                topk_group=r.topk_group if r.use_grouped_topk else 0,
                routed_scaling_factor=r.routed_scaling_factor if r.apply_routed_scaling_in_router else 1.0,
                e_score_correction_bias=r.e_score_correction_bias, out=buf[:M])
            if self.check_nan_in_output:
                torch.nan_to_num(out, nan=0.0, out=out)
            out = out.to(hidden_states.dtype)
            if r.routed_scaling_factor != 1.0 and not r.apply_routed_scaling_in_router:
                out *= r.routed_scaling_factor
            return out
        topk_weights, topk_ids = self.select_experts(hidden_states, router_logits)
        if self.eplb_state is not None:                       # BaseRouter._apply_eplb_mapping, base_router.py:204-223
            s = self.eplb_state
            topk_ids = self.ops.eplb_map_to_physical_and_record(
                topk_ids, s.expert_load_view, s.logical_to_physical_map, s.logical_replica_count,
                s.should_record_tensor, s.num_unpadded_tokens)
        if self.shared_slots is not None:                     # inject_shared_expert_weights, rocm_aiter_moe.py:113-158
            # The routed scaling factor applied to the OUTPUT below must not scale the shared expert: the
            # reference gives the shared slots the weight 1/routed_scaling_factor in that case
            # (fused_moe/layer.py:306-318), so that x * rsf leaves them at 1.
            r_ = self.routing
            comp = 1.0 / r_.routed_scaling_factor if (r_.routed_scaling_factor != 1.0
                                                      and not r_.apply_routed_scaling_in_router) else 1.0
            gate = None
            if self.shared_gate_weight is not None:           # qwen2_moe.py: sigmoid(shared_expert_gate(x)) * shared(x)
                g = torch.sigmoid(torch.nn.functional.linear(hidden_states, self.shared_gate_weight).float())
                gate = (g * comp).expand(M, self.shared_slots.n_shared)
            elif comp != 1.0:
                s = self.shared_slots
                gate = s.total_topk_weights.new_full((M, s.n_shared), s.shared_experts_score * comp)
            topk_weights, topk_ids = self.shared_slots.inject(topk_weights, topk_ids, gate)
        if self.expert_map is not None:                       # routed_experts.py:1332-1342
            topk_ids = self.ops.global_to_local_expert_ids(topk_ids, self.expert_map)
        if self.expert_parallel is not None:                  # rows -> owners -> local experts -> back, fp32 [M, H]
            out = self.expert_parallel.forward(hidden_states, topk_weights, topk_ids)
            if self.check_nan_in_output:
                torch.nan_to_num(out, nan=0.0, out=out)
            out = out.to(hidden_states.dtype)
        elif M <= self.max_num_seqs:                          # the cpu_decode contract: fp32 into the shared buffer
            buf = self._decode_buffer(hidden_states)
            out = self.engine.decode(hidden_states, topk_weights, topk_ids, out=buf[:M])
            if self.check_nan_in_output:                      # routed_experts.py:1852-1854
                torch.nan_to_num(out, nan=0.0, out=out)
            out = out.to(hidden_states.dtype)
        else:
            out = self.engine.prefill(hidden_states, topk_weights, topk_ids)
            if self.check_nan_in_output:                      # routed_experts.py:1895-1898
                out = torch.where(torch.isfinite(out), out, torch.zeros_like(out))
        if r.routed_scaling_factor != 1.0 and not r.apply_routed_scaling_in_router:   # moe_runner.py:391-408
            out *= r.routed_scaling_factor
        return out
