"""Layer tiers and HBM capacity planning.

The reference sorts every MoE layer into one of three tiers from two environment variables
(vllm/envs.py:2356-2407):
  * GPU-resident layers  -- `LVLLM_GPU_RESIDENT_MOE_LAYERS="0-5,7"`: experts live in VRAM, in-tree kernels;
  * GPU-prefill layers   -- `LVLLM_GPU_PREFILL_MIN_BATCH_SIZE > 0`: experts live in host RAM, decode on
    the CPU engine, batches >= the threshold are streamed through the GPU (routed_experts.py:1344-1357);
  * CPU layers           -- the rest: lk_moe on the NUMA CPUs.
The classification functions below restate that logic (same parsing of the layer list, same
precedence, MTP layers always resident) so that a host can keep its configuration surface.

On MI355X the tiers collapse: 288 GB of HBM3E per GPU hold every expert of the models the reference
targets, so the planning question becomes "do the experts of all layers fit next to the dense weights
and the KV cache, per GPU, at a given EP degree and weight format" -- `plan_hbm` answers it with the
engine's real (tile-padded) footprints.
"""
from __future__ import annotations

from dataclasses import dataclass


# ------------------------------------------------------------------ reference tier logic
def parse_layer_list(spec: str | None) -> set[int]:
    """`"0-5, 7,x,9-8"` -> {0..5, 7}: comma list of ints and inclusive ranges; malformed parts and
    descending ranges are ignored (vllm/envs.py:2383-2405)."""
    out: set[int] = set()
    if not spec:
        return out
    for part in spec.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            try:
                start, end = map(int, part.split("-"))
            except ValueError:
                continue
            if start <= end:
                out.update(range(start, end + 1))
        else:
            try:
                out.add(int(part))
            except ValueError:
                continue
    return out


def extract_layer_index(layer_name: str) -> int:
    """the single integer component of a dotted module name (vllm/model_executor/models/utils.py:917-938)"""
    vals = []
    for sub in layer_name.split("."):
        try:
            vals.append(int(sub))
        except ValueError:
            continue
    if len(vals) != 1:
        raise ValueError(f"layer name {layer_name} should only contain one integer")
    return vals[0]


def is_mtp_layer(layer_name: str) -> bool:
    return layer_name.startswith("mtp.")


@dataclass(frozen=True)
class TierConfig:
    """the three switches of the reference: LVLLM_MOE_NUMA_ENABLED, LVLLM_GPU_RESIDENT_MOE_LAYERS,
    LVLLM_GPU_PREFILL_MIN_BATCH_SIZE"""
    feature_enabled: bool = True
    resident_layers: str = ""
    gpu_prefill_min_batch_size: int = 0

    @classmethod
    def from_env(cls, env=None) -> "TierConfig":
        import os
        env = os.environ if env is None else env
        return cls(feature_enabled=str(env.get("LVLLM_MOE_NUMA_ENABLED", "0")).lower() in ("1", "true"),
                   resident_layers=env.get("LVLLM_GPU_RESIDENT_MOE_LAYERS", ""),
                   gpu_prefill_min_batch_size=int(env.get("LVLLM_GPU_PREFILL_MIN_BATCH_SIZE", "0") or 0))

    def is_gpu_resident_layer(self, layer_name: str) -> bool:
        if not self.feature_enabled or is_mtp_layer(layer_name):
            return True
        layer_id = extract_layer_index(layer_name)
        if not self.resident_layers:
            return False
        return layer_id in parse_layer_list(self.resident_layers)

    def is_gpu_prefill_layer(self, layer_name: str) -> bool:
        return (self.gpu_prefill_min_batch_size > 0 and not self.is_gpu_resident_layer(layer_name)
                and not is_mtp_layer(layer_name))

    def is_engine_layer(self, layer_name: str) -> bool:
        """envs.is_lk_moe_cpu_layer: the layers the reference hands to lk_moe's CPU path -- with this
        engine they are HBM-resident like the others, they only keep the lk_moe call surface."""
        return (self.feature_enabled and not self.is_gpu_resident_layer(layer_name)
                and not self.is_gpu_prefill_layer(layer_name) and not is_mtp_layer(layer_name))

    def should_use_gpu_prefill(self, layer_name: str, num_tokens: int, graph_capturing: bool = False) -> bool:
        """routed_experts.py:1344-1357: never under graph capture/replay, else by batch size"""
        if graph_capturing:
            return False
        return self.is_gpu_prefill_layer(layer_name) and num_tokens >= self.gpu_prefill_min_batch_size


# ------------------------------------------------------------------ HBM capacity planning
_BITS = {"bf16": 16.0, "fp16": 16.0, "fp8": 8.0, "int4": 4.0, "mxfp4": 4.0, "nvfp4": 4.0}


def expert_layer_bytes(num_local_experts: int, hidden: int, intermediate: int, fmt: str,
                       group_k: int = 128, gated: bool = True) -> int:
    """HBM bytes of ONE layer's experts in the engine's layout (lkm_common.h): rows padded to 64,
    K padded to the 64/128-element unit, plus the re-laid-out scales.  Matches lkm_weight_bytes()."""
    def rup(a, b):
        return -(-a // b) * b
    unit = 64 if fmt in ("bf16", "fp16") else 128
    halves = 2 if gated else 1
    n13, k13 = halves * rup(intermediate, 64), rup(hidden, unit)
    n2, k2 = rup(hidden, 64), rup(intermediate, unit)
    w = (n13 * k13 + n2 * k2) * _BITS[fmt] / 8
    units = n13 * (k13 // unit) + n2 * (k2 // unit)              # (row, unit) pairs
    if fmt == "fp8":
        s = units * 4
    elif fmt == "int4":
        s = units * (1 if group_k >= 128 else 128 // group_k) * 2
    elif fmt == "mxfp4":
        s = units * 4
    elif fmt == "nvfp4":
        s = units * 8
    else:
        s = 0
    return int(num_local_experts * (w + s))


@dataclass
class HbmPlan:
    per_gpu_expert_bytes: int
    per_gpu_total_bytes: int
    hbm_bytes: int
    fits: bool
    headroom_bytes: int
    min_ep_size: int | None      # smallest EP degree (dividing the experts, <= max_gpus) that fits; None = none


def plan_hbm(*, num_layers: int, num_experts: int, hidden: int, intermediate: int, fmt: str, ep_size: int = 1,
             group_k: int = 128, gated: bool = True, dense_bytes_per_gpu: int = 0, kv_cache_bytes_per_gpu: int = 0,
             hbm_bytes: int = 288 * 10**9, reserve_fraction: float = 0.08, max_gpus: int = 8) -> HbmPlan:
    """Experts are EP-sharded (num_experts / ep_size per GPU); `dense_bytes_per_gpu` and
    `kv_cache_bytes_per_gpu` are what the host framework needs next to them."""
    def need(ep):
        local = -(-num_experts // ep)
        return num_layers * expert_layer_bytes(local, hidden, intermediate, fmt, group_k, gated)
    budget = int(hbm_bytes * (1.0 - reserve_fraction))
    e = need(ep_size)
    total = e + dense_bytes_per_gpu + kv_cache_bytes_per_gpu
    min_ep = None
    for ep in range(1, max_gpus + 1):
        if num_experts % ep == 0 and need(ep) + dense_bytes_per_gpu + kv_cache_bytes_per_gpu <= budget:
            min_ep = ep
            break
    return HbmPlan(e, total, hbm_bytes, total <= budget, budget - total, min_ep)
