"""Expert-parallel load balancing for the HBM-resident expert tier (second half).

The reference parks cold experts in host DRAM; here every expert lives in one of the 8 x 288 GB HBM
stacks, so the capacity question of `lvllm_amd/residency.py` is followed by a placement question:
which rank holds which expert, and which hot experts get a second copy.  This module is the
host side of that loop, mirroring the reference's `vllm/distributed/eplb/` interface:

  rebalance_experts / DefaultEplbPolicy      policy/default.py:20-332  (replicate + pack, DeepSeek EPLB)
  compute_logical_maps                        eplb_state.py:1159-1235
  build_initial_global_physical_to_logical_map  eplb_state.py:297-314
  rearrange_expert_weights_inplace            rebalance_execute.py:511-616 (+ move_to_buffer :172-347)
  EplbState.step / rearrange / EplbLayerState  eplb_state.py:527-659, 722-930, 1087-1120
  eplb_map_to_physical_and_record             fused_moe/router/base_router.py:24-128  -> HIP kernel
                                              `eplb_map_record_kernel` (csrc/eplb.hip) via lvllm_amd.ops

What is re-designed for one MI355X node:
  * the policy is vectorised over layers (one greedy sweep serves every layer; the reference loops
    layer by layer in Python) and orders equal loads with a STABLE sort, so the placement is the same
    on every host (the reference's `np.argsort(-w)` picks a CPU-dependent order among exact ties;
    without ties the two agree index for index -- tests/test_eplb.py pins that against goldens
    produced by running the reference's policy);
  * weights move as ONE packed image per expert (`lkm_export_expert` / `lkm_import_expert`: the
    engine's pre-shuffled MFMA layout, all slabs of an expert contiguous) instead of one message per
    parameter tensor, several layers per `batch_isend_irecv` under a staging budget sized for 288 GB
    of HBM, and the senders of a replicated expert are chosen by least egress so that the 7 xGMI
    links of a GPU carry the exchange concurrently (xGMI is point to point: a single hot sender is
    link-bound);
  * load recording and the logical -> physical id map run in one integer HIP kernel after routing
    (bit-exact against the test suite's CPU restatement of the reference's kernel).

Pure host logic (numpy / torch.distributed); runs on gloo for the CPU tests.  No expert math here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Protocol, Sequence

import numpy as np
import torch
import torch.distributed as dist

__all__ = [
    "DefaultEplbPolicy", "rebalance_experts", "compute_logical_maps",
    "build_initial_global_physical_to_logical_map", "plan_layer_transfers", "LayerPlan",
    "ExpertStore", "TensorExpertStore", "EngineExpertStore", "rearrange_expert_weights_inplace",
    "plan_rearrangement", "begin_expert_exchange", "finish_expert_exchange", "PendingExchange",
    "EplbLayerState", "EplbState",
]


# ------------------------------------------------------------------------------------------ policy
def _as_2d_f(a) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"expected a [layers, n] array, got shape {a.shape}")
    return a


def _invert_rows(perm: np.ndarray) -> np.ndarray:
    inv = np.empty_like(perm)
    np.put_along_axis(inv, perm, np.broadcast_to(np.arange(perm.shape[1], dtype=perm.dtype), perm.shape), axis=1)
    return inv


class DefaultEplbPolicy:
    """Same entry points and results as the reference's DefaultEplbPolicy (policy/default.py)."""

    @classmethod
    def balanced_packing(cls, weight: np.ndarray, num_packs: int) -> tuple[np.ndarray, np.ndarray]:
        """n weighted items -> num_packs packs of exactly n/num_packs items, heaviest first into the
        lightest pack that still has room (first such pack on equal loads); policy/default.py:22-73.
        Returns (pack_index [X,n], rank_in_pack [X,n])."""
        weight = _as_2d_f(weight)
        X, n = weight.shape
        if num_packs <= 0 or n % num_packs:
            raise ValueError(f"{n} items do not divide into {num_packs} packs")
        per = n // num_packs
        if per == 1:
            idx = np.broadcast_to(np.arange(n, dtype=np.int64), (X, n)).copy()
            return idx, np.zeros_like(idx)
        order = np.argsort(-weight, axis=1, kind="stable")
        w64 = weight.astype(np.float64)
        load = np.zeros((X, num_packs), dtype=np.float64)
        fill = np.zeros((X, num_packs), dtype=np.int64)
        pack_index = np.empty((X, n), dtype=np.int64)
        rank_in_pack = np.empty((X, n), dtype=np.int64)
        rows = np.arange(X)
        for j in range(n):                      # one sweep position serves every row
            item = order[:, j]
            p = np.argmin(np.where(fill >= per, np.inf, load), axis=1)
            pack_index[rows, item] = p
            rank_in_pack[rows, item] = fill[rows, p]
            load[rows, p] += w64[rows, item]
            fill[rows, p] += 1
        return pack_index, rank_in_pack

    @classmethod
    def replicate_experts(cls, weight: np.ndarray, num_phy: int) -> tuple[np.ndarray, np.ndarray]:
        """num_log experts -> num_phy replicas, each extra replica to the expert with the largest
        load per replica (first on equal); policy/default.py:75-101.  Returns (phy2log [X,num_phy],
        logcnt [X,num_log])."""
        weight = _as_2d_f(weight)
        X, num_log = weight.shape
        if num_phy < num_log:
            raise ValueError(f"{num_phy} physical experts cannot hold {num_log} logical experts")
        w64 = weight.astype(np.float64)
        phy2log = np.empty((X, num_phy), dtype=np.int64)
        phy2log[:, :num_log] = np.arange(num_log, dtype=np.int64)
        logcnt = np.ones((X, num_log), dtype=np.int64)
        rows = np.arange(X)
        for slot in range(num_log, num_phy):
            hot = np.argmax(w64 / logcnt, axis=1)
            phy2log[:, slot] = hot
            logcnt[rows, hot] += 1
        return phy2log, logcnt

    @classmethod
    def rebalance_experts_hierarchical(cls, weight: np.ndarray, num_physical_experts: int, num_groups: int,
                                       num_nodes: int, num_gpus: int) -> np.ndarray:
        """groups -> nodes (balanced), replicas inside each node, physical experts -> the node's GPUs
        (balanced); policy/default.py:103-186.  Returns phy2log [layers, num_physical_experts]."""
        weight = _as_2d_f(weight)
        L, E = weight.shape
        if E % num_groups or num_groups % num_nodes or num_gpus % num_nodes or num_physical_experts % num_gpus:
            raise ValueError(f"indivisible EPLB geometry: experts {E}, groups {num_groups}, nodes {num_nodes}, "
                             f"gpus {num_gpus}, physical {num_physical_experts}")
        gsize = E // num_groups
        groups_per_node = num_groups // num_nodes
        e_node = E // num_nodes                   # logical experts per node
        p_node = num_physical_experts // num_nodes
        p_gpu = num_physical_experts // num_gpus
        # 1. groups -> nodes.  node-local logical position of expert e = (slot of its group) * gsize + e % gsize
        group_load = weight.reshape(L, num_groups, gsize).sum(axis=-1)
        g_node, g_rank = cls.balanced_packing(group_load, num_nodes)
        g_slot = g_node * groups_per_node + g_rank                                  # [L, groups]
        log2mlog = (g_slot[:, :, None] * gsize + np.arange(gsize, dtype=np.int64)).reshape(L, E)
        mlog2log = _invert_rows(log2mlog)
        # 2. replicas inside each node (rows = layer x node)
        node_load = np.take_along_axis(weight, mlog2log, axis=1).reshape(L * num_nodes, e_node)
        phy2mlog, mlogcnt = cls.replicate_experts(node_load, p_node)
        # 3. physical experts -> GPUs of the node, by load per replica
        phy_load = np.take_along_axis(node_load / mlogcnt, phy2mlog, axis=1)
        gpu, rank_on_gpu = cls.balanced_packing(phy_load, num_gpus // num_nodes)
        final_pos = gpu * p_gpu + rank_on_gpu                                        # [L*nodes, p_node]
        placed = np.take_along_axis(phy2mlog, _invert_rows(final_pos), axis=1)       # node-local logical per slot
        placed = placed.reshape(L, num_nodes, p_node) + (np.arange(num_nodes, dtype=np.int64) * e_node)[None, :, None]
        return np.take_along_axis(mlog2log, placed.reshape(L, num_physical_experts), axis=1)

    @classmethod
    def preserve_intragpu_slots(cls, phy2log: np.ndarray, num_ranks: int, old_phy2log: np.ndarray) -> np.ndarray:
        """Reorder each GPU's new experts so that an expert which stays on the GPU keeps its slot
        (no copy), the others fill the free slots in order; policy/default.py:188-271."""
        P = phy2log.shape[1]
        if num_ranks <= 0 or P % num_ranks or old_phy2log.shape != phy2log.shape:
            return phy2log
        per = P // num_ranks
        out = phy2log.copy()
        for layer in range(phy2log.shape[0]):
            for r in range(num_ranks):
                lo = r * per
                new = phy2log[layer, lo:lo + per].tolist()
                old = old_phy2log[layer, lo:lo + per].tolist()
                taken = [False] * per
                kept = [False] * per
                for s, want in enumerate(old):
                    for j, have in enumerate(new):
                        if have == want and not taken[j]:
                            out[layer, lo + s] = have
                            taken[j] = kept[s] = True
                            break
                rest = (new[j] for j in range(per) if not taken[j])
                for s in range(per):
                    if not kept[s]:
                        out[layer, lo + s] = next(rest)
        return out

    @classmethod
    def rebalance_experts(cls, weight, num_replicas: int, num_groups: int, num_nodes: int, num_ranks: int,
                          old_global_expert_indices=None) -> torch.Tensor:
        """[layers, logical] load statistics -> phy2log [layers, num_replicas] (int64 CPU tensor);
        policy/default.py:273-332.  Hierarchical when the groups divide over the nodes, else global."""
        w = weight.float().cpu().numpy() if isinstance(weight, torch.Tensor) else np.asarray(weight, dtype=np.float32)
        if num_groups % num_nodes == 0:
            phy2log = cls.rebalance_experts_hierarchical(w, num_replicas, num_groups, num_nodes, num_ranks)
        else:
            phy2log = cls.rebalance_experts_hierarchical(w, num_replicas, 1, 1, num_ranks)
        if old_global_expert_indices is not None:
            old = (old_global_expert_indices.cpu().numpy() if isinstance(old_global_expert_indices, torch.Tensor)
                   else np.asarray(old_global_expert_indices))
            phy2log = cls.preserve_intragpu_slots(phy2log, num_ranks, old)
        return torch.from_numpy(np.ascontiguousarray(phy2log))


rebalance_experts = DefaultEplbPolicy.rebalance_experts
EPLB_POLICIES = {"default": DefaultEplbPolicy}


def build_initial_global_physical_to_logical_map(num_routed_experts: int, num_redundant_experts: int) -> list[int]:
    """[the routed experts, then redundant slots cycling over them]; eplb_state.py:297-314."""
    return list(range(num_routed_experts)) + [i % num_routed_experts for i in range(num_redundant_experts)]


def compute_logical_maps(physical_to_logical_map: torch.Tensor, num_logical_experts: int,
                         max_slots: int | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """phy2log [layers, P] (or [P]) -> (log2phy [layers, E, R] padded with -1, replica count [layers, E]);
    replicas listed by ascending physical slot, slots holding -1 ignored; eplb_state.py:1159-1235.
    R = the largest replica count (the reference's shape) unless `max_slots` fixes it (a constant shape
    keeps device-side maps valid across rearrangements, cf. _commit_eplb_maps_for_layer :1245-1280)."""
    if physical_to_logical_map.device.type != "cpu":
        raise ValueError("compute_logical_maps works on CPU maps")
    one = physical_to_logical_map.dim() == 1
    p2l = physical_to_logical_map.unsqueeze(0) if one else physical_to_logical_map
    if p2l.dim() != 2:
        raise ValueError("physical_to_logical_map must be [layers, physical] or [physical]")
    a = p2l.numpy().astype(np.int64)
    L, P = a.shape
    E = num_logical_experts
    if a.size and a.max() >= E:
        raise ValueError(f"logical id {int(a.max())} out of range (num_logical_experts={E})")
    key = np.where(a < 0, E, a)                                   # unused slots sort last
    order = np.argsort(key, axis=1, kind="stable")                # physical slots grouped by logical id
    skey = np.take_along_axis(key, order, axis=1)
    cnt = np.zeros((L, E + 1), dtype=np.int64)
    np.add.at(cnt, (np.arange(L)[:, None], key), 1)
    start = np.concatenate([np.zeros((L, 1), np.int64), np.cumsum(cnt, axis=1)[:, :-1]], axis=1)
    replica = np.arange(P, dtype=np.int64)[None, :] - np.take_along_axis(start, skey, axis=1)
    cnt = cnt[:, :E]
    R = int(cnt.max()) if cnt.size else 0
    if max_slots is not None:
        if max_slots < R:
            raise ValueError(f"max_slots={max_slots} < largest replica count {R}")
        R = max_slots
    l2p = np.full((L, E, max(R, 0)), -1, dtype=np.int64)
    lay = np.broadcast_to(np.arange(L)[:, None], (L, P))
    ok = skey < E
    l2p[lay[ok], skey[ok], replica[ok]] = order[ok]
    dt = physical_to_logical_map.dtype
    l2p_t, cnt_t = torch.from_numpy(l2p).to(dt), torch.from_numpy(cnt).to(dt)
    return (l2p_t[0], cnt_t[0]) if one else (l2p_t, cnt_t)


# -------------------------------------------------------------------------- weight rearrangement
@dataclass
class LayerPlan:
    """Who sends what to whom for ONE layer; identical on every rank (pure function of the two maps).

    p2p    : (src_rank, src_slot, dst_rank, dst_slot, logical)   one per (destination rank, logical expert)
    local  : (rank, src_slot, dst_slot, logical)                 expert already on the rank, other slot
    fanout : (rank, primary_dst_slot, dst_slot, logical)         further local slots of an expert that arrived by p2p
    Slots are LOCAL slot numbers (0 .. P/ranks - 1).  Unchanged slots and slots whose new id is -1 do not appear."""
    p2p: list[tuple[int, int, int, int, int]] = field(default_factory=list)
    local: list[tuple[int, int, int, int]] = field(default_factory=list)
    fanout: list[tuple[int, int, int, int]] = field(default_factory=list)

    def egress(self, num_ranks: int) -> list[int]:
        out = [0] * num_ranks
        for s, *_ in self.p2p:
            out[s] += 1
        return out


def plan_layer_transfers(old_indices, new_indices, num_ranks: int, egress: list[int] | None = None) -> LayerPlan:
    """Transfers that turn the placement `old_indices` [P] into `new_indices` [P] (logical id per global
    physical slot, -1 = empty).  Result state == rebalance_execute.py:172-425 (every slot p ends up holding
    the weights of logical expert new[p]); the choice of sender differs: the reference deals a replicated
    expert's receivers to its holders in equal runs, here each receive goes to the holder with the least
    egress so far (ties: the holder that comes first), which spreads the exchange over the xGMI mesh.
    `egress` (images sent per rank so far, updated in place) carries the balance across the layers of one
    rearrangement, which travel together."""
    old = np.asarray(old_indices, dtype=np.int64).reshape(-1)
    new = np.asarray(new_indices, dtype=np.int64).reshape(-1)
    if old.shape != new.shape:
        raise ValueError("old and new placements differ in size")
    P = old.size
    if num_ranks <= 0 or P % num_ranks:
        raise ValueError(f"{P} physical slots do not divide over {num_ranks} ranks")
    per = P // num_ranks
    holders: dict[int, list[tuple[int, int]]] = {}            # logical -> [(rank, first local slot)] by rank
    for p in range(P):
        e = int(old[p])
        if e < 0:
            continue
        r = p // per
        hs = holders.setdefault(e, [])
        if not hs or hs[-1][0] != r:
            hs.append((r, p - r * per))
    plan = LayerPlan()
    if egress is None:
        egress = [0] * num_ranks
    for r in range(num_ranks):
        primary: dict[int, int] = {}                          # logical -> local slot that receives it
        local_src = {int(old[r * per + s]): s for s in range(per - 1, -1, -1) if old[r * per + s] >= 0}
        for s in range(per):
            e, was = int(new[r * per + s]), int(old[r * per + s])
            if e < 0 or e == was:
                continue
            if e in local_src:
                plan.local.append((r, local_src[e], s, e))
            elif e in primary:
                plan.fanout.append((r, primary[e], s, e))
            else:
                hs = holders.get(e)
                if not hs:
                    raise ValueError(f"logical expert {e} is needed at rank {r} but no rank holds it")
                src_rank, src_slot = min(hs, key=lambda h: egress[h[0]])
                egress[src_rank] += 1
                primary[e] = s
                plan.p2p.append((src_rank, src_slot, r, s, e))
    return plan


class ExpertStore(Protocol):
    """One layer's local experts as opaque fixed-size byte images (uint8 tensors on the store's device)."""
    num_local: int
    expert_nbytes: int
    device: torch.device

    def export_expert(self, slot: int, out: torch.Tensor) -> None: ...
    def import_expert(self, slot: int, src: torch.Tensor) -> None: ...


class TensorExpertStore:
    """ExpertStore over plain parameter tensors [num_local, ...] (the reference's `expert_weights` of a
    layer, rebalance_execute.py:515-523); used by the CPU tests and by callers that keep torch weights."""

    def __init__(self, tensors: Sequence[torch.Tensor]):
        if not tensors:
            raise ValueError("no tensors")
        self.tensors = list(tensors)
        self.num_local = self.tensors[0].size(0)
        for t in self.tensors:
            if t.size(0) != self.num_local or not t.is_contiguous():
                raise ValueError("expert tensors must be contiguous with the local experts on dim 0")
        self._sizes = [t[0].numel() * t.element_size() for t in self.tensors]
        self.expert_nbytes = sum(self._sizes)
        self.device = self.tensors[0].device

    def _views(self, slot: int):
        return [t[slot].reshape(-1).view(torch.uint8) for t in self.tensors]

    def export_expert(self, slot: int, out: torch.Tensor) -> None:
        off = 0
        for v, n in zip(self._views(slot), self._sizes):
            out[off:off + n].copy_(v, non_blocking=True)
            off += n

    def import_expert(self, slot: int, src: torch.Tensor) -> None:
        off = 0
        for v, n in zip(self._views(slot), self._sizes):
            v.copy_(src[off:off + n], non_blocking=True)
            off += n


class EngineExpertStore:
    """ExpertStore over an engine (lvllm_amd.ops.RoutedExpertsEngine or a bare lk_moe.MOE_* object): an
    expert's packed image is its slabs of the engine's pre-shuffled HBM layout, copied device
    to device on the current stream by lkm_export_expert / lkm_import_expert.  Nothing is re-shuffled: the
    image already is the MFMA operand layout, valid for every engine of the same configuration."""

    def __init__(self, engine, num_local: int | None = None, device: torch.device | str | None = None):
        moe = getattr(engine, "engine", engine)            # RoutedExpertsEngine wraps the lk_moe object
        cfg = getattr(engine, "cfg", None)
        if num_local is None:
            if cfg is None:
                raise ValueError("num_local is required for a bare lk_moe engine object")
            num_local = cfg.expert_num
        # An engine with folded shared experts holds routed + n_shared slots (shared_experts.py): pass
        # num_local = the ROUTED physical slots; the trailing shared slots are replicated on every rank and never move.
        self.moe = moe
        self.num_local = int(num_local)
        self.expert_nbytes = moe.expert_bytes()
        if device is None:
            device = torch.device("cuda", cfg.gpu_id if cfg is not None else torch.cuda.current_device())
        self.device = torch.device(device)

    def _check(self, t: torch.Tensor) -> None:
        if not (t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.numel() >= self.expert_nbytes):
            raise ValueError(f"expert image must be a contiguous uint8 device tensor of >= {self.expert_nbytes} bytes")

    def export_expert(self, slot: int, out: torch.Tensor) -> None:
        self._check(out)
        self.moe.export_expert(torch.cuda.current_stream(out.device).cuda_stream, slot, out.data_ptr())

    def import_expert(self, slot: int, src: torch.Tensor) -> None:
        self._check(src)
        self.moe.import_expert(torch.cuda.current_stream(src.device).cuda_stream, slot, src.data_ptr())


def _post(sends: list[tuple[torch.Tensor, int]], recvs: list[tuple[torch.Tensor, int]], group) -> list:
    """post the point-to-point transfers of one batch; returns the outstanding works (not waited)"""
    def peer_of(r: int) -> int:
        return dist.get_global_rank(group, r) if group is not None else r
    ops = [dist.P2POp(dist.isend, t, peer_of(r), group) for t, r in sends]
    ops += [dist.P2POp(dist.irecv, t, peer_of(r), group) for t, r in recvs]
    return list(dist.batch_isend_irecv(ops)) if ops else []


@dataclass
class PendingExchange:
    """One batch of layers in flight: images exported and transfers posted, nothing imported yet.  The engine's
    weights are untouched until `finish_expert_exchange`, so forwards may keep running on the old placement."""
    layers: range
    works: list
    commits: list            # (layer, dst_slot, image)        images that arrived by p2p or by a local move
    fanout: list             # (layer, image, dst_slot)        further local slots of an image that arrived by p2p
    keepalive: list          # send images: must outlive the transfers

    def is_completed(self) -> bool:
        """NB gloo's send/recv works only report completion after wait(); meaningful with RCCL"""
        return all(w.is_completed() for w in self.works)


def _layer_batches(plans: list[LayerPlan], stores: Sequence[ExpertStore], world: int, budget: int) -> list[range]:
    """consecutive layers whose staged images (outgoing + incoming) fit `budget` bytes; at least one layer per
    batch.  Every rank must cut the SAME batches (a batch is one collective batch_isend_irecv), so the budget is
    compared against the busiest rank's need, which every rank computes from the global plans."""

    def need(l: int) -> int:
        pl, worst = plans[l], 0
        for r in range(world):
            out_slots = {s for sr, s, *_ in pl.p2p if sr == r}
            inc = sum(1 for _, _, dr, *_ in pl.p2p if dr == r) + sum(1 for rr, *_ in pl.local if rr == r)
            worst = max(worst, len(out_slots) + inc)
        return worst * stores[l].expert_nbytes
    out, l0, L = [], 0, len(plans)
    while l0 < L:
        l1, used = l0, 0
        while l1 < L:
            n = need(l1)
            if l1 > l0 and used + n > budget:
                break
            used += n
            l1 += 1
        out.append(range(l0, l1))
        l0 = l1
    return out


def begin_expert_exchange(plans: list[LayerPlan], layers: range, expert_stores: Sequence[ExpertStore], rank: int,
                          ep_group=None) -> PendingExchange:
    """Export the outgoing experts of `layers` (once per slot) and post ALL their transfers in one
    batch_isend_irecv.  Every incoming image (remote or local move) is staged; no slot is overwritten here."""
    sends: list[tuple[torch.Tensor, int]] = []
    recvs: list[tuple[torch.Tensor, int]] = []
    commits, fan, keep = [], [], []
    for l in layers:
        st, pl = expert_stores[l], plans[l]
        nb = st.expert_nbytes
        packed: dict[int, torch.Tensor] = {}

        def image_of(slot: int, st=st, nb=nb, packed=packed) -> torch.Tensor:
            if slot not in packed:
                buf = torch.empty(nb, dtype=torch.uint8, device=st.device)
                st.export_expert(slot, buf)
                packed[slot] = buf
            return packed[slot]
        arrived: dict[int, torch.Tensor] = {}                  # primary dst slot -> staged image
        for sr, ss, dr, ds, _e in pl.p2p:                      # the plan's order: identical on sender and receiver
            if sr == rank:
                sends.append((image_of(ss), dr))
            if dr == rank:
                buf = torch.empty(nb, dtype=torch.uint8, device=st.device)
                recvs.append((buf, sr))
                arrived[ds] = buf
                commits.append((l, ds, buf))
        for r, ss, ds, _e in pl.local:
            if r == rank:
                commits.append((l, ds, image_of(ss)))
        for r, prim, ds, _e in pl.fanout:
            if r == rank:
                fan.append((l, arrived[prim], ds))
        keep.extend(packed.values())
    if sends or recvs:
        dev = (sends or recvs)[0][0].device
        if dev.type == "cuda":                                 # the exports ran on the current stream
            torch.cuda.current_stream(dev).synchronize()
    return PendingExchange(layers, _post(sends, recvs, ep_group), commits, fan, keep)


def finish_expert_exchange(pending: PendingExchange, expert_stores: Sequence[ExpertStore]) -> None:
    """wait for the batch's transfers, then import the staged images into their slots"""
    for w in pending.works:
        w.wait()
    for l, ds, img in pending.commits:
        expert_stores[l].import_expert(ds, img)
    for l, img, ds in pending.fanout:
        expert_stores[l].import_expert(ds, img)
    pending.works, pending.commits, pending.fanout, pending.keepalive = [], [], [], []


def _np_maps(old_global_expert_indices, new_global_expert_indices, n_layers: int):
    old = np.asarray(old_global_expert_indices.cpu() if isinstance(old_global_expert_indices, torch.Tensor)
                     else old_global_expert_indices, dtype=np.int64)
    new = np.asarray(new_global_expert_indices.cpu() if isinstance(new_global_expert_indices, torch.Tensor)
                     else new_global_expert_indices, dtype=np.int64)
    if old.shape != new.shape or old.ndim != 2 or old.shape[0] != n_layers:
        raise ValueError("old/new must be [layers, physical] with one expert store per layer")
    return old, new


def plan_rearrangement(old_global_expert_indices, new_global_expert_indices, expert_stores: Sequence[ExpertStore],
                       world: int) -> list[LayerPlan]:
    old, new = _np_maps(old_global_expert_indices, new_global_expert_indices, len(expert_stores))
    L, P = old.shape
    if P % world:
        raise ValueError(f"{P} physical experts do not divide over {world} ranks")
    for l, st in enumerate(expert_stores):
        if st.num_local != P // world:
            raise ValueError(f"layer {l}: store holds {st.num_local} experts, placement has {P // world} per rank")
    egress = [0] * world                                  # balanced over the whole rearrangement, not per layer
    return [plan_layer_transfers(old[l], new[l], world, egress) for l in range(L)]


def rearrange_expert_weights_inplace(old_global_expert_indices, new_global_expert_indices,
                                     expert_stores: Sequence[ExpertStore], ep_group=None,
                                     max_staging_bytes: int = 8 << 30, rank: int | None = None,
                                     world: int | None = None) -> list[LayerPlan]:
    """Move expert weights so that layer l's physical slot p holds logical expert new[l, p]
    (rebalance_execute.py:511-616).  old/new: [layers, P] logical ids; expert_stores[l]: this rank's
    experts of layer l -- an ExpertStore, or (the reference's `expert_weights[l]`) a sequence of parameter tensors
    [num_local, ...], updated in place.  Collective over ep_group (every rank calls it with the same maps).

    Per batch of layers (as many as fit `max_staging_bytes` of staging on the busiest rank): pack outgoing
    experts once per slot -> ONE batch_isend_irecv for the whole batch -> import staged images into their
    slots.  All incoming images (remote and local moves) are staged before any slot is overwritten, so a slot can
    be both a source and a destination.  Returns the per-layer plans (for logging / tests)."""
    if world is None:
        world = dist.get_world_size(ep_group) if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank(ep_group) if dist.is_initialized() else 0
    # the reference passes `expert_weights`: per layer a sequence of parameter tensors [num_local, ...]
    expert_stores = [st if hasattr(st, "export_expert") else TensorExpertStore(st) for st in expert_stores]
    plans = plan_rearrangement(old_global_expert_indices, new_global_expert_indices, expert_stores, world)
    for layers in _layer_batches(plans, expert_stores, world, max_staging_bytes):
        finish_expert_exchange(begin_expert_exchange(plans, layers, expert_stores, rank, ep_group), expert_stores)
    for st in expert_stores:
        if st.device.type == "cuda":
            torch.cuda.current_stream(st.device).synchronize()
            break
    return plans


# ------------------------------------------------------------------------------------------ state
@dataclass
class EplbLayerState:
    """What the router of one layer reads each step (eplb_state.py:1087-1120)."""
    expert_load_view: torch.Tensor | None = None            # int32 [P]   += tokens per physical expert
    logical_to_physical_map: torch.Tensor | None = None     # int32 [E, R]
    logical_replica_count: torch.Tensor | None = None       # int32 [E]
    should_record_tensor: torch.Tensor | None = None        # int32 scalar (device): graph-safe switch
    num_unpadded_tokens: torch.Tensor | None = None         # int32 scalar (device) or None


class EplbState:
    """Load window + periodic rearrangement for one model's MoE layers (eplb_state.py:220-930, the
    synchronous path; the reference's async worker / elastic rank re-mapping / NIXL transports are
    out of scope).  One instance per rank; `step()` is called once per forward on every rank."""

    def __init__(self, num_layers: int, num_logical_experts: int, num_redundant_experts: int, *,
                 num_groups: int = 1, num_nodes: int = 1, window_size: int = 1000, step_interval: int = 3000,
                 device: torch.device | str = "cpu", ep_group=None, policy=DefaultEplbPolicy,
                 expert_stores: Sequence[ExpertStore] | None = None, overlap: bool = False,
                 commit_after_steps: int = 1, max_staging_bytes: int = 8 << 30,
                 initial_physical_to_logical_map: torch.Tensor | None = None):
        self.group = ep_group
        self.expert_stores = expert_stores                   # one per layer; may be attached later
        # overlap: the exchange of a batch of layers is only POSTED at the step that is due; forwards keep running
        # on the old placement (weights and maps untouched) while the images travel; `commit_after_steps` steps
        # later the batch is imported and its layers' maps are switched, and the next batch is posted (the
        # reference's async mode, eplb_state.py:624-650 / async_worker.py, without a worker thread: RCCL p2p runs
        # on the communicator's own stream).  Give EPLB its own process group then (dist.new_group): its p2p
        # traffic must not interleave with the forward's all-to-all on one communicator.
        self.overlap = overlap
        self.commit_after_steps = max(1, commit_after_steps)
        self._pending_age = 0
        self.max_staging_bytes = max_staging_bytes
        self._pending: PendingExchange | None = None
        self._todo: list[range] = []
        self._plans: list[LayerPlan] = []
        self._new_p2l: torch.Tensor | None = None
        self.world = dist.get_world_size(ep_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(ep_group) if dist.is_initialized() else 0
        self.L, self.E = num_layers, num_logical_experts
        self.P = num_logical_experts + num_redundant_experts
        if self.P % self.world:
            raise ValueError(f"{self.P} physical experts do not divide over {self.world} ranks")
        self.R = num_redundant_experts + 1                   # widest possible replica list: constant map shape
        self.num_groups, self.num_nodes = num_groups, num_nodes
        self.window_size, self.step_interval = window_size, step_interval
        self.policy = policy
        self.device = torch.device(device)
        if initial_physical_to_logical_map is not None:      # resume a saved placement (cf. EplbState.from_mapping, :1043-1085)
            p2l0 = initial_physical_to_logical_map.detach().to("cpu", torch.int64).reshape(num_layers, -1).clone()
            if p2l0.size(1) != self.P or int(p2l0.max()) >= self.E:
                raise ValueError(f"initial placement must be [{num_layers}, {self.P}] with ids below {self.E}")
            for row in p2l0:
                if set(row[row >= 0].tolist()) != set(range(self.E)):
                    raise ValueError("initial placement: every logical expert needs at least one physical slot")
            self.physical_to_logical_map = p2l0
        else:
            init = build_initial_global_physical_to_logical_map(num_logical_experts, num_redundant_experts)
            self.physical_to_logical_map = torch.tensor(init, dtype=torch.int64).repeat(num_layers, 1)   # CPU
        i32 = dict(dtype=torch.int32, device=self.device)
        self.logical_to_physical_map = torch.full((self.L, self.E, self.R), -1, **i32)
        self.logical_replica_count = torch.zeros((self.L, self.E), **i32)
        self.expert_load_pass = torch.zeros((self.L, self.P), **i32)
        self.expert_load_window = torch.zeros((window_size, self.L, self.P), **i32)
        self.should_record = torch.ones((), **i32)
        self.window_step = 0
        self.rearrangement_step = 0
        self._commit(self.physical_to_logical_map)

    # ---- per-layer view for the router
    def layer_state(self, layer: int) -> EplbLayerState:
        return EplbLayerState(self.expert_load_pass[layer], self.logical_to_physical_map[layer],
                              self.logical_replica_count[layer], self.should_record, None)

    def local_logical_ids(self, layer: int) -> list[int]:
        per = self.P // self.world
        return self.physical_to_logical_map[layer, self.rank * per:(self.rank + 1) * per].tolist()

    def _commit(self, p2l: torch.Tensor, layers: range | None = None) -> None:
        """new placement (of all layers, or of `layers` only) -> the device maps, IN PLACE (captured graphs keep
        reading the same buffers; cf. _commit_eplb_maps_for_layer, eplb_state.py:1245-1280)."""
        if layers is None:
            layers = range(self.L)
            self.physical_to_logical_map = p2l.clone()
        else:
            self.physical_to_logical_map[layers.start:layers.stop] = p2l[layers.start:layers.stop]
        sel = slice(layers.start, layers.stop)
        l2p, cnt = compute_logical_maps(p2l[sel], self.E, max_slots=self.R)
        self.logical_to_physical_map[sel].copy_(l2p.to(torch.int32))
        self.logical_replica_count[sel].copy_(cnt.to(torch.int32))

    # ---- per forward
    def step(self, is_dummy: bool = False) -> bool:
        """Close this forward's load pass into the window; every `step_interval` steps rearrange
        (collective: every rank steps in lockstep, dummy steps included).  Returns True when a
        rearrangement ran.  eplb_state.py:527-659."""
        if is_dummy:
            self.expert_load_pass.zero_()
        else:
            self.expert_load_window[self.window_step].copy_(self.expert_load_pass)
            self.expert_load_pass.zero_()
            self.window_step = (self.window_step + 1) % self.window_size
        self.rearrangement_step += 1
        changed = self._advance_overlapped() if self.in_flight else False
        if self.rearrangement_step >= self.step_interval:
            if self.in_flight:                       # still moving the previous round: keep the counter (eplb_state.py:641-650)
                return changed
            self.rearrangement_step = 0
            if self.overlap:
                self._begin_overlapped()
            else:
                self.rearrange()
                changed = True
        return changed

    # ---- overlapped rearrangement
    @property
    def in_flight(self) -> bool:
        return self._pending is not None

    def _begin_overlapped(self) -> None:
        stores = self._stores()
        load = self.global_logical_load()
        self._new_p2l = self.policy.rebalance_experts(load, self.P, self.num_groups, self.num_nodes, self.world,
                                                      self.physical_to_logical_map)
        self._plans = plan_rearrangement(self.physical_to_logical_map, self._new_p2l, stores, self.world)
        self._todo = _layer_batches(self._plans, stores, self.world, self.max_staging_bytes)
        self._pending = begin_expert_exchange(self._plans, self._todo.pop(0), stores, self.rank, self.group)
        self._pending_age = 0

    def _advance_overlapped(self, force: bool = False) -> bool:
        """`commit_after_steps` steps after a batch was posted: wait for its transfers (with RCCL that is a stream
        dependency, not a host stall), import it, switch ITS layers' maps, post the next batch.  Every rank does this
        at the same step by construction, so -- unlike the reference, which all-reduces a "transfers done" flag every
        step while a layer is in flight (eplb_state.py:984-1003) -- no agreement collective and no host
        synchronisation are needed; a transfer that is not finished by then delays that step instead."""
        self._pending_age += 1
        if not force and self._pending_age < self.commit_after_steps:
            return False
        stores = self._stores()
        finish_expert_exchange(self._pending, stores)
        self._commit(self._new_p2l, self._pending.layers)
        self._pending_age = 0
        self._pending = (begin_expert_exchange(self._plans, self._todo.pop(0), stores, self.rank, self.group)
                         if self._todo else None)
        return True

    def drain(self) -> None:
        """finish an overlapped rearrangement now (collective; e.g. before shutdown or a checkpoint)"""
        while self._pending is not None:
            self._advance_overlapped(force=True)

    def _stores(self) -> Sequence[ExpertStore]:
        if self.expert_stores is None or len(self.expert_stores) != self.L:
            raise RuntimeError("EplbState needs one expert store per MoE layer (expert_stores) to rearrange")
        return self.expert_stores

    def global_logical_load(self) -> torch.Tensor:
        """window summed over steps, physical -> logical, summed over ranks: float32 [L, E] on CPU
        (eplb_state.py:754-780)."""
        phys = self.expert_load_window.sum(dim=0, dtype=torch.int64)                      # [L, P]
        logical = torch.zeros((self.L, self.E), dtype=torch.int64, device=phys.device)
        idx = self.physical_to_logical_map.to(phys.device)
        logical.scatter_add_(1, idx.clamp(min=0), torch.where(idx >= 0, phys, torch.zeros_like(phys)))
        if self.world > 1:
            # the collective runs where the group's backend can reach the tensor (gloo: host memory)
            backend = dist.get_backend(self.group)
            if "nccl" not in str(backend) and logical.is_cuda:
                logical = logical.cpu()
            dist.all_reduce(logical, group=self.group)
        return logical.float().cpu()

    def rearrange(self, max_staging_bytes: int | None = None) -> list[LayerPlan]:
        """load statistics -> policy -> weight exchange -> commit maps (eplb_state.py:722-930)."""
        expert_stores = self._stores()
        if self.in_flight:
            raise RuntimeError("an overlapped rearrangement is in flight: drain() first")
        load = self.global_logical_load()
        new_p2l = self.policy.rebalance_experts(load, self.P, self.num_groups, self.num_nodes, self.world,
                                                self.physical_to_logical_map)
        plans = rearrange_expert_weights_inplace(self.physical_to_logical_map, new_p2l, expert_stores, self.group,
                                                 max_staging_bytes or self.max_staging_bytes, rank=self.rank,
                                                 world=self.world)
        self._commit(new_p2l)
        return plans

    # ---- statistics (eplb_state.py:566-607)
    def balancedness(self) -> float:
        """mean / max of the per-rank token load of the last recorded step, summed over layers."""
        last = self.expert_load_window[(self.window_step - 1) % self.window_size].to(torch.int64)
        if self.world > 1:
            last = last.clone()
            dist.all_reduce(last, group=self.group)
        per_rank = last.view(self.L, self.world, -1).sum(dim=-1).float()                  # [L, ranks]
        mx = per_rank.max(dim=1).values.sum().item()
        return float(per_rank.mean(dim=1).sum().item() / mx) if mx > 0 else 0.0
