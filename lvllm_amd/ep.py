"""Expert-parallel sharding of the routed experts across the GPUs of one MI355X node.

This replaces the reference's CPU-NUMA offload tier: instead of parking experts in host DRAM
(lk_moe) each rank keeps E/ep experts resident in its 288 GB of HBM.  Two data paths:

  "a2a"  (default, BASELINE.json north_star)  tokens stay DP-sharded, each token visits the ranks that own
         its experts and comes back as one partial row per visited rank.  Two flavours, chosen by size:
           fixed   (decode): TOKEN-granular records [activations | top-k ids | top-k weights], one per
                   (token, destination rank) -- a token travels to a rank at most once however many of its
                   experts live there -- in `capacity` record slots per destination.  capacity = the token
                   count is the exact worst case, so nothing can overflow: ONE equal-split all-to-all out,
                   ONE back (the owner's engine already formed the weighted sum over its experts; the row
                   returns in the activation dtype by default), no split-size exchange, no host sync -- the
                   whole step (pack kernel, RCCL, grouped GEMMs, RCCL, combine kernel) is capturable in a
                   hipGraph, which is the only way the reference ever calls its decode engine
                   (moe_runner.py:609-614).  With DeepSeek-V3 routing (8 experts out of <= 4 groups = ranks
                   at ep 8) the bytes on the wire are <= the routed-row bytes M*K*H*2 of the slot-granular
                   exchanges the reference uses (all2all.py:101-150); `wire_bytes()` reports them.
           ragged  (prefill): exact split sizes are exchanged first, only routed rows travel.
  "ar"   reference-compatible mode (what LvLLM does today, moe_runner.py:600,494 and
         routed_experts.py:1332-1342): every rank sees ALL tokens (all_gather), computes only
         its local experts (other ids -> -1), and the [M,H] partial outputs are summed with
         reduce_scatter (= the reference's all-reduce, kept sharded).

Contract on token counts: the fixed path and "ar" use equal-split collectives, so every rank of the group must
pass the SAME capacity (fixed) / token count ("ar") in the same step -- what vLLM guarantees for captured decode
steps by padding the DP ranks to a common count (num_tokens_across_dp).  `forward(..., capacity=C)` (or the
constructor's `capacity_tokens`) names the common value explicitly; a rank may then hold any M <= C tokens,
including none.  The fixed-vs-ragged decision is made from that capacity, never from the rank-local M.  The ragged
path exchanges its sizes and takes any M.  `validate_uniform=True` checks the contract with an all-gather (eager
debugging aid; it synchronises the host).

Capacity below the worst case.  With group-limited routing (DeepSeek-V3 / GLM: top-k out of `topk_group` of `n_group`
expert groups, groups laid out rank by rank) a token visits at most topk_group * max(1, ep / n_group) ranks, so a
destination receives about M * that / ep records from a source of M tokens, not M.  `routing_groups=(n_group,
topk_group)` makes capacity = ceil(M * ranks_per_token / ep * capacity_slack) the default (BASELINE configs[3]: 0.625 M
at slack 1.25 -> 0.625 x the bytes on the wire, and the owner's engine scans 0.625 x the record slots).  The pack
kernel COUNTS what does not fit (it never drops silently); an eager step notices (`check_overflow=True`, one host read)
and repeats itself at the exact bound, a captured step cannot branch: its caller reads `overflow_count()` after the
replay and re-runs that step eagerly (`capacity=M`) when it moved.

Placement is the reference's linear map (expert_map_manager.py:62-79).  With EPLB (lvllm_amd/eplb.py) the ids
handed to forward() are PHYSICAL expert ids and `num_experts` is the number of physical slots: the linear map over
the slots is exactly the EPLB placement (slot p lives on rank p // (P / ep)); tests/test_eplb.py runs that end to
end.  `torch.distributed` with backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests
(tests/test_ep_gloo.py), where the local expert computation and the two exchange kernels are injected.
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist

from .ops import determine_expert_map

# local_compute(rows [R,H] act dtype (row-strided view), ids int32 [R,Kx], weights fp32 [R,Kx], out_dtype) ->
# [R,H] contiguous in out_dtype: for every row the weighted sum over the LOCAL experts among its Kx ids (< 0 = skip)
# A callable that also declares a keyword `valid_den` is told how sparse the records are (ep x capacity slots of which
# ~1/valid_den carry a local id): the engine plans its tiles for the rows that exist (ops.forward_rows(valid_den=...)).
LocalCompute = Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.dtype], torch.Tensor]
# transport(out, inp): equal-split all-to-all over the group on tensors of shape [ep, ...]
Transport = Callable[[torch.Tensor, torch.Tensor], None]


# Exchange buffers: ONE grow-only pool per (device, group, pool tag), shared by every layer's ExpertParallelExperts with
# that tag -- the layers of a model run one after the other on a stream, so they can exchange through the same memory; a
# pool that was handed out stays alive (captured graphs replay on its addresses), a larger request allocates a new one.
# Two micro-batches in flight (overlapped layers / dual-batch overlap on two streams) need two pools: give their
# ExpertParallelExperts different `pool_tag`s.  A pool remembers whether a dispatch is waiting for its combine and refuses a
# second dispatch until then (the guard used to be per instance while the memory is per pool).
_POOLS: dict = {}
_RETIRED: list = []
_POOL_GROUPS: dict = {}      # id(group) -> group: pins the object, so the id in a pool key cannot be reused by a new group
_IN_FLIGHT: dict = {}        # (device, id(group), tag) -> description of the dispatch that has not been combined yet


def _pool_key(dev, group, tag: str) -> tuple:
    if group is not None:
        _POOL_GROUPS[id(group)] = group
    return (str(dev), id(group), tag)


def _pool(dev, group, name: str, nbytes: int, tag: str = "") -> torch.Tensor:
    key = _pool_key(dev, group, tag) + (name,)
    t = _POOLS.get(key)
    if t is None or t.numel() < nbytes:
        if t is not None:
            _RETIRED.append(t)                      # (a captured graph may still replay on it)
            nbytes = max(nbytes, t.numel() * 3 // 2)   # geometric growth: a rising batch size retires O(log) pools
        t = torch.empty((max(nbytes, 1),), dtype=torch.uint8, device=dev)
        _POOLS[key] = t
    return t


def release_group_exchange_buffers(group) -> None:
    """forget every pool of a process group that is being destroyed (its id may then be reused)"""
    gid = id(group)
    for k in [k for k in _POOLS if k[1] == gid]:
        del _POOLS[k]
    for k in [k for k in _IN_FLIGHT if k[1] == gid]:
        del _IN_FLIGHT[k]
    _POOL_GROUPS.pop(gid, None)


def release_retired_exchange_buffers() -> int:
    """drop the pools that were outgrown (call when no captured graph that used them will be replayed again); returns
    the bytes released"""
    n = sum(t.numel() for t in _RETIRED)
    _RETIRED.clear()
    return n


def ranks_per_token(ep: int, n_group: int, topk_group: int, top_k: int | None = None) -> int:
    """upper bound of the ranks one token's experts live on under group-limited routing with the groups laid out
    rank by rank (linear placement)"""
    r = topk_group * max(1, -(-ep // max(n_group, 1)))
    if top_k is not None:
        r = min(r, top_k)
    return max(1, min(ep, r))


def owner_of(ids: torch.Tensor, num_experts: int, ep_size: int) -> torch.Tensor:
    """rank that owns each global expert id under linear placement (remainder to the first ranks,
    expert_map_manager.py:66-74); ids < 0 -> -1."""
    base, rem = divmod(num_experts, ep_size)
    idl = ids.to(torch.int64)
    cut = rem * (base + 1)
    if base > 0:
        r = torch.where(idl < cut, idl // (base + 1), rem + (idl - cut) // base)
    else:
        r = idl // (base + 1)
    return torch.where(idl < 0, torch.full_like(idl, -1), r)


def _capturing(t: torch.Tensor) -> bool:
    """a host read of the overflow counter is impossible while the step is being captured into a hipGraph"""
    return bool(t.is_cuda and torch.cuda.is_current_stream_capturing())


class ExpertParallelExperts:
    def __init__(self, local_compute: LocalCompute, num_experts: int, hidden_size: int,
                 group: dist.ProcessGroup | None = None, mode: str = "a2a", kernels=None,
                 transport: Transport | None = None, fixed_max_tokens: int = 1024,
                 capacity_tokens: int | None = None, return_dtype: torch.dtype | None = None,
                 global_ids: bool = False, validate_uniform: bool = False,
                 routing_groups: tuple[int, int] | None = None, capacity_slack: float = 1.25,
                 check_overflow: bool | None = None, pool_tag: str = "", capture_slack: float = 1.75):
        """kernels: namespace with ep_row_bytes / ep_pack_tokens / ep_combine (default lvllm_amd.ops = the HIP
        kernels, no CPU path; the gloo tests inject torch doubles).  transport: equal-split all-to-all (default
        dist.all_to_all_single over `group`).  fixed_max_tokens: largest capacity served by the fixed path.
        capacity_tokens: the group-wide record capacity per destination (default: this step's token count).
        return_dtype: dtype of the partial rows on the way back (default: the activation dtype; torch.float32
        keeps the single-rank fp32 sum exactly).  global_ids: records carry GLOBAL expert ids (for a receiver
        that applies expert_map itself, modular.LkmPrepareAndFinalize) instead of ids local to the owner.
        routing_groups = (n_group, topk_group), capacity_slack, check_overflow: the capacity below the worst case and
        its fallback (module docstring).  check_overflow defaults to True whenever routing_groups is given: an eager
        step then never returns rows with dropped contributions (one 8-byte all-reduce + host read per step); a step
        being captured cannot branch and only counts -- its caller MUST read overflow_count() after the replay.
        check_overflow=False is the explicit opt-in to count silently in eager steps too.
        pool_tag: which exchange-buffer pool of the (device, group) this instance uses (two micro-batches in flight
        need two tags).
        capture_slack: the slack of the group-limited capacity while the step is being CAPTURED (a graph cannot branch
        on the overflow counter, so its capacity is sized for < 1e-4 overflow probability per (source, destination)
        pair instead of the eager path's 5 %: DeepSeek-V3 routing at 32 tokens per rank, Binomial(32, 1/2) records per
        pair, capacity ceil(32 * 4 * 1.75 / 8) = 28 -> P(X > 28) = 1.3e-6; the caller still reads overflow_count()
        after the replay)."""
        if mode not in ("a2a", "ar"):
            raise ValueError(f"unknown EP mode {mode!r} (expected 'a2a' or 'ar')")
        self.local_compute = local_compute
        if kernels is None:
            from . import ops as kernels           # the HIP kernels (no CPU path)
        self.kernels = kernels
        self.fixed_max_tokens = fixed_max_tokens
        self.capacity_tokens = capacity_tokens
        self.return_dtype = return_dtype
        self.global_ids = global_ids
        self.validate_uniform = validate_uniform
        self.routing_groups = routing_groups
        self.capacity_slack = capacity_slack
        self.check_overflow = (routing_groups is not None) if check_overflow is None else bool(check_overflow)
        self.capture_slack = max(capture_slack, capacity_slack)
        self.pool_tag = pool_tag
        self._overflow_seen = 0              # (device counter value at the last check)
        import inspect
        try:
            self._lc_takes_den = "valid_den" in inspect.signature(local_compute).parameters
        except (TypeError, ValueError):
            self._lc_takes_den = False
        self.E, self.H = num_experts, hidden_size
        self.group = group
        self.ep = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.mode = mode
        self.transport = transport if transport is not None else self._dist_a2a
        self.local_num, emap = determine_expert_map(self.ep, self.rank, num_experts, "linear")
        self.expert_map = emap if emap is not None else torch.arange(num_experts, dtype=torch.int32)
        base, rem = divmod(num_experts, self.ep)
        self.first_expert = [r * base + min(r, rem) for r in range(self.ep)]
        self._overflow_bufs: dict = {}
        self._uniform_checked: set = set()
        self.last_wire: dict | None = None

    def _dist_a2a(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        dist.all_to_all_single(out, inp, group=self.group)

    # -------------------------------------------------------------------------------- a2a, fixed
    def capacity_for(self, M: int, capacity: int | None = None, K: int | None = None, capturing: bool = False) -> int:
        """record slots per destination: the explicit `capacity`, else the constructor's `capacity_tokens`, else (with
        `routing_groups`) the group-limited estimate -- at `capture_slack` while a graph is being captured -- else the
        token count (the exact worst case)"""
        if capacity is not None:
            cap = capacity
        elif self.capacity_tokens is not None:
            cap = self.capacity_tokens
        elif self.routing_groups is not None and self.ep > 1:
            n_group, topk_group = self.routing_groups
            rpt = ranks_per_token(self.ep, n_group, topk_group, K)
            slack = self.capture_slack if capturing else self.capacity_slack
            cap = min(M, -(-int(M * rpt * slack) // self.ep))
            return max(cap, 1)
        else:
            cap = M
        if M > cap and self.routing_groups is None:
            raise ValueError(f"{M} tokens exceed the group-wide record capacity {cap}")
        return max(cap, 1)

    def _buffers(self, M: int, K: int, cap: int, act_dtype: torch.dtype, ret_dtype: torch.dtype, dev):
        """views of the shared exchange pool for this step's shape (a captured graph replays on these addresses: the
        pool never moves once handed out, and a step's views are prefixes of it)"""
        rowb = self.kernels.ep_row_bytes(self.H, K)
        n = self.ep * cap
        rsz = torch.empty((), dtype=ret_dtype).element_size()
        send = _pool(dev, self.group, "send", n * rowb, self.pool_tag)[: n * rowb].view(self.ep, cap, rowb)
        recv = _pool(dev, self.group, "recv", n * rowb, self.pool_tag)[: n * rowb].view(self.ep, cap, rowb)
        back = _pool(dev, self.group, "back", n * self.H * rsz, self.pool_tag)[: n * self.H * rsz].view(ret_dtype).view(self.ep, cap, self.H)
        slot_of = _pool(dev, self.group, "slot_of", self.ep * M * 4, self.pool_tag)[: self.ep * M * 4].view(torch.int32).view(self.ep, M)
        ov = self._overflow_bufs.get(str(dev))
        if ov is None:
            ov = self._overflow_bufs[str(dev)] = torch.zeros((1,), dtype=torch.int32, device=dev)
        return dict(rowb=rowb, send=send, recv=recv, slot_of=slot_of, overflow=ov, back=back)

    def dispatch_fixed(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor, capacity: int | None = None,
                       return_handle: bool = False):
        """tokens -> owners, ONE collective: (rows [ep*cap, H] act dtype, ids int32 [ep*cap, K], weights fp32
        [ep*cap, K]) -- row-strided views into the receive buffer; unused record slots carry ids -1.  With
        return_handle the step's state comes back as a fourth value to be passed to combine_fixed (two micro-batches
        in flight: each keeps its own handle; their buffers must then be distinct -- see `_buffers`)."""
        M, K = ids.shape
        self._ensure_uniform(M, capacity, hidden)
        cap = self.capacity_for(M, capacity, K)
        ret = self.return_dtype or hidden.dtype
        pkey = _pool_key(hidden.device, self.group, self.pool_tag)
        if pkey in _IN_FLIGHT:
            raise RuntimeError(f"exchange pool {self.pool_tag!r} already holds a dispatch that was not combined "
                               f"({_IN_FLIGHT[pkey]}); two micro-batches in flight need instances with different pool_tag")
        _IN_FLIGHT[pkey] = f"{M} tokens, capacity {cap}"
        try:
            b = self._buffers(M, K, cap, hidden.dtype, ret, hidden.device)
            b["pool_key"] = pkey
            self.kernels.ep_pack_tokens(hidden, tw, ids, self.E, self.ep, cap, b["send"], b["slot_of"], b["overflow"],
                                        self.global_ids)
            self.transport(b["recv"], b["send"])
        except BaseException:
            # the pool is shared by every layer's instance on this (device, group, tag): a failed dispatch (an OOM while
            # the pool grows, a transport error) must not leave it marked busy for all of them
            _IN_FLIGHT.pop(pkey, None)
            raise
        rec = b["recv"].view(self.ep * cap, b["rowb"])
        H2 = self.H * 2
        rows = rec[:, :H2].view(hidden.dtype)
        rids = rec[:, H2:H2 + 4 * K].view(torch.int32)
        rws = rec[:, H2 + 4 * K:H2 + 8 * K].view(torch.float32)
        handle = (b, M, K, cap)
        self._last = handle
        return (rows, rids, rws, handle) if return_handle else (rows, rids, rws)

    def combine_fixed(self, y: torch.Tensor, M: int, out_dtype: torch.dtype = torch.float32,
                      out: torch.Tensor | None = None, handle=None) -> torch.Tensor:
        """owners -> tokens, ONE collective: y [ep*cap, H] (the owners' weighted partial rows, return dtype) ->
        [M, H] out_dtype = fixed-order fp32 sum over the ranks a token visited.  `handle`: what dispatch_fixed
        returned for THIS step (default: the most recent dispatch)."""
        b, M_, K, cap = handle if handle is not None else self._last
        if M_ != M or y.shape != (self.ep * cap, self.H) or y.dtype != b["back"].dtype:
            raise RuntimeError(f"combine_fixed for {M} tokens / rows {tuple(y.shape)} {y.dtype} does not match its "
                               f"dispatch ({M_} tokens, capacity {cap}, {b['back'].dtype})")
        _IN_FLIGHT.pop(b.get("pool_key"), None)
        self.transport(b["back"], y.view(self.ep, cap, self.H))
        if out is None:
            out = torch.empty((M, self.H), dtype=out_dtype, device=y.device)
        return self.kernels.ep_combine(b["back"], b["slot_of"], out)

    def _local(self, rows, rids, rws, ret):
        if self._lc_takes_den:
            return self.local_compute(rows, rids, rws, ret, valid_den=self.ep)
        return self.local_compute(rows, rids, rws, ret)

    def abandon_dispatch(self, handle=None) -> None:
        """give up a dispatch without combining it (error paths, the overflow re-run): frees its pool for the next one"""
        b = (handle if handle is not None else self._last)[0]
        _IN_FLIGHT.pop(b.get("pool_key"), None)

    def _forward_a2a_fixed(self, hidden, tw, ids, cap: int, out_dtype: torch.dtype, out=None, trimmed: bool = False) -> torch.Tensor:
        """trimmed: this step's capacity may be below what a destination can receive.  The flag is formed from
        group-agreed values only (forward()), so either every rank looks at the overflow counters or none does."""
        M, K = ids.shape
        rows, rids, rws, h = self.dispatch_fixed(hidden, tw, ids, cap, return_handle=True)
        if self.check_overflow and trimmed and not _capturing(hidden):
            over, m_max = self._any_rank_overflowed(hidden.device, M)
            if over:
                # a destination ran out of record slots on some rank: every rank repeats the step at the exact bound of
                # the LARGEST token count in the group (a rank-local M would post mismatched exchanges)
                self.abandon_dispatch(h)
                return self._forward_a2a_fixed(hidden, tw, ids, max(m_max, 1), out_dtype, out, trimmed=False)
        ret = self.return_dtype or hidden.dtype
        try:
            y = self._local(rows, rids, rws, ret)
        except BaseException:
            self.abandon_dispatch(h)
            raise
        return self.combine_fixed(y, M, out_dtype, out, handle=h)

    def _any_rank_overflowed(self, dev, M: int = 0):
        """the overflow decision is collective (a rank that re-ran alone would post mismatched exchanges): the MAX over
        ranks of (`dropped since the last look`, token count), one 8-byte all-reduce and one host read.  Eager steps
        only.  -> (overflowed anywhere, largest token count of the group)"""
        ov = self._overflow_bufs[str(dev)]
        delta = (ov - self._overflow_seen).to(torch.int32).reshape(1)
        self._overflow_seen = ov.clone()
        pair = torch.cat([delta, torch.tensor([M], dtype=torch.int32, device=delta.device)])
        if self.ep > 1 and dist.is_initialized():
            dist.all_reduce(pair, op=dist.ReduceOp.MAX, group=self.group)
        vals = pair.tolist()
        return bool(vals[0] > 0), int(vals[1])

    def overflow_count(self) -> int:
        """tokens dropped so far for lack of record capacity (only possible with a capacity below the token
        count); reading it synchronises the host."""
        return int(sum(int(b.item()) for b in self._overflow_bufs.values()))

    def wire_bytes(self, M: int, K: int, capacity: int | None = None, act_bytes: int = 2,
                   ret_bytes: int | None = None, capturing: bool = False) -> dict:
        """bytes one rank puts on xGMI per fixed-path step (the blocks for the other ep-1 ranks) next to the
        routed-row bytes of a slot-granular exchange of the same step"""
        cap = self.capacity_for(M, capacity, K, capturing)
        rowb = self.kernels.ep_row_bytes(self.H, K)
        rb = act_bytes if ret_bytes is None else ret_bytes
        out_b, back_b = (self.ep - 1) * cap * rowb, (self.ep - 1) * cap * self.H * rb
        routed = M * K * self.H * act_bytes
        return {"dispatch_bytes": out_b, "return_bytes": back_b, "routed_row_bytes": routed,
                "dispatch_over_routed": round(out_b / max(routed, 1), 3),
                "return_over_routed": round(back_b / max(M * K * self.H * rb, 1), 3),
                "collectives_per_step": 2, "capacity_tokens": cap}

    # -------------------------------------------------------------------------------- a2a, ragged
    def _forward_a2a(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        M, K = ids.shape
        dev = hidden.device
        flat_ids = ids.reshape(-1)
        owner = owner_of(flat_ids, self.E, self.ep)                       # [M*K], -1 = dropped
        # stable order by destination rank; dropped slots last
        key = torch.where(owner < 0, torch.full_like(owner, self.ep), owner)
        order = torch.argsort(key, stable=True)
        send_counts = torch.bincount(key, minlength=self.ep + 1)[: self.ep]
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()              # host sync: split sizes
        n_send, n_recv = sum(sc), sum(rc)
        order = order[:n_send]
        tok = torch.div(order, K, rounding_mode="floor")
        x_send = hidden.index_select(0, tok)
        first = torch.tensor(self.first_expert, device=dev, dtype=torch.int64)
        lid_send = (flat_ids[order].to(torch.int64) - first[owner[order]]).to(torch.int32)
        w_send = tw.reshape(-1)[order].contiguous()
        x_recv = torch.empty((n_recv, self.H), dtype=hidden.dtype, device=dev)
        lid_recv = torch.empty((n_recv,), dtype=torch.int32, device=dev)
        w_recv = torch.empty((n_recv,), dtype=torch.float32, device=dev)
        dist.all_to_all_single(x_recv, x_send, rc, sc, group=self.group)
        dist.all_to_all_single(lid_recv, lid_send, rc, sc, group=self.group)
        dist.all_to_all_single(w_recv, w_send, rc, sc, group=self.group)
        # local experts: each received row is one (token, slot) pair => top_k = 1
        y_recv = self.local_compute(x_recv, lid_recv.view(-1, 1), w_recv.view(-1, 1), torch.float32)
        y_back = torch.empty((n_send, self.H), dtype=torch.float32, device=dev)
        dist.all_to_all_single(y_back, y_recv.contiguous(), sc, rc, group=self.group)
        out = torch.zeros((M, self.H), dtype=torch.float32, device=dev)
        out.index_add_(0, tok, y_back)
        return out

    # -------------------------------------------------------------------------------- ar
    def _forward_ar(self, hidden: torch.Tensor, tw: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
        """all ranks see all tokens: ONE all-gather of token records [activations | ids | weights] (the same record the
        a2a path ships), local experts only, ONE reduce-scatter of the [ep*M, H] partials in the return dtype (the
        activation dtype by default -- what the reference all-reduces, moe_runner.py:494; return_dtype=float32 keeps the
        single-rank fp32 sum)"""
        M, K = ids.shape
        dev = hidden.device
        ep = self.ep
        H2 = self.H * 2
        rowb = H2 + 8 * K
        rec = torch.empty((M, rowb), dtype=torch.uint8, device=dev)
        rec[:, :H2].view(hidden.dtype).copy_(hidden)
        rec[:, H2:H2 + 4 * K].view(torch.int32).copy_(ids)
        rec[:, H2 + 4 * K:].view(torch.float32).copy_(tw)
        allrec = torch.empty((ep * M, rowb), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(allrec, rec, group=self.group)
        xs = allrec[:, :H2].view(hidden.dtype)
        ii = allrec[:, H2:H2 + 4 * K].view(torch.int32)
        ww = allrec[:, H2 + 4 * K:].view(torch.float32)
        if self.expert_map.device != dev:               # once, outside any graph capture (the warm-up step)
            self.expert_map = self.expert_map.to(dev)
        emap = self.expert_map
        local = torch.where(ii < 0, torch.full_like(ii, -1),
                            emap[ii.clamp(0, self.E - 1).to(torch.int64)])    # routed_experts.py:1332-1342
        ret = self.return_dtype or hidden.dtype
        part = self._local(xs, local.contiguous(), ww, ret)                    # [ep*M, H]
        out = torch.empty((M, self.H), dtype=ret, device=dev)
        dist.reduce_scatter_tensor(out, part.contiguous(), group=self.group)
        return out

    def _ensure_uniform(self, M: int, capacity, like: torch.Tensor) -> None:
        """no group-wide capacity was named: the rank-local token count sizes the collectives -- directly, or through the
        group-limited estimate of `routing_groups` -- so ranks with different counts would post mismatched
        exchanges (a hang or silent corruption).  Checked ONCE per new token count, outside any capture (one small
        all-gather): fail loudly instead."""
        if self.ep > 1 and capacity is None and self.capacity_tokens is None \
                and not self.validate_uniform and not _capturing(like) and M not in self._uniform_checked:
            self._check_uniform(M, "token count (no common capacity was given)")
            self._uniform_checked.add(M)

    def _check_uniform(self, value: int, what: str) -> None:
        t = torch.tensor([value], dtype=torch.int64)
        if dist.is_initialized() and dist.get_backend(self.group) != "gloo":
            t = t.cuda()
        got = [torch.empty_like(t) for _ in range(self.ep)]
        dist.all_gather(got, t, group=self.group)
        vals = [int(g.item()) for g in got]
        if len(set(vals)) != 1:
            raise RuntimeError(f"expert-parallel step with different {what} across ranks: {vals}; pad the ranks to a "
                               f"common count or pass a common capacity (see the module docstring)")

    def forward(self, hidden: torch.Tensor, topk_weights: torch.Tensor, topk_ids: torch.Tensor,
                force_collectives: bool = False, capacity: int | None = None,
                out_dtype: torch.dtype = torch.float32, out: torch.Tensor | None = None) -> torch.Tensor:
        """hidden [M,H] (this rank's tokens), GLOBAL expert ids int32 [M,K] -> [M,H] (fp32 by default; `out`, a
        contiguous [M,H] tensor, receives it in place on the fixed path).
        force_collectives runs the collective data path even on a single rank (plumbing checks).
        capacity: see the module docstring (the token count every rank of the group agrees on)."""
        if out is not None:
            out_dtype = out.dtype
        self._ensure_uniform(topk_ids.size(0), capacity, hidden)
        if self.ep == 1 and not force_collectives:
            y = self.local_compute(hidden, topk_ids, topk_weights, out_dtype)
            return y if out is None else out.copy_(y)
        M = topk_ids.size(0)
        if self.mode == "a2a":
            cap = self.capacity_for(M, capacity, topk_ids.size(1), capturing=_capturing(hidden))
            if self.validate_uniform:
                self._check_uniform(cap, "record capacity")
            if cap <= self.fixed_max_tokens:
                # may a destination receive more records than it has slots?  From group-agreed values only: a capacity
                # derived from M is below the worst case iff cap < M (M is then uniform, _ensure_uniform); a NAMED common
                # capacity under group-limited routing may be below some rank's token count, which this rank cannot know
                named = capacity is not None or self.capacity_tokens is not None
                trimmed = (self.routing_groups is not None) if named else cap < M
                return self._forward_a2a_fixed(hidden, topk_weights, topk_ids, cap, out_dtype, out, trimmed=trimmed)
            y = self._forward_a2a(hidden, topk_weights, topk_ids)
        else:
            if self.validate_uniform:
                self._check_uniform(M, "token count")
            y = self._forward_ar(hidden, topk_weights, topk_ids)
        if out is not None:
            return out.copy_(y)
        return y if y.dtype == out_dtype else y.to(out_dtype)


def forward_two_microbatches(ep0: ExpertParallelExperts, ep1: ExpertParallelExperts, batch0, batch1, *,
                             out_dtype: torch.dtype = torch.float32, shared: Callable | None = None,
                             comm_stream=None):
    """Two micro-batches through the fixed-capacity exchange with the RETURN exchange of batch 0 under the expert GEMMs of
    batch 1 (SURVEY 7 "overlap"; the reference overlaps its shared experts with the exchange the same way,
    runner/shared_experts.py:45-80, and its DBO hooks overlap two micro-batches, modular_kernel.py:1219-1276).

        main stream : pack0 a2a0 | pack1 a2a1 | GEMMs(0) ............ | GEMMs(1) ......... | sum(1)
        comm stream :                                                   a2a-back(0) sum(0)   [shared experts]   a2a-back(1)
    ep0 / ep1: two ExpertParallelExperts on the same group with DIFFERENT `pool_tag`s (each micro-batch owns its exchange
    buffers while in flight).  batch = (hidden [M,H], topk_weights [M,K], topk_ids [M,K]).  shared(hidden) -> tensor: an
    optional always-on expert (or any independent work) launched on the communicator stream between the two return
    exchanges.  On a CPU group (gloo tests) there are no streams: the same calls run in the same order, which is what
    makes the result independent of the overlap.  -> (out0, out1[, shared0, shared1])"""
    if ep0.pool_tag == ep1.pool_tag:
        raise ValueError("forward_two_microbatches needs two instances with different pool_tag (each micro-batch owns its "
                         "exchange buffers while in flight)")
    x0, w0, i0 = batch0
    x1, w1, i1 = batch1
    cuda = x0.is_cuda
    M0, M1 = i0.size(0), i1.size(0)
    ep0._ensure_uniform(M0, None, x0)
    ep1._ensure_uniform(M1, None, x1)
    capturing = _capturing(x0)
    cap0 = ep0.capacity_for(M0, None, i0.size(1), capturing=capturing)
    cap1 = ep1.capacity_for(M1, None, i1.size(1), capturing=capturing)

    def dispatch_both(c0, c1):
        a = ep0.dispatch_fixed(x0, w0, i0, c0, return_handle=True)
        try:
            b = ep1.dispatch_fixed(x1, w1, i1, c1, return_handle=True)
        except BaseException:
            ep0.abandon_dispatch(a[3])
            raise
        return a, b
    r0, r1 = dispatch_both(cap0, cap1)
    # A capacity below the token count (group-limited estimate) may have dropped records.  Same rule as forward(): an eager
    # step with check_overflow looks at the counters -- collectively, both instances in the same order on every rank --
    # and repeats BOTH dispatches at the exact bound of the largest token count in the group; a step being captured can
    # only count (its caller reads overflow_count() after the replay).  The flags come from group-agreed values only
    # (capacity vs the uniform token count), so every rank takes the same branch.
    t0, t1 = ep0.check_overflow and cap0 < M0, ep1.check_overflow and cap1 < M1
    if (t0 or t1) and not capturing:
        over0, m0 = ep0._any_rank_overflowed(x0.device, M0) if t0 else (False, M0)
        over1, m1 = ep1._any_rank_overflowed(x1.device, M1) if t1 else (False, M1)
        if over0 or over1:
            ep0.abandon_dispatch(r0[3])
            ep1.abandon_dispatch(r1[3])
            r0, r1 = dispatch_both(max(m0, 1) if t0 else cap0, max(m1, 1) if t1 else cap1)
    ret0, ret1 = ep0.return_dtype or x0.dtype, ep1.return_dtype or x1.dtype
    try:
        y0 = ep0._local(r0[0], r0[1], r0[2], ret0)
        if cuda:
            side = comm_stream if comm_stream is not None else torch.cuda.Stream(device=x0.device)
            main = torch.cuda.current_stream(x0.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                out0 = ep0.combine_fixed(y0, M0, out_dtype, handle=r0[3])
                sh = (shared(x0), shared(x1)) if shared is not None else None
            y0.record_stream(side)
            y1 = ep1._local(r1[0], r1[1], r1[2], ret1)          # overlaps the return exchange of batch 0
            out1 = ep1.combine_fixed(y1, M1, out_dtype, handle=r1[3])
            main.wait_stream(side)
            # allocated under the side stream, consumed by the caller on the main stream: tell the caching allocator, or
            # it may hand the blocks to the next side-stream allocation while main still reads them
            out0.record_stream(main)
            for t in sh or ():
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)
        else:
            out0 = ep0.combine_fixed(y0, M0, out_dtype, handle=r0[3])
            sh = (shared(x0), shared(x1)) if shared is not None else None
            y1 = ep1._local(r1[0], r1[1], r1[2], ret1)
            out1 = ep1.combine_fixed(y1, M1, out_dtype, handle=r1[3])
    except BaseException:
        # (whatever raised -- the experts, the shared callable, a transport -- both pools are free for the next step)
        ep0.abandon_dispatch(r0[3])
        ep1.abandon_dispatch(r1[3])
        raise
    return (out0, out1) if sh is None else (out0, out1, sh[0], sh[1])


def agree_tuned_plans(engines, group: dist.ProcessGroup | None = None) -> int:
    """First-call autotune under expert parallelism (round-4 verdict item 8).  Each rank's engine timed the candidate plans
    of its step shapes by itself (`set_tuning(autotune=2)`, the value an expert-parallel engine accepts), so two ranks may
    remember different winners = different fp32 summation orders inside one group.  This makes them agree: for every engine
    (same order on every rank) the remembered (shape key, candidate index) pairs are compared across the group and every
    rank takes the MIN index per key -- the candidate list is a function of the engine configuration and the step shape
    only, so index i names the same plan everywhere.  Call it after the warm-up steps and BEFORE capturing a graph;
    ranks must have run the same step shapes (checked: differing key sets raise).  Works on any backend (the gloo tests
    drive it with a stand-in engine exposing tuned_plans() / set_tuned_plan()).  -> number of plans that changed here."""
    engines = list(engines)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    changed = 0
    for n, eng in enumerate(engines):
        plans = sorted(eng.tuned_plans())
        keys = [k for k, _ in plans]
        if world > 1:
            got = [None] * world
            dist.all_gather_object(got, keys, group=group)
            if any(g != keys for g in got):
                raise RuntimeError(f"agree_tuned_plans: engine {n} remembers plans for different step shapes across ranks "
                                   f"({[len(g) for g in got]} keys): run the same warm-up shapes on every rank")
        if not plans:
            continue
        idx = torch.tensor([i for _, i in plans], dtype=torch.int64)
        if world > 1:
            on_gpu = dist.get_backend(group) != "gloo"
            t = idx.cuda() if on_gpu else idx
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            idx = t.cpu()
        for (key, mine), agreed in zip(plans, idx.tolist()):
            if agreed != mine:
                eng.set_tuned_plan(key, int(agreed))
                changed += 1
    return changed
