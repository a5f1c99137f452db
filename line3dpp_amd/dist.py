"""Pair-sharded matchImages across the GPUs of one node (one process per GPU).

Phase A (Line3D::matchingCPU per directed view pair, line3D.cc:728-735) reads only static per-view
arrays, so the directed pairs are independent units: every rank holds all views (<= 34 MB even for
the largest BASELINE config), matches a contiguous, cost-balanced range of the pair list, and the
ranks then exchange their slices of the fixed-layout slot buffer with an all-gather over RCCL/xGMI -- in
compact form: only the uint32 target index of every slot travels (4 B instead of the 32-byte l3d_slot
record), the receiving rank re-derives overlap and depths with the match kernel's own device functions
(l3d_pack_slot_indices / l3d_expand_slot_indices; L3D_EXCHANGE_FULL=1 sends the full records instead; the
keep-all mode kNN <= 0 has no fixed slot layout and runs unsharded on every rank).  Phase B (the per-view part, line3D.cc:745-773): its dense part -- the pass over every 2D
segment's hypothesis list that finds the supporting pairs of scoringCPU -- is sharded by views as well
(l3d_lists_shard); the records it produces (edges / headers, a few MB) are all-gathered and the sparse remainder (the
chain of inverse matches in ascending camID, scores, filterMatches) runs on every rank (SURVEY.md §8e option 2 for
the part that carries the work, option 1 for the order-dependent rest).

Default since round 3 -- the HALO form (match_images_halo): phase B's list pass is sharded by views, so a rank needs
the slots of the pairs that TOUCH ITS VIEWS only.  The views are cut into contiguous ranges whose outgoing pairs carry
equal shares of the matching cost (l3d_plan_shards); a rank matches the pairs whose source view it owns and sends, point
to point, the compact indices of the pairs whose target view another rank owns (for a ring neighbourhood: the few pairs
across a range boundary, ~2/N of what the all-gather moved).  Those pairs are matched first and travel while the rank
matches the rest.  The record slabs of the list pass are all-gathered in place and the tail of phase B runs on the
records alone (l3d_lists.h HypHdr carries what it needs of a slot).  L3D_DIST_ALLGATHER=1 selects the round-2 form below.

The all-gather form: the exchange is written as one in-place broadcast per owning rank: slices are uneven, and
`ncclBroadcast` of a slice of the one shared buffer is the all-gather(v) primitive RCCL offers.
Works with backend "nccl" (= RCCL, device buffers) and "gloo" (CPU tensors, used by the tests).
"""
import os

import numpy as np


def pair_ranges(costs, world_size):
    """Split pairs [0, n) into `world_size` contiguous ranges of roughly equal total cost.

    costs[i] = Ms*Mt of pair i.  Returns [(first, count)] * world_size (count may be 0)."""
    costs = np.asarray(costs, np.float64)
    n = len(costs)
    if n == 0:
        return [(0, 0)] * world_size
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        b = int(np.searchsorted(cum, target, side="left"))
        # pick the boundary closest to the target
        if b > 0 and abs(cum[b - 1] - target) <= abs(cum[min(b, n)] - target):
            b -= 1
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1] - bounds[r]) for r in range(world_size)]


def slot_byte_ranges(ranges, slot_offsets, n_slots, slot_bytes=32):
    """byte range of the slot buffer owned by each rank, given per-pair slot offsets"""
    off = list(map(int, slot_offsets)) + [int(n_slots)]
    return [(off[f] * slot_bytes, off[f + c] * slot_bytes) for f, c in ranges]


def exchange_slots(buf, byte_ranges, group=None):
    """In-place all-gather(v): after the call every rank's `buf` (flat uint8 torch tensor) holds every
    rank's slice.  byte_ranges[r] = (lo, hi) owned by rank r.

    Equal, contiguous slices (the usual case: BASELINE configs have uniform pairs) go through ONE
    all_gather_into_tensor -- a ring all-gather drives all xGMI links at once, whereas N successive
    broadcasts serialise them.  Uneven slices take one broadcast per owning rank.  Which of the two runs is a
    function of byte_ranges alone -- identical on every rank, so the ranks can never issue different collectives --
    and an error of the chosen collective propagates to the caller (RCCL and gloo both implement
    all_gather_into_tensor)."""
    import torch.distributed as dist
    world = len(byte_ranges)
    rank = dist.get_rank(group)
    sizes = [hi - lo for lo, hi in byte_ranges]
    contiguous = all(byte_ranges[r][1] == byte_ranges[r + 1][0] for r in range(world - 1))
    if world > 1 and contiguous and len(set(sizes)) == 1 and sizes[0] > 0:
        lo0, hi_last = byte_ranges[0][0], byte_ranges[-1][1]
        lo, hi = byte_ranges[rank]
        # the send side is a copy of the rank's slice (1/N of a few MB), so that input and output never alias
        dist.all_gather_into_tensor(buf[lo0:hi_last], buf[lo:hi].clone(), group=group)
        return "all_gather"
    for r, (lo, hi) in enumerate(byte_ranges):
        if hi > lo:
            dist.broadcast(buf[lo:hi], src=r if group is None else dist.get_global_rank(group, r), group=group)
    return "broadcast"


class _DevMem:
    """raw device pointer -> torch tensor without a copy (__cuda_array_interface__, HIP/ROCm torch)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_tensor(ptr, nbytes, device):
    import torch
    return torch.as_tensor(_DevMem(ptr, nbytes), device=device)


def _issue_stream(buf, device):
    """torch's current stream at the moment a collective is ISSUED (the backend orders the collective behind what that stream
    holds then): handed to _wait_for_exchange, so that a caller who changes the stream context between issue and wait cannot
    make the shortcut look valid"""
    if buf is None or not getattr(buf, "is_cuda", False):
        return None
    import torch
    return int(torch.cuda.current_stream(device).cuda_stream)


def _wait_for_exchange(buf, device, l3d=None, group=None, issued_on=None):
    """The collective runs on the backend's own stream; the library's stream must not touch `buf` before it is done.
    torch's NCCL (= RCCL) backend orders that by itself when the library launches on torch's CURRENT stream: a collective
    starts after what that stream holds at the call, and Work.wait() (called by the synchronous collectives and by the
    loops over batch_isend_irecv here) makes that stream -- not the host -- wait for it.  line3dpp_amd.Line3D runs on the
    stream it was created with (default 0 = torch's default stream), so in the usual set-up nothing is left to do and
    the host goes on enqueuing (round 3 synchronised the device at every phase boundary: five host round trips per
    call).  Only a context on ANOTHER stream needs the device-wide wait -- and any backend other than NCCL / RCCL, whose
    Work.wait() promises nothing about streams: the shortcut is taken for backend "nccl" only."""
    if buf.is_cuda:
        import torch
        import torch.distributed as dist
        now = int(torch.cuda.current_stream(device).cuda_stream)
        ctx_stream = int(getattr(l3d, "stream", -1)) if l3d is not None else -1
        # (the context's stream must be the stream the collective was issued on AND still torch's current one)
        same = l3d is not None and ctx_stream == now and (issued_on is None or issued_on == now)
        if not (same and dist.get_backend(group) == "nccl"):
            torch.cuda.synchronize(device)


def _all_ok(ok, device, group, level=1):
    """Every rank learns whether ANY rank failed locally (a HIP error in matchPairs / expandSlotIndices / listsShardViews,
    an allocation): a rank that left the call on its own would leave its peers inside a collective for ever.  A failing
    rank therefore keeps taking part in the communication pattern (with whatever its buffers hold) up to the next status
    exchange, where all ranks give up together.  One int32 MIN all-reduce, placed where the host waits anyway.
    Round 6: every exchange is a host round trip (an 8-GPU C1 call is 0.5 ms), so only the one in front of the record gather
    -- the point a failing rank reaches without the buffers the gather needs -- is on by default (level 1); the status of
    the tail travels with its counts (_gather_counts), and the exchanges that only guarded against a failure on ONE rank of a
    step whose outcome is a function of state every rank holds alike (layout, commit, the affinity's begin) are level 2:
    L3D_DIST_STATUS=2 turns them back on, =0 turns all of them off."""
    want = int(os.environ.get("L3D_DIST_STATUS", "1"))
    if want < level:
        return ok
    import torch
    import torch.distributed as dist
    on_gpu = device is not None and dist.get_backend(group) == "nccl"
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


def shard_needs(pairs, M, view_bounds, world_size):
    """needs[r] = the ranks (other than r) whose RECORDS rank r's chain depends on.  View v depends on view u < v when a pair
    (u -> v) exists (its matches are handed to v as inverse hypotheses, line3D.cc:1680) and, through u's own inverse
    hypotheses, on everything u depends on: the ancestors of r's views in that DAG, by owner.  A function of the pair list
    alone.  N rings without pairs between them (weak scaling): every set is empty -- no record travels, every rank's chain
    covers its own records only."""
    cams = sorted(M)
    vidx = {c: i for i, c in enumerate(cams)}
    V = len(cams)
    preds = [[] for _ in range(V)]
    for s, t in pairs:
        u, v = vidx[int(s)], vidx[int(t)]
        if v > u:
            preds[v].append(u)
    owner = np.searchsorted(np.asarray(view_bounds)[1:], np.arange(V), side="right")
    needs = []
    for r in range(world_size):
        seen = np.zeros(V, bool)
        stack = list(range(int(view_bounds[r]), int(view_bounds[r + 1])))
        for v in stack:
            seen[v] = True
        while stack:
            v = stack.pop()
            for u in preds[v]:
                if not seen[u]:
                    seen[u] = True
                    stack.append(u)
        needs.append(sorted(set(int(q) for q in owner[seen]) - {r}))
    return needs


def gather_slabs(slabs, rank, world_size, device, group=None, l3d=None, needs=None):
    """Every rank's slab of every record array of the sharded list pass (Line3D.listsShard) lands at its place in every
    rank's array.  slabs = [(slab pointer, slab bytes, full-array pointer)]; equal slab sizes by construction.

    needs (round 6, shard_needs): the three RECORD arrays (edges, headers, segment headers) only travel from a rank to the
    ranks whose chain depends on its records; the fourth array -- the pool counters with every rank's overflow flags, on
    which all ranks decide alike -- always reaches everyone.  None: everything to everyone.

    Default: DIRECT exchange -- one send and one receive per peer, all posted at once (batch_isend_irecv).  The xGMI
    fabric of an 8-GPU node is a full mesh (7 links per GPU, one per peer), so the seven transfers of a rank run on seven
    links side by side and a link carries one slab; a ring all-gather moves (N-1) slabs over ONE link per rank, N-1
    times as long.  L3D_GATHER_COLLECTIVE=1 uses all_gather_into_tensor instead (RCCL's choice of algorithm), for
    whoever has a node to compare them on."""
    import torch.distributed as dist
    collective = bool(os.environ.get("L3D_GATHER_COLLECTIVE"))
    issued_on = _issue_stream(device_tensor(slabs[0][2], 1, device), device) if slabs else None
    ops = []
    for i, (sp, sb, fp) in enumerate(slabs):
        if not sb:
            continue
        full = device_tensor(fp, sb * world_size, device)
        mine = full[rank * sb:(rank + 1) * sb]
        if (collective and needs is None) or world_size == 1:
            # RCCL gathers in place when the input is the rank's own slice of the output (no copy of the slab on the
            # send side); the CPU backend of the tests gets a copy
            dist.all_gather_into_tensor(full, mine if full.is_cuda else mine.clone(), group=group)
            continue
        glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        # (P2POp tags are IGNORED by the NCCL / RCCL backend: messages between two ranks are matched in the order in
        # which both sides post them.  Every rank walks the arrays i and the peers q in the same order and posts, per
        # (array, peer), its receive then its send, so the k-th transfer from a to b is array k on both sides.  The tag
        # is what the gloo backend of the CPU tests matches by.)
        everyone = needs is None or i == len(slabs) - 1           # (the counter slab is the last array)
        for q in range(world_size):       # disjoint slices of one array: sends read `mine`, receives fill the others
            if q != rank:
                if everyone or q in needs[rank]:
                    ops.append(dist.P2POp(dist.irecv, full[q * sb:(q + 1) * sb], glob(q), group, tag=i))   # tag = array
                if everyone or rank in needs[q]:
                    ops.append(dist.P2POp(dist.isend, mine, glob(q), group, tag=i))
    if ops:
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    if slabs:
        _wait_for_exchange(device_tensor(slabs[0][2], 1, device), device, l3d, group, issued_on)


def exchange_parts(layout, rank, world_size, device, group=None, l3d=None):
    """The outputs of a sharded tail (Line3D.tailShardLayout): rank r's part of every array goes to all other ranks, in
    place -- variable sizes, known to every rank from the all-gathered counts; one send and one receive per peer and
    array in a single group, like the record slabs (one transfer per xGMI link and array).  Empty parts are skipped on
    both sides.  Order: arrays ascending, peers ascending, on both sides (NCCL / RCCL match the messages of a pair of ranks
    in posting order; the tag is what gloo matches by)."""
    import torch.distributed as dist
    glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
    recvs, sends, last = [], [], None
    issued_on = None
    for k, (ptr, elt, parts) in enumerate(layout):
        total = max(f + n for f, n in parts) * elt
        if total == 0:
            continue
        full = device_tensor(ptr, total, device)
        if issued_on is None:
            issued_on = _issue_stream(full, device)
        f_me, n_me = parts[rank]
        for q in range(world_size):
            if q == rank:
                continue
            f_q, n_q = parts[q]
            if n_q:
                recvs.append(dist.P2POp(dist.irecv, full[f_q * elt:(f_q + n_q) * elt], glob(q), group, tag=100 + k))
            if n_me:
                sends.append(dist.P2POp(dist.isend, full[f_me * elt:(f_me + n_me) * elt], glob(q), group, tag=100 + k))
        last = full
    if recvs or sends:
        for r in dist.batch_isend_irecv(recvs + sends):
            r.wait()
        _wait_for_exchange(last, device, l3d, group, issued_on)


def _gather_counts(n_r, h_r, rc, world_size, device, group):
    """the two counts of every rank's part of a sharded tail AND the status of its count step (round 6: one exchange where
    round 5 had a status all-reduce followed by the count all-gather) -> ([(n, h)] x world, [rc] x world)"""
    import torch
    import torch.distributed as dist
    on_gpu = device is not None and dist.get_backend(group) == "nccl"
    dev = device if on_gpu else "cpu"
    mine = torch.tensor([n_r, h_r, rc], dtype=torch.int64, device=dev)
    out = torch.zeros(3 * world_size, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine, group=group)
    host = out.cpu().tolist()                    # ONE read of everything
    return [(int(host[3 * r]), int(host[3 * r + 1])) for r in range(world_size)], [int(host[3 * r + 2]) for r in range(world_size)]


def plan_halo(pairs, M, world_size):
    """The partition of a call over `world_size` ranks (l3d_plan_shards) and what follows from it for every rank:
    view_bounds, pair_bounds, and per rank the runs of consecutive pairs it sends (pairs it owns whose TARGET view another
    rank owns): runs[r] = [(peer, first pair, pair count)].  A function of the pair list alone: identical on every rank."""
    import ctypes as C
    from . import _lib
    cams = sorted(M)
    vidx = {c: i for i, c in enumerate(cams)}
    src = np.ascontiguousarray([vidx[int(s)] for s, _ in pairs], np.uint32)
    tgt = np.asarray([vidx[int(t)] for _, t in pairs], np.uint32)
    cost = np.ascontiguousarray([M[int(s)] * M[int(t)] for s, t in pairs], np.uint64)
    vb = np.zeros(world_size + 1, np.uint32); pb = np.zeros(world_size + 1, np.uint32)
    rc = _lib.load().l3d_plan_shards(len(cams), len(src), _lib.ptr(src), _lib.ptr(cost), world_size, _lib.ptr(vb), _lib.ptr(pb))
    if rc != 0:
        raise RuntimeError("l3d_plan_shards: " + _lib.last_error())
    owner = np.searchsorted(vb[1:], np.arange(len(cams)), side="right")      # rank of every view
    src_rank, tgt_rank = owner[src], owner[tgt]
    runs = [[] for _ in range(world_size)]
    for p in range(len(src)):
        r, q = int(src_rank[p]), int(tgt_rank[p])
        if r == q:
            continue
        if runs[r] and runs[r][-1][0] == q and runs[r][-1][1] + runs[r][-1][2] == p:
            runs[r][-1] = (q, runs[r][-1][1], runs[r][-1][2] + 1)
        else:
            runs[r].append((q, p, 1))
    return dict(view_bounds=vb, pair_bounds=pb, runs=runs, needs=shard_needs(pairs, M, vb, world_size))


def early_ranges(first, count, halo_pairs):
    """Order of a rank's own pair range [first, first + count): the pairs it has to SEND (halo_pairs, ascending) should
    be matched first, the largest stretch without any last (it is matched while the halo travels).  Returns
    ([(first, count)] to match before the exchange starts, (first, count) of the stretch matched during it)."""
    if not count:
        return [], (first, 0)
    if not len(halo_pairs):
        return [], (first, count)
    edges = [first - 1] + [int(p) for p in halo_pairs] + [first + count]
    gaps = [(edges[i + 1] - edges[i] - 1, edges[i] + 1) for i in range(len(edges) - 1)]
    glen, gfirst = max(gaps)
    early = []
    if gfirst > first:
        early.append((first, gfirst - first))
    if gfirst + glen < first + count:
        early.append((gfirst + glen, first + count - gfirst - glen))
    return early, (gfirst, glen)


def match_images_halo(l3d, rank, world_size, device=None, group=None, **params):
    """matchImages over `world_size` ranks in the halo form (module docstring).  Same contract as match_images_sharded:
    every rank ends with the complete result; False after a successful matchBegin means the context was closed with
    matchAbort."""
    import time
    import torch.distributed as dist
    acc = getattr(l3d, "dist_ms", None)
    if acc is None:
        acc = l3d.dist_ms = dict(match=0.0, exchange_slots=0.0, lists=0.0, exchange_lists=0.0, finish=0.0, calls=0)
    acc.setdefault("match_early", 0.0)
    t_last = [time.perf_counter()]

    def lap(key):
        now = time.perf_counter()
        acc[key] += 1e3 * (now - t_last[0])
        t_last[0] = now
    acc["calls"] += 1
    if not l3d.matchBegin(**params):
        return False

    def give_up():
        l3d.matchAbort()
        return False
    pairs, slot_off = l3d.pairs()
    M = l3d._M
    plan = plan_halo(pairs, M, world_size)
    l3d.halo_plan = plan
    vb, pb, runs, needs = plan["view_bounds"], plan["pair_bounds"], plan["runs"], plan["needs"]
    # the chain of this rank covers the records its views depend on (the ranks in needs[rank], all below it), and the
    # library's sharded entries need not wait for the device when the exchanges order themselves behind its stream (RCCL on
    # the context's stream = torch's current one); l3d_shard_options
    if hasattr(l3d, "shardOptions"):
        ordered = False
        # (L3D_DIST_ORDERED=0: the sharded entries wait for the device before they return, whatever the backend)
        if device is not None and getattr(l3d, "stream", None) is not None and dist.get_backend(group) == "nccl" and \
                os.environ.get("L3D_DIST_ORDERED", "1") != "0":
            import torch
            ordered = int(torch.cuda.current_stream(device).cuda_stream) == int(l3d.stream)
        if not l3d.shardOptions(min(needs[rank] + [rank]), ordered):
            return give_up()
    first, count = int(pb[rank]), int(pb[rank + 1] - pb[rank])
    send = runs[rank]
    recv = [(r, f, n) for r in range(world_size) for (q, f, n) in runs[r] if q == rank]   # (source rank, first, count)
    halo_pairs = [p for (_, f, n) in send for p in range(f, f + n)]
    early, late = early_ranges(first, count, halo_pairs)
    # A local failure from here on does NOT leave the call: the rank keeps posting what the plan says (its peers are
    # inside the same exchanges) and all ranks give up together at the next status exchange (_all_ok).
    ok = True
    for f, n in early:
        ok = ok and (not n or l3d.matchPairs(f, n))
    for _, f, n in send:
        ok = ok and l3d.packSlotIndices(f, n)
    lap("match_early")
    # ---- the halo: compact indices of whole pair runs, point to point, at the pairs' own places in the index buffer ----
    reqs = []
    buf = None
    # (the index buffer is asked for on EVERY rank of a call that exchanges anything -- whether a rank has halo pairs is a
    # function of the plan, identical everywhere.  A failing allocation is the one failure that cannot be carried through
    # the exchange; like a failing matchBegin it is a failure of the call's set-up, and the status exchange that guarded
    # against it on ONE rank alone is level 2 since round 6)
    if any(runs):
        ptr, n_slots = l3d.slot_index_buffer()     # (also on a rank that has failed: it keeps posting what the plan says)
        if not _all_ok(ptr is not None, device, group, level=2):
            return give_up()
        if ptr is None:
            return give_up()
    if send or recv:
        buf = device_tensor(ptr, n_slots * 4, device)
        off = [int(o) for o in slot_off] + [int(n_slots)]

        def span(f, n):
            return buf[4 * off[f]:4 * off[f + n]]
        glob = (lambda r: r) if group is None else (lambda r: dist.get_global_rank(group, r))
        # (tags: ignored by NCCL / RCCL, which matches the messages of a pair of ranks in posting order.  Both sides list
        # the runs between them in ascending pair order -- `recv` is built from the senders' `runs`, which are ascending --
        # so the k-th send of rank a to rank b meets the k-th receive b posted for a.  gloo matches by the tag.)
        ops = [dist.P2POp(dist.irecv, span(f, n), glob(r), group, tag=f) for (r, f, n) in recv] + \
              [dist.P2POp(dist.isend, span(f, n), glob(q), group, tag=f) for (q, f, n) in send]
        issued_on = _issue_stream(buf, device)
        reqs = dist.batch_isend_irecv(ops)
    # ---- the rest of this rank's pairs, matched while the halo travels ----
    ok = ok and (not late[1] or l3d.matchPairs(late[0], late[1]))
    lap("match")
    for r in reqs:
        r.wait()
    if reqs:
        _wait_for_exchange(buf, device, l3d, group, issued_on)
    lap("exchange_slots")
    for _, f, n in recv:
        ok = ok and l3d.expandSlotIndices(f, n)
    shard_tail = os.environ.get("L3D_SHARD_TAIL", "1") != "0" and hasattr(l3d, "tailShardCount")
    for _ in range(8):
        slabs = l3d.listsShardViews(rank, world_size, int(vb[rank]), int(vb[rank + 1])) if ok else None
        ok = ok and slabs is not None
        lap("lists")
        if not _all_ok(ok, device, group):
            return give_up()
        gather_slabs(slabs, rank, world_size, device, group, l3d, needs if shard_tail else None)
        lap("exchange_lists")
        # the tail: sharded by views as well (chain on every rank, scores / filterMatches / outputs / medians by the owner
        # of the view, the outputs exchanged in place), or replicated on the records of all ranks (L3D_SHARD_TAIL=0)
        n_r = h_r = 0
        if shard_tail:
            rc, n_r, h_r = l3d.tailShardCount()
        else:
            rc = l3d.L.l3d_match_finish(l3d.h)
        l3d.last_status = rc
        lap("finish")
        # (every rank sees every counter, so rc is the same everywhere unless a rank failed locally: the status keeps a rank
        # whose finish succeeded from leaving while another is about to repeat the exchange.  Sharded tail: it travels WITH
        # the counts -- one exchange, one host read)
        if shard_tail:
            counts, rcs = _gather_counts(n_r, h_r, rc, world_size, device, group)
            if all(x == 0 for x in rcs):
                layout = l3d.tailShardLayout(world_size, counts, [int(v) for v in vb])
                if not _all_ok(layout is not None, device, group, level=2) or layout is None:
                    return give_up()                 # (a rank whose layout failed has closed its call already: a no-op there)
                exchange_parts(layout, rank, world_size, device, group, l3d)
                rc = l3d.tailShardCommit()
                l3d.last_status = rc
                lap("finish")
                if not _all_ok(rc == 0, device, group, level=2):
                    return give_up() if rc == 0 else l3d._check(rc, "tailShardCommit")
                return True if rc == 0 else l3d._check(rc, "tailShardCommit")
            if all(x == -10 for x in rcs):
                continue                             # L3D_ERR_RETRY on every rank: pools enlarged alike, repeat list pass + exchange
            # a rank failed locally (or the ranks disagree, which the shared counters rule out): all give up together
            if rc in (0, -10):
                l3d.matchAbort()
                return False
            return l3d._check(rc, "tailShardCount")
        same = _all_ok(rc in (0, -10), device, group)
        if rc == 0 and same:
            return True
        if not same:
            if rc in (0, -10):
                l3d.matchAbort()           # (rc == 0: the call is closed already -- a no-op; the results are discarded)
                return False
            return l3d._check(rc, "matchFinish")
        # rc == L3D_ERR_RETRY on every rank: the pools were enlarged alike, repeat the list pass and the exchange
    return give_up()


def match_images_sharded(l3d, rank, world_size, device=None, group=None, shard_lists=None, **params):
    """matchImages with phase A sharded over `world_size` ranks.  `l3d` is a line3dpp_amd.Line3D that
    already holds all views (every rank adds the same views).

    shard_lists (default: on, L3D_SHARD_LISTS=0 turns it off): the dense part of phase B -- the pass over every 2D
    segment's hypothesis list -- is sharded by views as well (Line3D.listsShard); its records (a few MB) are
    all-gathered and only the cheap sparse remainder of phase B runs replicated.  Off: all of phase B is replicated.

    Keep-all mode (kNN <= 0, line3D.cc:987-992): a row's slot count is only known after the count pass over ALL
    pairs, so there is no fixed slot layout to shard; every rank then runs the whole call itself (replicas, no
    exchange) -- correct by construction, just not faster.

    Failure behaviour: whenever this function returns False after matchBegin succeeded, the context has been
    closed with matchAbort (views untranslated, idle), so the next call starts from a clean state."""
    if world_size == 1 or params.get("kNN", 10) <= 0:
        return l3d.matchImages(**params)   # one call: nothing waits for the GPU between the phases
    if os.environ.get("L3D_DIST_ALLGATHER") is None and os.environ.get("L3D_EXCHANGE_FULL") is None and shard_lists is not False \
            and os.environ.get("L3D_SHARD_LISTS", "1") != "0" and hasattr(l3d, "listsShardViews"):
        return match_images_halo(l3d, rank, world_size, device=device, group=group, **params)
    if shard_lists is None:
        shard_lists = os.environ.get("L3D_SHARD_LISTS", "1") != "0"
    # host wall time between the synchronisation points of the call, summed over calls in l3d.dist_ms (bench.py
    # prints it per step for N > 1: match = begin + this rank's pairs, exchange_slots = pack + collective, lists =
    # expansion of the received pairs + this rank's share of the list pass, exchange_lists, finish)
    import time
    acc = getattr(l3d, "dist_ms", None)
    if acc is None:
        acc = l3d.dist_ms = dict(match=0.0, exchange_slots=0.0, lists=0.0, exchange_lists=0.0, finish=0.0, calls=0)
    t_last = [time.perf_counter()]

    def lap(key):
        now = time.perf_counter()
        acc[key] += 1e3 * (now - t_last[0])
        t_last[0] = now
    acc["calls"] += 1
    if not l3d.matchBegin(**params):
        return False                       # a failing matchBegin restores the context itself

    def give_up():
        l3d.matchAbort()
        return False
    pairs, slot_off = l3d.pairs()
    M = l3d._M
    costs = [M[int(s)] * M[int(t)] for s, t in pairs]
    ranges = pair_ranges(costs, world_size)
    first, count = ranges[rank]
    if count and not l3d.matchPairs(first, count):
        return give_up()
    lap("match")
    n_pairs = len(pairs)
    if os.environ.get("L3D_EXCHANGE_FULL") is None:
        # compact exchange: 4 B per slot (the target index) travel; the rest of a slot is re-derived on arrival
        if count and not l3d.packSlotIndices(first, count):
            return give_up()
        ptr, n_slots = l3d.slot_index_buffer()
        if ptr is None:
            return give_up()
        if n_slots:
            buf = device_tensor(ptr, n_slots * 4, device)
            exchange_slots(buf, slot_byte_ranges(ranges, slot_off, n_slots, slot_bytes=4), group)
            _wait_for_exchange(buf, device, l3d, group)
        lap("exchange_slots")
        ok = (first == 0 or l3d.expandSlotIndices(0, first)) and \
             (first + count == n_pairs or l3d.expandSlotIndices(first + count, n_pairs - first - count))
        if not ok:
            return give_up()
    else:
        # full 32-byte records (L3D_EXCHANGE_FULL=1)
        ptr, n_slots = l3d.slot_buffer()
        if n_slots:
            buf = device_tensor(ptr, n_slots * 32, device)
            exchange_slots(buf, slot_byte_ranges(ranges, slot_off, n_slots), group)
            _wait_for_exchange(buf, device, l3d, group)
        lap("exchange_slots")
        # every pair is now present on this rank
        l3d.L.l3d_slots_exchanged(l3d.h)
    if shard_lists and hasattr(l3d, "listsShard"):
        # phase B's list pass for this rank's views; its record pools are all-gathered slab by slab; the tail of
        # phase B then runs on the complete records on every rank.  Pools that turn out too small are enlarged by
        # l3d_match_finish on EVERY rank alike (all ranks see all counters) and the step is repeated.
        for _ in range(8):
            slabs = l3d.listsShard(rank, world_size)
            if slabs is None:
                return give_up()           # (l3d_lists_shard closes the call on every failing exit; matchAbort is a
                                           # no-op then, and the safety net for a binding that fails before the call)
            lap("lists")
            gather_slabs(slabs, rank, world_size, device, group, l3d)
            lap("exchange_lists")
            rc = l3d.L.l3d_match_finish(l3d.h)
            l3d.last_status = rc
            lap("finish")
            if rc == 0:
                return True
            if rc != -10:                  # L3D_ERR_RETRY
                return l3d._check(rc, "matchFinish")
        return give_up()
    ok = l3d.matchFinish()                 # a failing matchFinish restores the context itself
    lap("finish")
    return ok


def compute_affinity_sharded(l3d, rank, world_size, device=None, group=None):
    """The affinity fill (Line3D.computeAffinity) over `world_size` ranks: after a matchImages whose tail was sharded by
    views (match_images_halo), every rank computes the similarities of its views' surviving matches
    (l3d_affinity_shard_begin), the float values are exchanged in place (one send and one receive per peer, like the
    tail's parts) and the bookkeeping runs on every rank (l3d_affinity_shard_finish): same A_ everywhere.  Falls back to
    the replicated computeAffinity on ALL ranks together when any rank cannot shard (tail not sharded, collinearity
    links, a local failure of begin): the ranks agree through one status exchange, so none is left inside the exchange."""
    if world_size == 1 or os.environ.get("L3D_SHARD_AFFINITY", "1") == "0" or not hasattr(l3d, "affinityShardBegin"):
        return l3d.computeAffinity()
    part = l3d.affinityShardBegin(rank, world_size)
    # (whether the fill can be sharded is a function of state every rank holds alike -- the tail was sharded over this world
    # size, no collinearity links --, so the status exchange is level 2 since round 6: L3D_DIST_STATUS=2)
    if not _all_ok(part is not None, device, group, level=2) or part is None:
        if part is not None:               # this rank's shard is open, a peer's is not: close it WITHOUT the bookkeeping pass
            l3d.affinityShardAbort()       # (the other ranks' similarities never arrived)
        return l3d.computeAffinity()
    exchange_parts([part], rank, world_size, device, group, l3d)
    return l3d.affinityShardFinish()
