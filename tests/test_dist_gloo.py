"""world_size-2 (and 3) gloo test of the N>1 path's host logic: pair sharding + in-place all-gather(v)
of slot-buffer slices, as full 32-byte records and in the compact 4-byte form (line3dpp_amd/dist.py).  The slot contents come from the oracle (tests may use
it): each rank fills only the slices of its own pairs, the exchange must reproduce the buffer a single
process would have produced."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist_t
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _slots_from_oracle(scene, kNN):
    """full slot buffer (uint8) + per-pair slot offsets as the HIP path lays them out"""
    from line3dpp_amd._lib import EMPTY, SLOT_DTYPE
    from oracle.oracle import Oracle
    o = Oracle(threads=1); o.add_scene(scene); o.begin_match(kNN=kNN)
    _, pairs = scene.pair_tests()
    M = {v.cam: len(v.segs) for v in scene.views}
    offs, bufs, n = [], [], 0
    for s, t in pairs:
        m, off = o.match_pair(s, t)
        sl = np.zeros((M[s], kNN), SLOT_DTYPE); sl["tgt_seg"] = EMPTY
        for r in range(M[s]):
            rows = m[off[r]:off[r + 1]]
            # HIP order: (overlap desc, tgt asc)
            rows = rows[np.lexsort((rows["tgt_seg"], -rows["overlap"]))]
            for j, x in enumerate(rows):
                sl[r, j] = (x["tgt_seg"], x["overlap"], x["d_p1"], x["d_p2"], x["d_q1"], x["d_q2"], 0.0, 0)
        offs.append(n); n += sl.size; bufs.append(sl.reshape(-1))
    o.end_match()
    full = np.concatenate(bufs).view(np.uint8)
    return full, offs, n, pairs, M


class _StubLine3D:
    """stands in for line3dpp_amd.Line3D (no GPU here): records what match_images_sharded asks of it and keeps the
    compact index buffer in host memory"""

    def __init__(self, pairs, offs, n_slots, M, idx_full):
        self.pairs_, self.offs, self.n_slots, self._M = np.asarray(pairs), np.asarray(offs, np.uint64), n_slots, M
        self.truth = idx_full.view(np.uint32)
        self.idx = np.zeros(n_slots, np.uint32)
        self.done = np.zeros(len(pairs), bool)
        self.matched, self.log = [], []
        self.L, self.h = self, None

    fail_pack = False

    def matchBegin(self, **kw):
        self.log.append("begin"); return True

    def matchImages(self, **kw):
        self.log.append("images"); return True

    def matchAbort(self):
        self.log.append("abort"); return True

    def pairs(self):
        return self.pairs_, self.offs

    def _span(self, first, count):
        end = list(map(int, self.offs)) + [self.n_slots]
        return end[first], end[first + count]

    def matchPairs(self, first, count):
        self.matched.append((first, count)); self.done[first:first + count] = True; return True

    def packSlotIndices(self, first, count):
        if self.fail_pack or not self.done[first:first + count].all():
            return False
        lo, hi = self._span(first, count)
        self.idx[lo:hi] = self.truth[lo:hi]
        return True

    def slot_index_buffer(self):
        return self.idx, self.n_slots

    def expandSlotIndices(self, first, count):
        lo, hi = self._span(first, count)
        if not np.array_equal(self.idx[lo:hi], self.truth[lo:hi]):
            return False
        self.done[first:first + count] = True
        return True

    def matchFinish(self):
        self.log.append("finish")
        return bool(self.done.all())


class _ReplayContext:
    """A CPU context behind the interface match_images_sharded drives (line3dpp_amd.api.Line3D): phase A of a pair
    range is COMPUTED here (the restatement's matchingCPU, pair by pair, laid out as the HIP path lays slots out), the
    index buffer the ranks exchange is this object's real buffer, the expansion checks that what arrived for a foreign
    pair is what that pair's owner computed (against its own recomputation), and matchFinish only succeeds when every
    pair is present -- then it produces the scene's real results (surviving matches, A_) for the cross-rank comparison.
    No GPU, no stub answers: a wrong range, a missing broadcast or a mis-sized slice changes the outcome."""

    def __init__(self, scene, kNN):
        from line3dpp_amd._lib import EMPTY
        from oracle.oracle import Oracle
        self.scene, self.kNN, self.EMPTY = scene, kNN, EMPTY
        self.o = Oracle(threads=1); self.o.add_scene(scene)
        self._M = {v.cam: len(v.segs) for v in scene.views}
        self.state, self.log = "idle", []
        self.L, self.h = self, None

    def _pair_indices(self, s, t):
        m, off = self.o.match_pair(int(s), int(t))
        sl = np.full((self._M[int(s)], self.kNN), self.EMPTY, np.uint32)
        for r in range(self._M[int(s)]):
            rows = m[off[r]:off[r + 1]]
            sl[r, :len(rows)] = rows["tgt_seg"]          # the reference's own row order (priority_queue pops)
        return sl.reshape(-1)

    def matchBegin(self, **kw):
        assert self.state == "idle"
        self.o.begin_match(kNN=self.kNN)
        _, prs = self.scene.pair_tests()
        self.pairs_ = np.array(prs, np.uint32).reshape(-1, 2)
        sizes = [self._M[int(s)] * self.kNN for s, _ in self.pairs_]
        self.offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
        self.n_slots = int(self.offs[-1])
        self.idx = np.full(self.n_slots, 0xDEADBEEF, np.uint32)        # poison: an unexchanged slice is noticed
        self.present = np.zeros(len(self.pairs_), bool)
        self.packed = np.zeros(len(self.pairs_), bool)
        self.own = {}
        self.state = "begun"; self.log.append("begin")
        return True

    def pairs(self):
        return self.pairs_, self.offs[:-1]

    def matchPairs(self, first, count):
        if self.state != "begun":
            return False
        for p in range(first, first + count):
            self.own[p] = self._pair_indices(*self.pairs_[p])
            self.present[p] = True
        return True

    def packSlotIndices(self, first, count):
        for p in range(first, first + count):
            if p not in self.own:
                return False
            self.idx[int(self.offs[p]):int(self.offs[p + 1])] = self.own[p]
            self.packed[p] = True
        return True

    def slot_index_buffer(self):
        return self.idx, self.n_slots

    def expandSlotIndices(self, first, count):
        for p in range(first, first + count):
            if self.present[p]:
                return False
            got = self.idx[int(self.offs[p]):int(self.offs[p + 1])]
            if not np.array_equal(got, self._pair_indices(*self.pairs_[p])):
                return False
            self.present[p] = True
        return True

    def matchAbort(self):
        if self.state == "begun":
            self.o.end_match()
        self.state = "idle"; self.log.append("abort")
        return True

    def matchFinish(self):
        if self.state != "begun" or not self.present.all():
            return False
        self.o.end_match()
        self.o.match_images(kNN=self.kNN); self.o.compute_affinity()
        self.state = "matched"; self.log.append("finish")
        return True

    def digest(self):
        import hashlib
        h = hashlib.sha256()
        for v in self.scene.views:
            m, off = self.o.matches(v.cam)
            h.update(m.tobytes()); h.update(off.tobytes())
        e, l = self.o.affinity()
        h.update(e.tobytes()); h.update(l.tobytes())
        return h.hexdigest()


def _replayed_run(dist, rank, world, scene, kNN):
    """match_images_sharded over gloo on a context that really matches its pair range; returns the result digest"""
    ctx = _ReplayContext(scene, kNN)
    real = dist.device_tensor
    dist.device_tensor = lambda arr, nbytes, device: torch.from_numpy(arr.view(np.uint8))[:nbytes]
    try:
        ok = dist.match_images_sharded(ctx, rank, world, device=None, kNN=kNN)
    finally:
        dist.device_tensor = real
    M = ctx._M
    ranges = dist.pair_ranges([M[int(s)] * M[int(t)] for s, t in ctx.pairs_], world)
    mine = set(range(ranges[rank][0], ranges[rank][0] + ranges[rank][1]))
    assert ok and ctx.log == ["begin", "finish"], ctx.log
    assert set(ctx.own) == mine, "a rank matches exactly its own range"
    assert not (ctx.idx == 0xDEADBEEF).any(), "every slice arrived"
    return ctx.digest()


def _sharded_control_flow(dist, rank, world, pairs, offs, n_slots, M, idx_full):
    """match_images_sharded end to end over gloo with the stub: own range matched and packed, every foreign pair
    expanded from what the exchange delivered, finish only after that"""
    stub = _StubLine3D(pairs, offs, n_slots, M, idx_full)
    real = dist.device_tensor
    dist.device_tensor = lambda arr, nbytes, device: torch.from_numpy(arr.view(np.uint8))[:nbytes]
    try:
        ok = dist.match_images_sharded(stub, rank, world, device=None, kNN=5)
    finally:
        dist.device_tensor = real
    ranges = dist.pair_ranges([M[s] * M[t] for s, t in pairs], world)
    ok = bool(ok) and stub.matched == [ranges[rank]] and stub.log == ["begin", "finish"] and \
        np.array_equal(stub.idx, stub.truth)
    # keep-all mode (kNN <= 0) has no fixed slot layout: every rank runs the whole call, nothing is begun or exchanged
    keep_all = _StubLine3D(pairs, offs, n_slots, M, idx_full)
    ok = ok and dist.match_images_sharded(keep_all, rank, world, device=None, kNN=0) and keep_all.log == ["images"]
    # a failure between begin and finish closes the context (views untranslated, idle) before False is returned
    broken = _StubLine3D(pairs, offs, n_slots, M, idx_full); broken.fail_pack = True
    ok = ok and dist.match_images_sharded(broken, rank, world, device=None, kNN=5) is False and \
        broken.log == ["begin", "abort"]
    return ok


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from line3dpp_amd import dist
    from line3dpp_amd.scene import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # world 2: 12 uniform pairs split evenly -> the single all-gather path; world 3 with ragged views ->
        # uneven slices -> the broadcast path
        scene = make_scene(6, 90 + 7, n_neighbors=4, seed=4)
        if world == 3:
            for i, v in enumerate(scene.views):
                v.segs = v.segs[:97 - 9 * i].copy()
        full, offs, n_slots, pairs, M = _slots_from_oracle(scene, 5)
        ranges = dist.pair_ranges([M[s] * M[t] for s, t in pairs], world)
        br = dist.slot_byte_ranges(ranges, offs, n_slots)
        mine = np.zeros_like(full)
        lo, hi = br[rank]
        mine[lo:hi] = full[lo:hi]            # this rank matched only its own pairs
        buf = torch.from_numpy(mine)
        how = dist.exchange_slots(buf, br)
        ok = bool(np.array_equal(buf.numpy(), full)) and how == ("all_gather" if world == 2 else "broadcast")
        covered = sum(h - l for l, h in br) == len(full)
        # the compact form the product exchanges by default: the uint32 target index of every slot
        from line3dpp_amd._lib import SLOT_DTYPE
        idx_full = np.ascontiguousarray(full.view(SLOT_DTYPE)["tgt_seg"]).view(np.uint8)
        br4 = dist.slot_byte_ranges(ranges, offs, n_slots, slot_bytes=4)
        mine4 = np.zeros_like(idx_full)
        lo, hi = br4[rank]
        mine4[lo:hi] = idx_full[lo:hi]
        buf4 = torch.from_numpy(mine4)
        how4 = dist.exchange_slots(buf4, br4)
        ok = ok and bool(np.array_equal(buf4.numpy(), idx_full)) and how4 == how
        covered = covered and sum(h - l for l, h in br4) == len(idx_full) == 4 * n_slots
        ok = ok and _sharded_control_flow(dist, rank, world, pairs, offs, n_slots, M, idx_full)
        # the same control flow on a context that computes: every rank ends with the single-process result
        digest = _replayed_run(dist, rank, world, scene, 5)
        single = _ReplayContext(scene, 5)
        assert single.matchBegin() and single.matchPairs(0, len(single.pairs_)) and single.matchFinish()
        ok = ok and digest == single.digest()
        q.put((rank, ok, covered, [c for _, c in ranges], digest))
    finally:
        dist_t.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_reproduces_single_process_buffer(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, covered, counts, digest in res:
        assert ok and covered, (rank, ok, covered)
        assert sum(counts) == 12 and all(c > 0 for c in counts)
    assert len({r[4] for r in res}) == 1, "all ranks hold the same surviving matches and affinity matrix"


# ---- the halo form (default of match_images_sharded since round 3) ---------------------------------------------------
class _HaloReplayContext(_ReplayContext):
    """_ReplayContext plus the view-sharded list pass (Line3D.listsShardViews) and the retry protocol of
    l3d_match_finish, so that line3dpp_amd.dist.match_images_halo runs end to end on the CPU: point-to-point exchange of
    the pairs across the cuts, expansion, list pass of the rank's views with ONLY the pairs that touch them present,
    in-place all-gather of the record slabs, L3D_ERR_RETRY once, finish on the records of all ranks."""
    SLAB = 256

    def __init__(self, scene, kNN, rank, world):
        super().__init__(scene, kNN)
        self.rank, self.world = rank, world
        self.attempt, self.full, self.last_status = 0, None, 0
        cams = sorted(self._M)
        self.vidx = {c: i for i, c in enumerate(cams)}

    @staticmethod
    def _sig(r, v0, v1, attempt, k):
        import hashlib
        d = hashlib.sha256(f"{r}:{v0}:{v1}:{attempt}:{k}".encode()).digest()
        return np.frombuffer((d * (_HaloReplayContext.SLAB // len(d) + 1))[:_HaloReplayContext.SLAB], np.uint8)

    def listsShardViews(self, rank, world, v0, v1):
        assert (rank, world) == (self.rank, self.world) and self.state == "begun"
        for p, (s, t) in enumerate(self.pairs_):
            touches = v0 <= self.vidx[int(s)] < v1 or v0 <= self.vidx[int(t)] < v1
            if touches and not self.present[p]:
                self.matchAbort(); self.last_status = -7
                return None
        self.ranges_seen = (v0, v1)
        self.full = [np.zeros(world * self.SLAB, np.uint8) for _ in range(4)]
        for k in range(4):
            self.full[k][rank * self.SLAB:(rank + 1) * self.SLAB] = self._sig(rank, v0, v1, self.attempt, k)
        return [(None, self.SLAB, self.full[k]) for k in range(4)]

    def l3d_match_finish(self, h):
        if not self._slabs_arrived(sharded_tail=False):
            return -8                                # a slab that did not arrive (or arrived from another attempt)
        if self.attempt == 0:                        # "pools enlarged on every rank alike": repeat list pass + gather
            self.attempt = 1; self.log.append("retry")
            return -10
        self.o.end_match()
        self.o.match_images(kNN=self.kNN); self.o.compute_affinity()
        self.state = "matched"; self.log.append("finish")
        return 0

    def _check(self, rc, what):
        self.last_status = rc
        return rc == 0

    # ---- the sharded tail (l3d_tail_shard_count / _layout / _commit): nine arrays with parts of rank-dependent size
    # (one rank's are empty), every part carrying a signature that the commit checks for all ranks ----
    ELT = [40, 4, 4, 128, 8, 4, 4, 4, 4]

    def _counts(self, r):
        return (0, 0) if r == self.world - 1 and self.world > 2 else (10 + 3 * r, 2 + r)

    def _part_sig(self, r, k, nbytes):
        import hashlib
        d = hashlib.sha256(f"part:{r}:{k}:{self.attempt}".encode()).digest()
        return np.frombuffer((d * (nbytes // len(d) + 1))[:nbytes], np.uint8)

    def shardOptions(self, first_needed_rank=0, exchanges_stream_ordered=False):
        self.first_needed = int(first_needed_rank); self.log_options = (int(first_needed_rank), bool(exchanges_stream_ordered))
        return True

    def _slabs_arrived(self, sharded_tail=True):
        """round 6: the record arrays (k < 3) of the ranks this rank's chain DEPENDS on (dist.shard_needs) and of itself, the
        counter slab (k = 3) of every rank; the records of a rank it does not depend on must NOT have been sent to it"""
        vb = self.halo_plan["view_bounds"]
        needs = set(self.halo_plan["needs"][self.rank]) | {self.rank} if sharded_tail else set(range(self.world))
        assert getattr(self, "first_needed", 0) == min(needs)
        for r in range(self.world):
            for k in range(4):
                got = self.full[k][r * self.SLAB:(r + 1) * self.SLAB]
                want = self._sig(r, int(vb[r]), int(vb[r + 1]), self.attempt, k)
                if k == 3 or r in needs:
                    if not np.array_equal(got, want):
                        return False
                elif got.any():
                    return False                     # something arrived that nobody should have sent
        return True

    def tailShardCount(self):
        if not self._slabs_arrived():
            self.last_status = -8
            return -8, 0, 0
        if self.attempt == 0:                        # "pools enlarged on every rank alike": repeat list pass + gather
            self.attempt = 1; self.log.append("retry")
            return -10, 0, 0
        self.log.append("tail_count")
        return (0,) + self._counts(self.rank)

    def tailShardLayout(self, world, counts_all, view_bounds):
        assert world == self.world and list(counts_all) == [self._counts(r) for r in range(world)]
        assert [int(v) for v in view_bounds] == [int(v) for v in self.halo_plan["view_bounds"]]
        base_n = np.concatenate([[0], np.cumsum([c[0] for c in counts_all])]); base_h = np.concatenate([[0], np.cumsum([c[1] for c in counts_all])])
        seg = np.concatenate([[0], np.cumsum([5] * world)])                       # 5 "segments" and one "view" per rank
        self.parts = []
        for k, elt in enumerate(self.ELT):
            if k < 3:
                rng_ = [(int(base_n[r]), int(counts_all[r][0])) for r in range(world)]
            elif k < 5:
                rng_ = [(int(base_h[r]), int(counts_all[r][1])) for r in range(world)]
            elif k < 8:
                rng_ = [(int(seg[r]), 5 + (1 if k < 7 and r == world - 1 else 0)) for r in range(world)]
            else:
                rng_ = [(r, 1) for r in range(world)]
            total = max(f + n for f, n in rng_) * elt
            arr = np.full(max(total, 1), 0xEE, np.uint8)
            f, n = rng_[self.rank]
            arr[f * elt:(f + n) * elt] = self._part_sig(self.rank, k, n * elt)
            self.parts.append((arr, elt, rng_))
        self.log.append("tail_layout")
        return list(self.parts)

    def tailShardCommit(self):
        for k, (arr, elt, rng_) in enumerate(self.parts):
            for r, (f, n) in enumerate(rng_):
                if not np.array_equal(arr[f * elt:(f + n) * elt], self._part_sig(r, k, n * elt)):
                    self.last_status = -8
                    return -8                        # a part that did not arrive
        self.o.end_match()
        self.o.match_images(kNN=self.kNN); self.o.compute_affinity()
        self.state = "matched"; self.log.append("finish")
        self.last_status = 0
        return 0


def _aff_methods():
    """the sharded affinity fill (l3d_affinity_shard_begin / _finish) on the replay context: a float array whose parts
    have rank-dependent sizes (one rank's is empty), each carrying a signature that finish checks for ALL ranks; only after
    a sharded tail (as the library: the parts are the tail's), else None -> every rank falls back to computeAffinity"""
    def counts(self, r):
        return 0 if r == self.world - 1 and self.world > 2 else 7 + 2 * r

    def sig(self, r, n):
        import hashlib
        d = hashlib.sha256(f"aff:{r}".encode()).digest()
        return np.frombuffer((d * (4 * n // len(d) + 1))[:4 * n], np.uint8)

    def begin(self, rank, world):
        assert (rank, world) == (self.rank, self.world) and self.state == "matched"
        if "tail_count" not in self.log:
            self.last_status = -3
            return None
        base = np.concatenate([[0], np.cumsum([counts(self, r) for r in range(world)])])
        self.aff = np.full(4 * max(int(base[-1]), 1), 0xEE, np.uint8)
        self.aff_parts = [(int(base[r]), counts(self, r)) for r in range(world)]
        f, n = self.aff_parts[rank]
        self.aff[4 * f:4 * (f + n)] = sig(self, rank, n)
        self.log.append("aff_begin")
        return (self.aff, 4, list(self.aff_parts))

    def finish(self):
        for r, (f, n) in enumerate(self.aff_parts):
            if not np.array_equal(self.aff[4 * f:4 * (f + n)], sig(self, r, n)):
                return False                      # a part that did not arrive
        self.log.append("aff_finish")
        return True

    def whole(self):
        assert "aff_begin" not in self.log or "aff_finish" in self.log or "aff_abort" in self.log, "computeAffinity inside an open shard"
        self.log.append("affinity")
        return True

    def abort(self):
        self.log.append("aff_abort")
        return True
    return begin, finish, whole, abort


(_HaloReplayContext.affinityShardBegin, _HaloReplayContext.affinityShardFinish, _HaloReplayContext.computeAffinity,
 _HaloReplayContext.affinityShardAbort) = _aff_methods()


def _halo_worker(rank, world, port, q, shard_tail=True):
    sys.path.insert(0, ROOT)
    from line3dpp_amd import dist
    from line3dpp_amd.scene import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene = make_scene(9, 60, n_neighbors=4, seed=6)
        ctx = _HaloReplayContext(scene, 5, rank, world)
        os.environ["L3D_SHARD_TAIL"] = "1" if shard_tail else "0"
        real = dist.device_tensor
        dist.device_tensor = lambda arr, nbytes, device: torch.from_numpy(arr.view(np.uint8))[:nbytes]
        try:
            ok = dist.match_images_sharded(ctx, rank, world, device=None, kNN=5)
            # the affinity fill sharded by the same views (every rank's similarities arrive everywhere), or -- the tail was
            # not sharded -- the replicated fill on every rank, decided together
            ok = ok and dist.compute_affinity_sharded(ctx, rank, world, device=None)
        finally:
            dist.device_tensor = real
        plan = ctx.halo_plan
        pb, vb, runs = plan["pair_bounds"], plan["view_bounds"], plan["runs"]
        mine = set(range(int(pb[rank]), int(pb[rank + 1])))
        incoming = {p for r in range(world) for (qq, f, n) in runs[r] if qq == rank for p in range(f, f + n)}
        ok = bool(ok) and ctx.log == (["begin", "retry", "tail_count", "tail_layout", "finish", "aff_begin", "aff_finish"] if shard_tail
                                      else ["begin", "retry", "finish", "affinity"])
        ok = ok and set(ctx.own) == mine                                   # a rank matches exactly the pairs it owns
        ok = ok and set(np.nonzero(ctx.present)[0].tolist()) == mine | incoming   # and holds those plus its halo, nothing else
        ok = ok and ctx.ranges_seen == (int(vb[rank]), int(vb[rank + 1]))
        single = _ReplayContext(scene, 5)
        assert single.matchBegin() and single.matchPairs(0, len(single.pairs_)) and single.matchFinish()
        q.put((rank, ok and ctx.digest() == single.digest(), len(mine), len(incoming), sum(len(r) for r in runs)))
    finally:
        dist_t.destroy_process_group()


@pytest.mark.parametrize("world,shard_tail", [(2, True), (3, True), (4, True), (8, True), (3, False)])
def test_halo_form_over_gloo(world, shard_tail):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q, shard_tail)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert sum(r[2] for r in res) == 18                  # 9 views x 4 neighbours / 2: every pair matched exactly once
    assert sum(r[3] for r in res) > 0 and res[0][4] > 0  # pairs did cross the cuts


def test_halo_plan_balances_pairs_and_lists_every_crossing_pair_once():
    """l3d_plan_shards + plan_halo on BASELINE C2's pair list (host only): contiguous view ranges with equal pair cost,
    every pair owned by the rank of its source view, every pair whose target view lies elsewhere in exactly one run."""
    sys.path.insert(0, ROOT)
    from line3dpp_amd import dist
    from line3dpp_amd.scene import CONFIGS
    n, nb = CONFIGS["C2"]["n_views"], CONFIGS["C2"]["n_neighbors"]
    # the pair list of a ring of n views with +-nb/2 neighbours (line3D.cc:704-741), without building the scene
    M = {c: 4096 for c in range(n)}
    matched = {c: set() for c in range(n)}
    pairs = []
    for c in range(n):
        for t in sorted((c + d) % n for d in range(-nb // 2, nb // 2 + 1) if d):
            if t not in matched[c]:
                pairs.append((c, t)); matched[c].add(t); matched[t].add(c)
    assert len(pairs) == 2560
    for world in (2, 3, 8):
        plan = dist.plan_halo(pairs, M, world)
        vb, pb, runs = plan["view_bounds"], plan["pair_bounds"], plan["runs"]
        counts = np.diff(pb.astype(np.int64))
        assert counts.sum() == 2560 and counts.max() - counts.min() <= 2 * nb, counts
        owner = np.searchsorted(vb[1:], np.arange(n), side="right")
        crossing = {p for p, (s, t) in enumerate(pairs) if owner[s] != owner[t]}
        listed = [p for r in range(world) for (_, f, k) in runs[r] for p in range(f, f + k)]
        assert len(listed) == len(set(listed)) and set(listed) == crossing
        for r in range(world):
            for (qq, f, k) in runs[r]:
                assert all(owner[pairs[p][0]] == r and owner[pairs[p][1]] == qq for p in range(f, f + k))
            halo = [p for (_, f, k) in runs[r] for p in range(f, f + k)]
            early, late = dist.early_ranges(int(pb[r]), int(counts[r]), halo)
            assert sum(k for _, k in early) + late[1] == counts[r]
            assert not any(late[0] <= p < late[0] + late[1] for p in halo)     # the stretch matched last sends nothing
        assert len(crossing) < 0.2 * len(pairs) * (world - 1) / world + 120


# ---- a rank that fails locally does not strand the others (dist.py) --------------------------------------------
class _FailingHaloContext(_HaloReplayContext):
    """the halo context with ONE local failure on ONE rank: `where` = "match" (a matchPairs call returns False: a HIP error
    in the pair kernel), "expand" (expandSlotIndices refuses what arrived) or "finish" (l3d_match_finish fails with an
    error that is neither success nor RETRY on this rank only)"""

    def __init__(self, scene, kNN, rank, world, fail_rank, where):
        super().__init__(scene, kNN, rank, world)
        self.failing = rank == fail_rank
        self.where = where

    def matchPairs(self, first, count):
        ok = super().matchPairs(first, count)
        return ok and not (self.failing and self.where == "match")

    def expandSlotIndices(self, first, count):
        ok = super().expandSlotIndices(first, count)
        return ok and not (self.failing and self.where == "expand")

    def l3d_match_finish(self, h):
        rc = super().l3d_match_finish(h)
        if self.failing and self.where == "finish" and rc in (0, -10):
            self.matchAbort()              # (l3d_match_finish closes the call itself when it fails, l3d_api.hip)
            return -4
        return rc

    def tailShardCount(self):              # the sharded tail: the same two failure points
        rc, n, h = super().tailShardCount()
        if self.failing and self.where == "tail_count" and rc in (0, -10):
            self.matchAbort()
            return -4, 0, 0
        return rc, n, h

    def tailShardCommit(self):
        if self.failing and self.where == "finish":
            self.matchAbort()
            self.last_status = -4
            return -4
        return super().tailShardCommit()


def _failing_worker(rank, world, port, q, fail_rank, where, status_level="1"):
    sys.path.insert(0, ROOT)
    os.environ["L3D_DIST_STATUS"] = status_level
    from line3dpp_amd import dist
    from line3dpp_amd.scene import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene = make_scene(9, 40, n_neighbors=4, seed=6)
        ctx = _FailingHaloContext(scene, 5, rank, world, fail_rank, where)
        real = dist.device_tensor
        dist.device_tensor = lambda arr, nbytes, device: torch.from_numpy(arr.view(np.uint8))[:nbytes]
        try:
            ok = dist.match_images_sharded(ctx, rank, world, device=None, kNN=5)
        finally:
            dist.device_tensor = real
        # every rank gave up, together, and left its context idle (a later call starts clean)
        q.put((rank, bool(ok), ctx.state, ctx.log[0], "finish" in ctx.log and ctx.state == "matched"))
    finally:
        dist_t.destroy_process_group()


@pytest.mark.parametrize("where", ["match", "expand", "tail_count", "finish", "finish_default_level"])
def test_a_rank_that_fails_locally_takes_all_ranks_out_of_the_call_together(where):
    """round 3's early returns left the peers of a failing rank inside a collective for ever; now the failing rank keeps
    posting what the plan says and all ranks give up at the next status exchange (dist._all_ok): nobody hangs, nobody
    returns True, every context is idle again"""
    world, fail_rank = 3, 1
    # Round 6: the status exchanges of the default level are the one in front of the record gather and the one that travels
    # with the tail's counts; the guard around the COMMIT (a failure after every exchange of the call is done) is level 2
    # (L3D_DIST_STATUS=2): with it all ranks report the failure, without it the healthy ranks keep their (complete) result
    # and only the failing rank reports -- nobody hangs either way
    level = "2" if where == "finish" else "1"
    default_commit = where == "finish_default_level"
    where = "finish" if default_commit else where
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, fail_rank, where, level)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))          # a hang is a timeout here
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    if default_commit:
        assert [r[1] for r in res] == [r[0] != fail_rank for r in res], res
        assert all(r[2] == ("idle" if r[0] == fail_rank else "matched") for r in res), res
        return
    assert [r[1] for r in res] == [False] * world, res
    assert all(r[3] == "begin" for r in res)
    if where in ("finish", "tail_count"):
        # the two healthy ranks had closed the call (rc 0 / RETRY) when they learnt of the failure: results discarded or
        # context aborted; the failing one reports its own error -- nobody claims a result
        assert all(r[2] in ("idle", "matched") for r in res), res
    else:
        assert all(r[2] == "idle" for r in res), res


# ---- the affinity fill's fallback (round 6: ADVICE r5 + the status levels) ---------------------------------------------------
def _aff_fallback_worker(rank, world, port, q, mode):
    """mode "none_can": no rank can shard (the tail was not sharded) -> every rank takes the replicated fill, no exchange, no
    status all-reduce at the default level.  mode "one_cannot": ONE rank's begin fails although its peers' succeed (a local
    failure); with the single-rank guards on (L3D_DIST_STATUS=2) the peers close their open shards WITHOUT the bookkeeping
    pass (affinityShardAbort) and all ranks take the replicated fill together."""
    sys.path.insert(0, ROOT)
    from line3dpp_amd import dist
    from line3dpp_amd.scene import make_scene
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["L3D_DIST_STATUS"] = "2" if mode == "one_cannot" else "1"
    os.environ["L3D_SHARD_TAIL"] = "0" if mode == "none_can" else "1"
    dist_t.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scene = make_scene(9, 40, n_neighbors=4, seed=6)
        ctx = _HaloReplayContext(scene, 5, rank, world)
        if mode == "one_cannot" and rank == 1:
            ctx.affinityShardBegin = lambda r, w: None            # this rank's begin fails
        real = dist.device_tensor
        dist.device_tensor = lambda arr, nbytes, device: torch.from_numpy(arr.view(np.uint8))[:nbytes]
        try:
            ok = dist.match_images_sharded(ctx, rank, world, device=None, kNN=5)
            ok = ok and dist.compute_affinity_sharded(ctx, rank, world, device=None)
        finally:
            dist.device_tensor = real
        q.put((rank, bool(ok), [x for x in ctx.log if x.startswith("aff") or x == "affinity"]))
    finally:
        dist_t.destroy_process_group()


@pytest.mark.parametrize("mode", ["none_can", "one_cannot"])
def test_affinity_fill_falls_back_on_every_rank_together(mode):
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aff_fallback_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    if mode == "none_can":
        assert all(r[2] == ["affinity"] for r in res), res
    else:
        assert res[1][2] == ["affinity"], res                       # the failing rank never opened a shard
        assert all(r[2] == ["aff_begin", "aff_abort", "affinity"] for r in (res[0], res[2])), res
