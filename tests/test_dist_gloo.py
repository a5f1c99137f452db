"""world_size-2 (and 3) gloo test of the N>1 path's host logic: pair sharding + in-place all-gather(v)
of slot-buffer slices (line3dpp_amd/dist.py).  The slot contents come from the oracle (tests may use
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
        q.put((rank, ok, covered, [c for _, c in ranges]))
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
    for rank, ok, covered, counts in res:
        assert ok and covered, (rank, ok, covered)
        assert sum(counts) == 12 and all(c > 0 for c in counts)
