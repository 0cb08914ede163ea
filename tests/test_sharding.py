"""World-size-2 gloo tests of the sharding helpers (CPU).  The local solve is injected: the HIP path
cannot run here, so each rank uses the CPU oracle as its stand-in solver -- these tests cover the
partitioning / grouped send-recv scatter / pipelined gather plumbing, not the kernels: uniform batches with
shared and per-trajectory boundary conditions (RootPipeline, chunked), and ragged batches with
per-trajectory boundary conditions (solve_ragged_from_root)."""
import importlib.util
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_sharding():
    spec = importlib.util.spec_from_file_location("csp_sharding", os.path.join(ROOT, "cs-pathplan_amd", "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_solve(order):
    import oracle

    def local_solve(wp, tm, bc, out=None, seg_offsets=None):
        z = np.zeros((2, 3))
        if seg_offsets is None:
            c, _ = oracle.solve_batch(order, wp.numpy(), tm.numpy(), bc.numpy())
            c = torch.from_numpy(c)
        else:
            off = seg_offsets.numpy()
            bcn = bc.numpy().reshape(-1, 4, 3)
            rows = []
            for b in range(len(off) - 1):
                s0, s1 = int(off[b]), int(off[b + 1])
                q = bcn[b if bcn.shape[0] > 1 else 0]
                cb, _ = oracle.solve(order, wp.numpy()[s0 + b:s1 + b + 1], q[[0, 1]], q[[2, 3]], tm.numpy()[s0:s1])
                rows.append(cb.reshape(s1 - s0, 3, 2 * order))
            c = torch.from_numpy(np.concatenate(rows))
        if out is not None:
            out.copy_(c.reshape(out.shape))
            return out
        return c
    return local_solve


def _worker(rank, world, port, B, S, ret):
    sys.path.insert(0, ROOT)
    import oracle
    from tests import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding()
    local_solve = _oracle_local_solve(4)
    ok = True

    wp, tm = synth.make_batch(B, S, config_id=4)
    # resident mode: every rank generates its own rows of the same stream
    lo, hi = sh.shard_bounds(B, world, rank)
    wps, tms = synth.make_batch(hi - lo, S, config_id=4, offset=lo)
    assert np.array_equal(wps, wp[lo:hi]) and np.array_equal(tms, tm[lo:hi])
    res = sh.solve_batch_resident(torch.from_numpy(wps), torch.from_numpy(tms), torch.zeros(1, 4, 3, dtype=torch.float64),
                                  local_solve=local_solve)
    ref, _ = oracle.solve_batch(4, wp, tm)
    assert np.array_equal(res.numpy(), ref[lo:hi])

    # end-to-end mode, shared boundary conditions: grouped scatter, chunked solve, pipelined gather
    for chunks in (1, 3):
        if rank == 0:
            out = sh.solve_batch_from_root(torch.from_numpy(wp), torch.from_numpy(tm), order=4, local_solve=local_solve, chunks=chunks)
            ok &= np.array_equal(out.numpy(), ref)
        else:
            assert sh.solve_batch_from_root(None, None, order=4, local_solve=local_solve, batch=B, segments=S, chunks=chunks) is None

    # per-trajectory boundary conditions travel with their trajectories
    rng = np.random.default_rng(11)
    bc = rng.normal(size=(B, 4, 3))
    ref_bc, _ = oracle.solve_batch(4, wp, tm, bc)
    if rank == 0:
        out = sh.solve_batch_from_root(torch.from_numpy(wp), torch.from_numpy(tm), torch.from_numpy(bc), order=4,
                                       local_solve=local_solve, chunks=2)
        ok &= np.array_equal(out.numpy(), ref_bc)
        ok &= not np.array_equal(ref_bc, ref)
    else:
        assert sh.solve_batch_from_root(None, None, order=4, local_solve=local_solve, batch=B, segments=S, chunks=2,
                                        per_trajectory_bc=True) is None

    # a reusable pipeline object (what bench.py --end-to-end drives), run twice on the same buffers
    pipe = sh.RootPipeline(None, B, S, 4, torch.device("cpu"), chunks=2, dist=dist, rank=rank, world=world, local_solve=local_solve)
    for _ in range(2):
        o2 = pipe.run(torch.from_numpy(wp), torch.from_numpy(tm)) if rank == 0 else pipe.run()
    if rank == 0:
        ok &= np.array_equal(o2.numpy(), ref)

    # ragged batch + per-trajectory boundary conditions, partitioned by cumulative segment count
    trajs = [t for t in synth.make_ragged(60, smin=1, smax=12) if t[0] == 3][:17]
    ls3 = _oracle_local_solve(3)
    rwp = np.concatenate([t[1] for t in trajs])
    rtm = np.concatenate([t[2] for t in trajs])
    off = np.concatenate([[0], np.cumsum([len(t[2]) for t in trajs])]).astype(np.int64)
    rbc = rng.normal(size=(len(trajs), 4, 3))
    if rank == 0:
        whole = ls3(torch.from_numpy(rwp), torch.from_numpy(rtm), torch.from_numpy(rbc), seg_offsets=torch.from_numpy(off))
        out = sh.solve_ragged_from_root(torch.from_numpy(rwp), torch.from_numpy(rtm), torch.from_numpy(off), torch.from_numpy(rbc),
                                        order=3, local_solve=ls3)
        ok &= np.array_equal(out.numpy(), whole.numpy())
        cuts = sh.ragged_partition(torch.from_numpy(off), world)
        ok &= 0 < cuts[1] < len(trajs)        # both ranks got work
        seg = off[cuts]
        ok &= abs((seg[1] - seg[0]) - (seg[2] - seg[1])) <= 12   # balanced to within one trajectory
        ret.put(bool(ok))
    else:
        assert sh.solve_ragged_from_root(None, None, None, order=3, local_solve=ls3) is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_tile_the_batch():
    sh = _load_sharding()
    for B in (0, 1, 7, 64, 65537):
        for w in (1, 2, 3, 8):
            edges = [sh.shard_bounds(B, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1
            for lo, hi in edges:
                for chunks in (1, 4, 7):
                    cb = sh.chunk_bounds(lo, hi, chunks)
                    assert (not cb and hi == lo) or (cb[0][0] == lo and cb[-1][1] == hi)
                    assert all(cb[i][1] == cb[i + 1][0] for i in range(len(cb) - 1))


def test_single_rank_pipeline_needs_no_process_group():
    """N = 1 (bench.py --end-to-end on one GPU): the pipeline degenerates to the chunked local solve."""
    sys.path.insert(0, ROOT)
    import oracle
    from tests import synth
    sh = _load_sharding()
    wp, tm = synth.make_batch(9, 5, config_id=4)
    pipe = sh.RootPipeline(None, 9, 5, 4, torch.device("cpu"), chunks=4, local_solve=_oracle_local_solve(4))
    out = pipe.run(torch.from_numpy(wp), torch.from_numpy(tm))
    ref, _ = oracle.solve_batch(4, wp, tm)
    assert np.array_equal(out.numpy(), ref)


@pytest.mark.timeout(240)
def test_scatter_solve_gather_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    B, S = 37, 6   # odd batch: uneven shards and chunks
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, S, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(200)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) is True
