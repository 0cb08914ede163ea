"""World-size-2 gloo test of the sharding helper (CPU).  The local solve is injected: the HIP path
cannot run here, so each rank uses the CPU oracle as its stand-in solver -- this test covers the
partitioning / scatter / gather plumbing, not the kernels."""
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


def _worker(rank, world, port, B, S, ret):
    sys.path.insert(0, ROOT)
    import oracle
    from tests import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = _load_sharding()

    def local_solve(wp, tm, bc):
        c, _ = oracle.solve_batch(4, wp.numpy(), tm.numpy(), bc.numpy())
        return torch.from_numpy(c)

    wp, tm = synth.make_batch(B, S, config_id=4)
    # resident mode: every rank generates its own rows of the same stream
    lo, hi = sh.shard_bounds(B, world, rank)
    wps, tms = synth.make_batch(hi - lo, S, config_id=4, offset=lo)
    assert np.array_equal(wps, wp[lo:hi]) and np.array_equal(tms, tm[lo:hi])
    res = sh.solve_batch_resident(torch.from_numpy(wps), torch.from_numpy(tms), torch.zeros(1, 4, 3, dtype=torch.float64),
                                  local_solve=local_solve)
    ref, _ = oracle.solve_batch(4, wp, tm)
    assert np.array_equal(res.numpy(), ref[lo:hi])
    # end-to-end mode: scatter from root, gather on root
    if rank == 0:
        out = sh.solve_batch_from_root(torch.from_numpy(wp), torch.from_numpy(tm), order=4, local_solve=local_solve)
        ok = np.array_equal(out.numpy(), ref)
        ret.put(bool(ok))
    else:
        out = sh.solve_batch_from_root(None, None, order=4, local_solve=local_solve, batch=B, segments=S)
        assert out is None
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


@pytest.mark.timeout(180)
def test_scatter_solve_gather_world2():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    B, S = 37, 6   # odd batch: exercises the padded last chunk
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, S, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert ret.get(timeout=5) is True
