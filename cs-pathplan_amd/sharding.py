"""Multi-GPU sharding of a trajectory batch: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Trajectories are independent (no cross-trajectory term anywhere in
math_util/minimum_snap.cpp), so the solve itself needs NO collective: each rank solves a
contiguous chunk.  The only exchange steps are the optional ones either side of the solve when
the batch lives on one rank: a scatter of the inputs (536 B/trajectory at S=16) and a gather of
the coefficients (3072 B/trajectory) -- SURVEY.md §8e.  bench.py measures the resident mode
(inputs already sharded in each GPU's HBM); `solve_batch_from_root` is the end-to-end mode.
"""
import importlib

import torch
import torch.distributed as dist


def shard_bounds(batch, world, rank):
    """Contiguous chunk [lo, hi) of rank `rank`; the first batch % world ranks get one extra."""
    base, extra = divmod(int(batch), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _default_local_solve(order, path_weight, vel_zero_weight):
    csp = importlib.import_module("cs-pathplan_amd")

    def solve(wp, tm, bc):
        return csp.solve_batch(wp, tm, bc, order=order, path_weight=path_weight,
                               vel_zero_weight=vel_zero_weight).coeffs
    return solve


def solve_batch_resident(wp_shard, tm_shard, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0,
                         local_solve=None):
    """Every rank already holds its shard: plain local solve, no communication."""
    solve = local_solve or _default_local_solve(order, path_weight, vel_zero_weight)
    return solve(wp_shard, tm_shard, bc)


def solve_batch_from_root(waypoints, times, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0,
                          root=0, group=None, device=None, local_solve=None, batch=None, segments=None,
                          dtype=torch.float64):
    """The full batch lives on `root` (other ranks pass None and give `batch`/`segments`).
    scatter inputs -> local solve -> gather coefficients on root.  Returns [B,S,3,2o] on root,
    None elsewhere.  Chunks are padded to equal size for the collective; padding rows repeat the
    last real trajectory and are dropped after the gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == root:
        batch, segments = times.shape
        dtype = times.dtype
        device = device or times.device
    device = device or torch.device("cpu")
    B, S, m = int(batch), int(segments), 2 * int(order)
    per = (B + world - 1) // world
    wp_loc = torch.empty((per, S + 1, 3), dtype=dtype, device=device)
    tm_loc = torch.empty((per, S), dtype=dtype, device=device)
    wp_list = tm_list = None
    if rank == root:
        wp_list, tm_list = [], []
        for r in range(world):
            lo, hi = r * per, min((r + 1) * per, B)
            idx = torch.arange(lo, lo + per, device=device).clamp_(max=B - 1)
            wp_list.append(waypoints.to(device).index_select(0, idx).contiguous())
            tm_list.append(times.to(device).index_select(0, idx).contiguous())
    dist.scatter(wp_loc, wp_list, src=root, group=group)
    dist.scatter(tm_loc, tm_list, src=root, group=group)
    if bc is not None and bc.reshape(-1, 4, 3).shape[0] != 1:
        raise NotImplementedError("per-trajectory boundary conditions are not scattered yet; pass a shared [4,3]")
    bc_loc = torch.zeros((1, 4, 3), dtype=dtype, device=device)
    if rank == root and bc is not None:
        bc_loc.copy_(bc.reshape(1, 4, 3))
    dist.broadcast(bc_loc, src=root, group=group)
    solve = local_solve or _default_local_solve(order, path_weight, vel_zero_weight)
    co_loc = solve(wp_loc, tm_loc, bc_loc).reshape(per, S, 3, m).contiguous()
    out_list = [torch.empty_like(co_loc) for _ in range(world)] if rank == root else None
    dist.gather(co_loc, out_list, dst=root, group=group)
    if rank != root:
        return None
    return torch.cat(out_list, dim=0)[:B]
