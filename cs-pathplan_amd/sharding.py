"""Multi-GPU sharding of a trajectory batch: one process per GPU, torch.distributed (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Trajectories are independent (no cross-trajectory term anywhere in
math_util/minimum_snap.cpp), so the solve itself needs NO collective: each rank solves a
contiguous chunk.  The only exchange steps are the optional ones either side of the solve when
the batch lives on one rank: a scatter of the inputs (536 B/trajectory at S=16)
and a gather of the coefficients (3072 B/trajectory).  RCCL has no native scatter/gather, so both
are GROUPED point-to-point sends/receives from/to the root (`batch_isend_irecv` = ncclGroupStart ..
ncclSend/ncclRecv .. ncclGroupEnd): every peer uses its own direct xGMI link to the root.

  * resident mode (bench.py's headline): every rank already holds its shard -- `solve_batch_resident`;
  * end-to-end mode: `RootPipeline` (uniform batches; shared or per-trajectory boundary conditions) cuts
    every rank's shard into chunks, and the coefficients of chunk i travel to the root while chunk i+1 is
    being solved (the sends run on the communicator's stream, ordered after the solve that produced them;
    the next solve is enqueued on the compute stream without waiting for them);
  * `solve_ragged_from_root`: ragged batches (per-trajectory segment counts) partitioned by cumulative
    segment count, per-trajectory boundary conditions included.

`local_solve(wp, tm, bc, out, seg_offsets=None)` is injectable so that the world-size-2 gloo tests can
exercise the partitioning and the exchange on CPU tensors (there is no CPU solver in the product).
"""
import importlib

import torch
import torch.distributed as dist


def shard_bounds(batch, world, rank, align=1):
    """Contiguous chunk [lo, hi) of rank `rank`; the first batch % world ranks get one extra.  With `align` > 1 every
    interior boundary is a multiple of `align` (the register-resident kernels read 16-byte pieces: a slice of a
    [B][S+1][3] array that starts at an odd trajectory is not 16-byte aligned)."""
    batch, world, align = int(batch), int(world), int(align)
    if align > 1:
        units = (batch + align - 1) // align
        lo, hi = shard_bounds(units, world, rank)
        return min(lo * align, batch), min(hi * align, batch)
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_bounds(lo, hi, chunks, align=1):
    """[lo, hi) cut into `chunks` contiguous balanced pieces (empty pieces dropped); interior boundaries are multiples of
    `align` relative to `lo` (which the caller keeps aligned)."""
    n = hi - lo
    units = (n + align - 1) // align
    out = []
    for c in range(chunks):
        a, b = lo + min(units * c // chunks * align, n), lo + min(units * (c + 1) // chunks * align, n)
        if b > a:
            out.append((a, b))
    return out


def boundary_alignment(total, world, chunks):
    """Whole 64-trajectory slices per piece when the batch is big enough for that, otherwise multiples of 4 trajectories
    (enough for 16-byte alignment of fp64 and fp32 rows of any segment count)."""
    return 64 if total >= 64 * world * chunks else 4


def _default_local_solve(order, path_weight=0.0, vel_zero_weight=0.0):
    csp = importlib.import_module("cs-pathplan_amd")

    def solve(wp, tm, bc, out=None, seg_offsets=None):
        return csp.solve_batch(wp, tm, bc, order=order, path_weight=path_weight, vel_zero_weight=vel_zero_weight,
                               out=out, seg_offsets=seg_offsets).coeffs
    return solve


def solve_batch_resident(wp_shard, tm_shard, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0,
                         local_solve=None):
    """Every rank already holds its shard: plain local solve, no communication."""
    solve = local_solve or _default_local_solve(order, path_weight, vel_zero_weight)
    return solve(wp_shard, tm_shard, bc)


def _p2p(ops):
    """One grouped launch of point-to-point operations; returns the work handles ([] for no ops)."""
    return dist.batch_isend_irecv(ops) if ops else []


class RootPipeline:
    """End-to-end mode for a uniform batch that lives on `root`: scatter -> solve -> gather, pipelined over chunks.

    Every rank (the root included) owns the contiguous shard `shard_bounds(total, world, rank)`, cut into `chunks`
    pieces.  Per `run()`:
      root    posts, chunk by chunk, one grouped send of (waypoints, times[, bc]) to every peer, and one grouped
              receive of every peer's coefficients straight into its slice of the output (no concatenation); it
              solves its own shard in place meanwhile;
      peer    waits for chunk c's inputs, solves it, hands the coefficients to the communicator (ordered after
              the solve) and goes on to chunk c+1 without waiting for the send.
    Buffers are allocated once; `run()` returns the [total,S,3,2o] coefficients on the root, None elsewhere."""

    def __init__(self, csp, total, segments, order, device, chunks=4, dist=None, rank=0, world=1, group=None, root=0,
                 per_trajectory_bc=False, dtype=torch.float64, local_solve=None, path_weight=0.0, vel_zero_weight=0.0):
        self.total, self.S, self.o, self.m = int(total), int(segments), int(order), 2 * int(order)
        self.dev, self.dist, self.rank, self.world, self.group, self.root = device, dist, rank, world, group, root
        self.per_bc, self.dtype = bool(per_trajectory_bc), dtype
        self.solve = local_solve or _default_local_solve(order, path_weight, vel_zero_weight)
        self.chunks = max(1, int(chunks))
        self.align = boundary_alignment(self.total, world, self.chunks)
        self.lo, self.hi = shard_bounds(self.total, world, rank, self.align)
        self.local_count = self.hi - self.lo
        self.my_chunks = chunk_bounds(self.lo, self.hi, self.chunks, self.align)
        n = self.local_count
        if rank == root:
            self.out = torch.empty((self.total, self.S, 3, self.m), dtype=dtype, device=device)
        else:
            self.wp = torch.empty((n, self.S + 1, 3), dtype=dtype, device=device)
            self.tm = torch.empty((n, self.S), dtype=dtype, device=device)
            self.co = torch.empty((n, self.S, 3, self.m), dtype=dtype, device=device)
            self.bc = torch.empty((n if self.per_bc else 1, 4, 3), dtype=dtype, device=device)

    def _peers(self):
        return [r for r in range(self.world) if r != self.root]

    def run(self, waypoints=None, times=None, bc=None):
        d, S = self.dist, self.S
        multi = d is not None and self.world > 1
        if self.rank == self.root:
            if bc is None:
                bc = torch.zeros((1, 4, 3), dtype=self.dtype, device=self.dev)
            bc = bc.reshape(-1, 4, 3)
            pending = []
            if multi:
                if not self.per_bc:
                    pending += _p2p([d.P2POp(d.isend, bc, r, self.group) for r in self._peers()])
                # scatter: one group per chunk index, so that every peer's first chunk arrives first
                for c in range(self.chunks):
                    ops = []
                    for r in self._peers():
                        cb = chunk_bounds(*shard_bounds(self.total, self.world, r, self.align), self.chunks, self.align)
                        if c < len(cb):
                            a, b = cb[c]
                            ops.append(d.P2POp(d.isend, waypoints[a:b], r, self.group))
                            ops.append(d.P2POp(d.isend, times[a:b], r, self.group))
                            if self.per_bc:
                                ops.append(d.P2POp(d.isend, bc[a:b], r, self.group))
                    pending += _p2p(ops)
                # gather: receives posted up front, straight into the output slices
                for c in range(self.chunks):
                    ops = []
                    for r in self._peers():
                        cb = chunk_bounds(*shard_bounds(self.total, self.world, r, self.align), self.chunks, self.align)
                        if c < len(cb):
                            a, b = cb[c]
                            ops.append(d.P2POp(d.irecv, self.out[a:b], r, self.group))
                    pending += _p2p(ops)
            for a, b in self.my_chunks:   # the root's own shard, in place
                self.solve(waypoints[a:b], times[a:b], bc[a:b] if self.per_bc else bc, out=self.out[a:b])
            for w in pending:
                w.wait()
            return self.out
        # ---- peer ----
        recvs = []
        if not self.per_bc:
            recvs.append(_p2p([d.P2POp(d.irecv, self.bc, self.root, self.group)]))
        per_chunk = []
        for a, b in self.my_chunks:
            la, lb = a - self.lo, b - self.lo
            ops = [d.P2POp(d.irecv, self.wp[la:lb], self.root, self.group), d.P2POp(d.irecv, self.tm[la:lb], self.root, self.group)]
            if self.per_bc:
                ops.append(d.P2POp(d.irecv, self.bc[la:lb], self.root, self.group))
            per_chunk.append(_p2p(ops))
        for ws in recvs:
            for w in ws:
                w.wait()
        sends = []
        for (a, b), ws in zip(self.my_chunks, per_chunk):
            la, lb = a - self.lo, b - self.lo
            for w in ws:
                w.wait()          # the compute stream waits for this chunk's inputs only
            self.solve(self.wp[la:lb], self.tm[la:lb], self.bc[la:lb] if self.per_bc else self.bc, out=self.co[la:lb])
            sends += _p2p([d.P2POp(d.isend, self.co[la:lb], self.root, self.group)])   # overlaps the next chunk's solve
        for w in sends:
            w.wait()
        return None


def solve_batch_from_root(waypoints, times, bc=None, order=4, path_weight=0.0, vel_zero_weight=0.0,
                          root=0, group=None, device=None, local_solve=None, batch=None, segments=None,
                          dtype=torch.float64, chunks=2, per_trajectory_bc=None):
    """One-shot form of RootPipeline.  The full batch lives on `root` (other ranks pass None and give `batch`,
    `segments` and, when the boundary conditions are per trajectory, per_trajectory_bc=True).  Returns
    [B,S,3,2o] on root, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == root:
        batch, segments = times.shape
        dtype = times.dtype
        device = device or times.device
        if per_trajectory_bc is None:
            per_trajectory_bc = bc is not None and bc.reshape(-1, 4, 3).shape[0] == batch and batch != 1
    device = device or torch.device("cpu")
    pipe = RootPipeline(None, batch, segments, order, device, chunks=chunks, dist=dist, rank=rank, world=world, group=group,
                        root=root, per_trajectory_bc=bool(per_trajectory_bc), dtype=dtype,
                        local_solve=local_solve or _default_local_solve(order, path_weight, vel_zero_weight))
    if rank == root:
        return pipe.run(waypoints.to(device), times.to(device), None if bc is None else bc.to(device))
    return pipe.run()


def ragged_partition(seg_offsets, world):
    """Trajectory boundaries [world+1] that balance the cumulative SEGMENT count (= bytes) over the ranks."""
    B = seg_offsets.numel() - 1
    total = int(seg_offsets[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r // world
        cuts.append(int(torch.searchsorted(seg_offsets, torch.tensor(target, dtype=seg_offsets.dtype), right=False)))
        cuts[-1] = min(max(cuts[-1], cuts[-2]), B)
    cuts.append(B)
    return cuts


def solve_ragged_from_root(waypoints, times, seg_offsets, bc=None, order=4, root=0, group=None, device=None,
                           local_solve=None, dtype=torch.float64):
    """Ragged batch on `root` (waypoints [sum(S_b)+B,3], times [sum S_b], seg_offsets [B+1]; bc None, shared
    [1,4,3] or per trajectory [B,4,3]); other ranks pass None.  Partition by cumulative segment count, scatter,
    solve, gather.  Returns the concatenated coefficients [sum S_b,3,2o] on root, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    device = device or (times.device if rank == root else torch.device("cpu"))
    m = 2 * int(order)
    # the split table: trajectory cuts, segment cuts, and whether bc is per trajectory
    table = torch.zeros(2 * (world + 1) + 1, dtype=torch.int64, device=device)
    if rank == root:
        off = seg_offsets.to(device=device, dtype=torch.int64)
        cuts = ragged_partition(off.cpu(), world)
        per_bc = bc is not None and bc.reshape(-1, 4, 3).shape[0] == off.numel() - 1 and off.numel() - 1 != 1
        table[:world + 1] = torch.tensor(cuts, dtype=torch.int64)
        table[world + 1:2 * (world + 1)] = off[torch.tensor(cuts)]
        table[-1] = int(per_bc)
    dist.broadcast(table, src=root, group=group)
    cuts, segcuts, per_bc = table[:world + 1].tolist(), table[world + 1:2 * (world + 1)].tolist(), bool(table[-1])
    solve = local_solve or _default_local_solve(order)
    if rank == root:
        bcr = torch.zeros((1, 4, 3), dtype=dtype, device=device) if bc is None else bc.to(device).reshape(-1, 4, 3)
        total = int(off[-1])
        out = torch.empty((total, 3, m), dtype=dtype, device=device)
        ops, keep = [], []
        for r in range(world):
            if r == root or cuts[r + 1] == cuts[r]:
                continue
            t0, t1, s0, s1 = cuts[r], cuts[r + 1], segcuts[r], segcuts[r + 1]
            loc = (off[t0:t1 + 1] - s0).contiguous()
            keep.append(loc)
            ops += [dist.P2POp(dist.isend, loc, r, group), dist.P2POp(dist.isend, waypoints[s0 + t0:s1 + t1], r, group),
                    dist.P2POp(dist.isend, times[s0:s1], r, group),
                    dist.P2POp(dist.isend, bcr[t0:t1] if per_bc else bcr, r, group),
                    dist.P2POp(dist.irecv, out[s0:s1], r, group)]
        pending = _p2p(ops)
        t0, t1, s0, s1 = cuts[root], cuts[root + 1], segcuts[root], segcuts[root + 1]
        if t1 > t0:
            solve(waypoints[s0 + t0:s1 + t1], times[s0:s1], bcr[t0:t1] if per_bc else bcr, out=out[s0:s1],
                  seg_offsets=(off[t0:t1 + 1] - s0).contiguous())
        for w in pending:
            w.wait()
        return out
    t0, t1, s0, s1 = cuts[rank], cuts[rank + 1], segcuts[rank], segcuts[rank + 1]
    if t1 == t0:
        return None
    nt, ns = t1 - t0, s1 - s0
    loc = torch.empty(nt + 1, dtype=torch.int64, device=device)
    wp = torch.empty((ns + nt, 3), dtype=dtype, device=device)
    tm = torch.empty(ns, dtype=dtype, device=device)
    bcl = torch.empty((nt if per_bc else 1, 4, 3), dtype=dtype, device=device)
    co = torch.empty((ns, 3, m), dtype=dtype, device=device)
    for w in _p2p([dist.P2POp(dist.irecv, loc, root, group), dist.P2POp(dist.irecv, wp, root, group),
                   dist.P2POp(dist.irecv, tm, root, group), dist.P2POp(dist.irecv, bcl, root, group)]):
        w.wait()
    solve(wp, tm, bcl, out=co, seg_offsets=loc)
    for w in _p2p([dist.P2POp(dist.isend, co, root, group)]):
        w.wait()
    return None
