#!/usr/bin/env python3
"""bench.py -- minimum-snap solves/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (csp_minsnap_solve_batch through the C-ABI) over one
batch of synthetic trajectories already resident in HBM.  Workload = BASELINE config C3 per GPU:
B=65536 independent 3-axis trajectories, 16 segments, order 4 (degree 7), fp64.  Trajectories
are independent, so ranks shard the batch with no data-path collective (weak scaling: every
rank owns 65536 trajectories of one deterministic stream; `--scaling strong` divides ONE
65536-trajectory batch over the ranks instead).

One JSON line on rank 0; `roofline` prices the dominant kernel against the 8 TB/s HBM peak by
ALGORITHMIC bytes (3608 B/solve at S=16,o=4,f64 -- SURVEY.md §8d); `cpu_baseline` times the CPU
oracle (a restatement of the reference's dense path, kind "port") on a bounded sample.

Side records on the same line (N=1 only; `--no-side-records` skips them), each timed the same way
(HIP events on the launch stream, inputs resident in HBM):
  streaming     B=524288 (BASELINE C4's whole batch on one GPU: 1.9 GB per step, 7x the Infinity Cache)
  c2            BASELINE C2 (B=4096, S=8): the small-batch launch
  yaml_default  the reference's shipped configuration (minimum_snap_config.yaml:5-10: order 2,
                vel_zero_weight 0.01, path_weight 1e-7) at B=65536, S=16 -- the path-penalty kernel
  c2_multi      16 such batches per step in ONE csp_minsnap_solve_multi call / kernel launch
  sample        row N1: sampling + thinning + statistics of a resident B=65536 x 16 batch (csp_minsnap_sample_batch)
  single_altitude  row N4, the reference's own call: ONE pentadiagonal altitude problem (n = 2000 / 20000) on the GPU (block
                cyclic reduction) with the CPU oracle's banded Cholesky on one core beside it
  c5            BASELINE C5: mixed ragged batch, S~U{4..64}, order~U{3,4,5}, fp32 storage, ONE csp_minsnap_solve_mixed call
                (device-side bucketing inside the timed region)
  single_flight ONE flight (README uav31_0, the reference's own call pattern) through the C++ class shim,
                plan + sample, with the CPU oracle's GenerateTrajectoryMatrix time beside it
`--end-to-end` (N>=1) adds the root-scatter / solve / root-gather pipeline over RCCL.
"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# the translation units that define the headline kernel: roofline.traffic (PMC counters, collected by
# tools/profile_bench.sh) is only reported while their content hash equals the profiled one
HEADLINE_SOURCES = ("minsnap_fixed_impl.h", "minsnap_fixed.hip", "minsnap_fixed_o4a.hip", "minsnap_fixed_o4b.hip",
                    "minsnap_device.h", "minsnap_launch.h", "minsnap_tables.h")


def headline_source_sha():
    h = hashlib.sha256()
    for f in HEADLINE_SOURCES:
        with open(os.path.join(ROOT, "cs-pathplan_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_baseline(order, S, wp, tm, budget_s, pw=0.0, vw=0.0):
    """Oracle (dense restatement, kind 'port') on all host cores over a bounded sample."""
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    probe = min(4 * threads, wp.shape[0])
    t0 = time.perf_counter()
    oracle.solve_batch(order, wp[:probe], tm[:probe], path_weight=pw, vel_zero_weight=vw, nthreads=threads)
    per = (time.perf_counter() - t0) / probe
    n = int(min(wp.shape[0], max(probe, budget_s / max(per, 1e-9))))
    t0 = time.perf_counter()
    ref, _ = oracle.solve_batch(order, wp[:n], tm[:n], path_weight=pw, vel_zero_weight=vw, nthreads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "solves/s", "cores": threads, "kind": "port",
            "sample": "%d of the %d trajectories of the timed batch, dense LU restatement "
                      "(oracle/dense_oracle.c, -O2, OpenMP), %.1f s" % (n, wp.shape[0], dt)}, ref, n


def _load_mixed(csp):
    import importlib.util
    spec = importlib.util.spec_from_file_location("csp_mixed", os.path.join(os.path.dirname(csp.__file__), "mixed.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def timed(step, steps, warmup, dev):
    """ms per step by HIP events on torch's current stream (the launches go to that stream)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps


def bench_uniform(csp, dev, B, S, o, steps, warmup, config_id, pw=0.0, vw=0.0, label=""):
    """One uniform workload through PreparedSolve (one C-ABI call per step)."""
    wp, tm = synth.make_batch(B, S, config_id=config_id)
    d_wp, d_tm = torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev)
    prep = csp.PreparedSolve(d_wp, d_tm, order=o, path_weight=pw, vel_zero_weight=vw,
                             stream=torch.cuda.current_stream(dev).cuda_stream)
    ms = timed(prep.run, steps, warmup, dev)
    nbytes = B * synth.algorithmic_bytes(S, o, 8)
    gbps = nbytes / (ms * 1e-3) / 1e9
    rec = {"workload": label or "B=%d x %d segments, order %d, fp64" % (B, S, o), "batch": B, "segments": S, "order": o,
           "kernel": prep.kernel, "kernel_ms": ms, "solves_per_s": B / (ms * 1e-3), "algorithmic_bytes_per_launch": nbytes,
           "achieved_GBps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS, "steps": steps}
    if pw or vw:
        rec["path_weight"], rec["vel_zero_weight"] = pw, vw
    return rec, prep, wp, tm


def bench_c2_multi(csp, dev, nb=16):
    """BASELINE C2 as a planner meets it: `nb` independent 4096-trajectory batches per tick.  One csp_minsnap_solve_multi call =
    ONE kernel launch for all of them (the workgroups find their batch in a kernel-argument table) against one launch each."""
    B, S, o = 4096, 8, 4
    wps, tms = [], []
    for k in range(nb):
        wp, tm = synth.make_batch(B, S, config_id=2, offset=k * B)
        wps.append(torch.from_numpy(wp).to(dev))
        tms.append(torch.from_numpy(tm).to(dev))
    pm = csp.PreparedMulti(wps, tms, order=o)
    ms = timed(pm.run, 100, 10, dev)
    singles = [csp.PreparedSolve(w, t, order=o) for w, t in zip(wps, tms)]

    def loop():
        for ps in singles:
            ps.run()
    ms_loop = timed(loop, 100, 10, dev)
    torch.cuda.synchronize(dev)
    same = all(bool(torch.equal(a, ps.out)) for a, ps in zip(pm.out, singles))
    nbytes = nb * B * synth.algorithmic_bytes(S, o, 8)
    return {"workload": "%d independent batches of C2 (B=4096 x 8 segments, order 4, fp64) per step" % nb,
            "one_call_one_launch_us_per_batch": ms * 1e3 / nb, "one_launch_per_batch_us_per_batch": ms_loop * 1e3 / nb,
            "solves_per_s": nb * B / (ms * 1e-3), "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "bit_equal_to_single_launches": same}


def bench_sample(csp, dev, B=65536, S=16, order=4):
    """Row N1: the sampling half of GenerateTrajectoryMatrix (minimum_snap.cpp:97-205) for a resident batch --
    candidates at dt = min(0.1, T/10), sequential distance thinning, end-point rule, statistics.  Bytes = coefficients and
    times in + kept samples, counts and statistics out."""
    wp, _ = synth.make_batch(B, S, config_id=21)
    d_wp = torch.from_numpy(wp * 4.0).to(dev)
    plan = csp.plan_batch(d_wp, 5.0, 0.1, order=order, vel_zero_weight=0.02)
    cap = 256
    bufs = csp.sample_batch(plan.times, plan.coeffs, 0.7, cap)
    ms = timed(lambda: csp.sample_batch(plan.times, plan.coeffs, 0.7, cap, out=bufs), 20, 3, dev)
    samples, counts, stats = bufs
    kept = int(counts.sum().item())
    nbytes = B * S * (3 * 2 * order + 1) * 8 + kept * 24 + B * (4 + 16)
    return {"workload": "sampling + thinning + statistics of B=%d x %d segments, order %d, fp64 (sample_distance 0.7, capacity %d)" % (B, S, order, cap),
            "kernel_ms": ms, "trajectories_per_s": B / (ms * 1e-3), "mean_samples_kept": kept / B, "max_samples_kept": int(counts.max().item()),
            "algorithmic_bytes_per_launch": nbytes, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}


def bench_single_altitude(csp, dev):
    """Row N4 for the reference's own call pattern: ONE pentadiagonal altitude problem per plan (optimizeHeights,
    uavPathPlanning.cpp:1575-1713, solve at :1670-1676) -- block cyclic reduction in one workgroup -- with the CPU oracle's
    banded Cholesky on one core beside it.  GPU times: the kernel on resident data (HIP events) and the whole host-memory
    C-ABI call (staging included)."""
    import oracle
    rng = np.random.default_rng(3)
    out = {"workload": "one optimizeHeights problem (lambda_smooth 1, lambda_follow 0.5, safe_distance 50, max_climb_rate 2)"}
    for n in (2000, 20000):
        xy = np.cumsum(rng.uniform(20, 60, size=(n, 2)), axis=0)
        z = 100 + np.cumsum(rng.normal(0, 8, n))
        elev = 80 + 10 * np.sin(np.arange(n) / 5.0) + rng.normal(0, 2, n)
        xyz = np.column_stack([xy, z])
        off = np.array([0, n], dtype=np.int64)
        got = csp.alt_optimize_heights_batch(xyz, elev, off, 1.0, 0.5, 50.0, 2.0)
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            csp.alt_optimize_heights_batch(xyz, elev, off, 1.0, 0.5, 50.0, 2.0)
        host_us = (time.perf_counter() - t0) / reps * 1e6
        d = [torch.from_numpy(x).to(dev) for x in (xyz, elev, off)]
        ms = timed(lambda: csp.alt_optimize_heights_batch(d[0], d[1], d[2], 1.0, 0.5, 50.0, 2.0), 20, 3, dev)
        ref = oracle.alt_optimize(xyz, elev, 1.0, 0.5, 50.0, 2.0, banded=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.alt_optimize(xyz, elev, 1.0, 0.5, 50.0, 2.0, banded=True)
        cpu_us = (time.perf_counter() - t0) / reps * 1e6
        out["n_%d" % n] = {"gpu_host_call_us": host_us, "gpu_device_call_us": ms * 1e3, "cpu_banded_cholesky_one_core_us": cpu_us,
                           "max_rel_err_vs_cpu": float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))}
    return out


def bench_c5(csp, dev, batch, steps, warmup):
    """BASELINE config C5 -- mixed batch, S ~ U{4..64}, order ~ U{3,4,5}, fp32 storage -- through ONE csp_minsnap_solve_mixed call
    per step: the bucketing by (order, length class) runs on the device INSIDE the timed region, and the coefficients land in
    the caller's order (no un-permute exists to be left out)."""
    trajs = synth.make_ragged(batch)
    mixed = _load_mixed(csp)
    plan = mixed.MixedBatch(csp, trajs, dev, dtype=torch.float32)
    ms = timed(plan.run, steps, warmup, dev)
    return {"workload": "C5 mixed ragged: S~U{4..64}, order~U{3,4,5}, fp32 storage / fp64 arithmetic; one csp_minsnap_solve_mixed call per "
                        "step, device-side bucketing (histogram, scan, scatter) and caller-order output INCLUDED in the timed region",
            "batch": batch, "launches_per_step": "3 bucketing kernels + one persistent launch for all orders (max_segments <= 64)", "kernels": plan.kernels,
            "kernel_ms": ms, "solves_per_s": batch / (ms * 1e-3), "algorithmic_bytes_per_launch": plan.algorithmic_bytes,
            "achieved_GBps": plan.algorithmic_bytes / (ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": plan.algorithmic_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "steps": steps}


def single_flight():
    """ONE flight, the reference's own call pattern (UavPathPlanner::getPlan -> Minisnap_3D -> GenerateTrajectoryMatrix,
    uavPathPlanning.cpp:3684, :4461): README uav31_0 waypoints, the shipped yaml, leader_speed 200 -- through the C++
    class shim (host memory, B = 1), with the CPU oracle's GenerateTrajectoryMatrix restatement on one core beside it."""
    exe = os.path.join(ROOT, "cs-pathplan_amd", "host", "shim_selftest")
    if not os.path.exists(exe):
        return {"error": "cs-pathplan_amd/host/shim_selftest not built (run __graft_entry__.build())"}
    import oracle
    import tempfile
    out = {"workload": "README uav31_0 (7 waypoints, 6 segments), plan (time allocation + re-solve loop) + sampling, B=1, host memory"}
    for name, cfg in (("yaml_order2", dict(order=2, pw=1e-7, vw=0.01, V=200.0, mt=1.0, sd=300.0)),
                      ("order4", dict(order=4, pw=0.0, vw=0.0, V=200.0, mt=1.0, sd=300.0))):
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as fh:
            fh.write("%d %.17g %.17g %.17g %.17g %.17g %d\n" % (cfg["order"], cfg["pw"], cfg["vw"], cfg["V"], cfg["mt"], cfg["sd"], 7))
            for p in synth.README_UAV31_ENU:
                fh.write("%.17g %.17g %.17g\n" % tuple(p))
            path = fh.name
        r = subprocess.run([exe, "time3d", path, "200"], capture_output=True, text=True)
        os.unlink(path)
        rec = {}
        if r.returncode == 0:
            rec = json.loads(r.stdout.strip().splitlines()[-1])
        else:
            rec = {"error": "shim_selftest rc=%d %s" % (r.returncode, r.stderr[-200:])}
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            s, info = oracle.generate_trajectory(synth.README_UAV31_ENU, order=cfg["order"], path_weight=cfg["pw"],
                                                 vel_zero_weight=cfg["vw"], v_avg=cfg["V"], min_time_s=cfg["mt"], sample_distance=cfg["sd"])
        rec["cpu_oracle_us"] = (time.perf_counter() - t0) / reps * 1e6
        rec["cpu_oracle_samples"] = int(len(s))
        out[name] = rec
    return out


def end_to_end(args, csp, dev, dist, rank, world, share):
    """the batch lives on the ROOT GPU; per step the root scatters every rank's inputs (536 B/solve),
    every rank solves, and the root gathers the coefficients (3072 B/solve) -- chunked so that the gather of chunk i
    overlaps the solve of chunk i+1 (cs-pathplan_amd/sharding.py: grouped send/recv = RCCL over xGMI)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("csp_sharding", os.path.join(os.path.dirname(csp.__file__), "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    B, S, o = args.batch, args.segments, args.order
    total = B * world if args.scaling == "weak" else B
    if rank == 0:
        wp, tm = synth.make_batch(total, S, config_id=3)
        d_wp, d_tm = torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev)
    else:
        d_wp = d_tm = None
    if share and world > 1:
        raise SystemExit("--end-to-end needs one GPU per rank (gloo has no device-memory send/recv)")
    pipe = sh.RootPipeline(csp, total, S, o, dev, chunks=args.e2e_chunks, dist=dist if world > 1 else None, rank=rank, world=world)
    out = None
    for _ in range(args.warmup):
        out = pipe.run(d_wp, d_tm)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.run(d_wp, d_tm)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    rec = None
    if rank == 0:
        chk = min(total, 4096)
        ref = csp.solve_batch(d_wp[:chk].contiguous(), d_tm[:chk].contiguous(), order=o).coeffs
        tail = csp.solve_batch(d_wp[total - chk:].contiguous(), d_tm[total - chk:].contiguous(), order=o).coeffs
        rec = {"mode": "root scatter -> solve -> root gather, %d chunks per rank, gather of chunk i overlaps solve of chunk i+1" % pipe.chunks,
               "total_batch": total, "solves_per_s": total * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
               "bytes_scattered_per_step": (total - pipe.local_count) * 8 * (3 * (S + 1) + S),
               "bytes_gathered_per_step": (total - pipe.local_count) * 3 * S * 2 * o * 8,
               "bit_equal_to_one_device": bool(torch.equal(out[:chk], ref) and torch.equal(out[total - chk:], tail))}
    return rec


def end_to_end_cabi(args, csp, dev):
    """The same root-scatter / solve / root-gather pipeline INSIDE the C-ABI (csp_minsnap_solve_batch_sharded, CSP_MEM_DEVICE:
    single-process RCCL communicators, grouped send/recv, chunked overlap), from this one process over args.e2e_cabi devices."""
    n = args.e2e_cabi
    if n > csp.device_count():
        return {"error": "%d devices asked for, %d visible" % (n, csp.device_count())}
    S, o = args.segments, args.order
    total = args.batch * n if args.scaling == "weak" else args.batch
    wp, tm = synth.make_batch(total, S, config_id=3)
    d_wp, d_tm = torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev)
    out = torch.empty((total, S, 3, 2 * o), dtype=torch.float64, device=dev)
    for _ in range(max(1, args.warmup)):
        csp.solve_batch(d_wp, d_tm, order=o, out=out, ngpu=n)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        csp.solve_batch(d_wp, d_tm, order=o, out=out, ngpu=n)     # synchronous
    elapsed = time.perf_counter() - t0
    chk = min(total, 4096)
    ref = csp.solve_batch(d_wp[:chk].contiguous(), d_tm[:chk].contiguous(), order=o).coeffs
    tail = csp.solve_batch(d_wp[total - chk:].contiguous(), d_tm[total - chk:].contiguous(), order=o).coeffs
    torch.cuda.synchronize(dev)
    return {"mode": "csp_minsnap_solve_batch_sharded, CSP_MEM_DEVICE, %d device(s), one process" % n, "total_batch": total,
            "solves_per_s": total * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
            "bit_equal_to_one_device": bool(torch.equal(out[:chk], ref) and torch.equal(out[total - chk:], tail)),
            "verified_on_more_than_one_gpu": n > 1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU (weak) / in total (strong)")
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--path-weight", type=float, default=0.0, help="side benchmark: path-deviation penalty on (row A7)")
    ap.add_argument("--vel-zero-weight", type=float, default=0.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-records", action="store_true", help="only the headline workload (profiling runs)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--segment-major", action="store_true",
                    help="experiment: CSP_FLAG_SEGMENT_MAJOR coefficient layout [S][B][3][2o]")
    ap.add_argument("--no-persistent", action="store_true", help="A/B: CSP_FLAG_NO_PERSISTENT")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"], help="c5 = only the mixed ragged fp32 side benchmark")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch trajectories per rank; strong: --batch trajectories in total, divided over the ranks")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time the root-scatter / solve / root-gather pipeline (SURVEY.md 8e) and report it as `end_to_end`")
    ap.add_argument("--e2e-chunks", type=int, default=4)
    ap.add_argument("--e2e-cabi", type=int, default=0, metavar="NGPU",
                    help="single process only (no torchrun): also time csp_minsnap_solve_batch_sharded with a device-resident batch over "
                         "NGPU devices (RCCL scatter / solve / gather inside the C-ABI) and report it as `end_to_end_cabi`")
    ap.add_argument("--host-path", action="store_true",
                    help="also time the CSP_MEM_HOST boundary (PCIe-inclusive; reported as a side note, never `value`)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # Rehearsal switch for a ONE-GPU box (never set by the driver): CSP_BENCH_SHARE_GPU=1 lets N ranks
    # exercise the multi-rank control flow (sharded inputs, barrier, MAX-reduce) on cuda:0 over gloo.
    share = os.environ.get("CSP_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # CSP_BENCH_FORCE_DIST=1 (rehearsal only) makes a single rank go through the RCCL init / barrier /
    # all-reduce path too, so the N>1 control flow can be exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("CSP_BENCH_FORCE_DIST") == "1"
    dist = None
    if use_dist:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    csp = importlib.import_module("cs-pathplan_amd")
    if args.workload == "c5":
        print(json.dumps(bench_c5(csp, dev, args.batch, args.steps, args.warmup)))
        return
    S, o = args.segments, args.order
    if args.scaling == "strong":   # ONE batch of --batch trajectories, contiguous balanced chunks
        lo, hi = args.batch * rank // world, args.batch * (rank + 1) // world
        B = hi - lo
        wp, tm = synth.make_batch(B, S, config_id=3, offset=lo)
    else:
        B = args.batch
        wp, tm = synth.make_batch(B, S, config_id=3, offset=rank * B)
    d_wp, d_tm = torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev)
    d_bc = torch.zeros((1, 4, 3), dtype=torch.float64, device=dev)
    # descriptor, buffers and workspace are fixed for the run: a step is exactly one C-ABI call
    prep = csp.PreparedSolve(d_wp, d_tm, d_bc, order=o, path_weight=args.path_weight, vel_zero_weight=args.vel_zero_weight,
                             force_generic=args.force_generic,
                             segment_major=args.segment_major, no_persistent=args.no_persistent,
                             stream=torch.cuda.current_stream(dev).cuda_stream)
    out, kernel = prep.out, prep.kernel

    def step():
        prep.run()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    total_units = args.batch * world if args.scaling == "weak" else args.batch
    if use_dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    e2e = end_to_end(args, csp, dev, dist if use_dist else None, rank, world, share) if args.end_to_end else None

    if rank == 0:
        bytes_per_solve = synth.algorithmic_bytes(S, o, 8)
        solves_per_s = total_units * args.steps / elapsed
        achieved = B * bytes_per_solve / (kernel_ms * 1e-3) / 1e9  # GB/s, one launch on one GPU
        traffic, provenance = None, None
        tr_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tr_file):
            try:
                tr = json.load(open(tr_file))
                now = headline_source_sha()
                provenance = {"file": "profiles/traffic_latest.json", "profile_dir": tr.get("profile_dir"), "kernel": tr.get("kernel"),
                              "batch": tr.get("batch"), "source_sha_profiled": tr.get("source_sha"), "source_sha_now": now,
                              "kernel_avg_ns_rocprof": tr.get("kernel_avg_ns_rocprof")}
                fresh = tr.get("kernel") == kernel and tr.get("batch") == B and tr.get("source_sha") == now
                provenance["stale"] = not fresh
                if fresh:   # PMC counters cannot be read inside this run; a profile of another kernel/batch/source is not reported
                    traffic = tr.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "minimum-snap solves/sec (16-seg, order-7, 3-axis)",
            "value": solves_per_s, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: B=%d trajectories%s x %d segments, order %d (degree %d), "
                                   "3 axes, fp64, zero boundary vel/acc, %s" % (args.batch, "/GPU" if args.scaling == "weak" else " in total", S, o, 2 * o - 1,
                                       "penalties off" if args.path_weight == 0.0 and args.vel_zero_weight == 0.0 else
                                       "path_weight=%g vel_zero_weight=%g (side benchmark)" % (args.path_weight, args.vel_zero_weight)),
                       "batch_per_gpu": B, "segments": S, "order": o, "kernel": kernel,
                       "coeff_layout": "[S][B][3][2o] (CSP_FLAG_SEGMENT_MAJOR)" if args.segment_major else "[B][S][3][2o]",
                       "sharding": "independent trajectories per rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_provenance": provenance,
                         "algorithmic_bytes_per_solve": bytes_per_solve, "kernel": kernel,
                         "kernel_ms": kernel_ms,
                         "mfma_pct": 0.0,
                         "mfma_note": "the kernel issues no MFMA (SQ_INSTS_VALU_MFMA_F64 = 0 in profiles/*/summary.txt): one trajectory per "
                                      "lane, 3x3 / 4x4 blocks; v_mfma_f64_16x16x4 has the fp64 VALU's FLOP rate on gfx950 and would need "
                                      "cross-lane transposes (DESIGN.md 5.1) -- the bound is HBM, not the matrix cores"},
        }
        if e2e is not None:
            res["end_to_end"] = e2e
        if args.e2e_cabi and world == 1:
            res["end_to_end_cabi"] = end_to_end_cabi(args, csp, dev)
        if args.host_path and world == 1:
            # PCIe-inclusive rate from PAGEABLE caller memory; the result array is allocated and touched once, outside
            # the timed region (numpy's allocation + first-touch faults of a 201 MB array cost more than the transfer)
            h_out = csp.solve_batch(wp, tm, order=o, force_generic=args.force_generic).coeffs
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                csp.solve_batch(wp, tm, order=o, force_generic=args.force_generic, out=h_out)
            res["host_path_solves_per_s"] = B * reps / (time.perf_counter() - t1)
        if world == 1 and not args.no_side_records:
            side_steps = max(5, min(args.steps, 20))
            try:
                res["streaming"], p_, _, _ = bench_uniform(csp, dev, 524288, 16, 4, side_steps, 3, 4,
                                                           label="C4's whole batch on one GPU: B=524288 x 16 segments, order 4, fp64 (1.9 GB per step)")
                del p_
                torch.cuda.empty_cache()
                res["c2"], p_, _, _ = bench_uniform(csp, dev, 4096, 8, 4, 200, 20, 2, label="C2: B=4096 x 8 segments, order 4, fp64")
                res["c2_multi"] = bench_c2_multi(csp, dev)
                res["sample"] = bench_sample(csp, dev)
                res["single_altitude"] = bench_single_altitude(csp, dev)
                res["yaml_default"], p_, _, _ = bench_uniform(
                    csp, dev, 65536, 16, 2, side_steps, 3, 3, pw=1e-7, vw=0.01,
                    label="shipped yaml (minimum_snap_config.yaml:5-10): order 2, vel_zero_weight 0.01, path_weight 1e-7; B=65536 x 16 segments")
                res["yaml_default_order4"], p_, _, _ = bench_uniform(
                    csp, dev, 65536, 16, 4, side_steps, 3, 3, pw=1e-7, vw=0.01,
                    label="the yaml's penalties at order 4: B=65536 x 16 segments")
                del p_
                res["c5"] = bench_c5(csp, dev, 65536, side_steps, 3)
            except Exception as e:   # a side record never takes the headline down
                res["side_record_error"] = repr(e)
        if world == 1 and not args.no_cpu_baseline:
            cb, ref, n = cpu_baseline(o, S, wp, tm, args.cpu_budget, args.path_weight, args.vel_zero_weight)
            res["cpu_baseline"] = cb
            # SURVEY.md 8(d) also asks for the one-core figure: a short sample of the same batch
            import oracle as _o
            n1 = min(48, wp.shape[0])
            t1c = time.perf_counter()
            _o.solve_batch(o, wp[:n1], tm[:n1], path_weight=args.path_weight, vel_zero_weight=args.vel_zero_weight, nthreads=1)
            res["cpu_baseline_one_core"] = {"value": n1 / (time.perf_counter() - t1c), "unit": "solves/s", "cores": 1, "kind": "port",
                                            "sample": "%d trajectories of the timed batch" % n1}
            chk = min(n, 1024)
            got = out.view(S, B, 3, 2 * o).permute(1, 0, 2, 3)[:chk] if args.segment_major else out[:chk]
            got_np = got.cpu().numpy()
            # the gate that means something: every coefficient power against its own magnitude (tests/synth.py);
            # `parity_max_rel_err` is the norm-wise figure of SURVEY.md 8d, dominated by the constant term
            res["parity_per_power_rel_err"] = synth.rel_err_per_power(got_np, ref[:chk])
            res["parity_max_rel_err"] = synth.rel_err(got_np, ref[:chk])
            import oracle
            nld = 128  # 80-bit long-double build of the oracle as the yardstick for both
            ld, _ = oracle.solve_batch(o, wp[:nld], tm[:nld], path_weight=args.path_weight, vel_zero_weight=args.vel_zero_weight,
                                       nthreads=oracle.max_threads(), long_double=True)
            res["parity_vs_long_double"] = {"hip": synth.rel_err(got_np[:nld], ld), "cpu_port_fp64": synth.rel_err(ref[:nld], ld),
                                            "hip_per_power": synth.rel_err_per_power(got_np[:nld], ld),
                                            "cpu_port_fp64_per_power": synth.rel_err_per_power(ref[:nld], ld), "trajectories": nld}
            res["parity_note"] = "parity UNPINNED w.r.t. the real Eigen build: the oracle is a restatement (DESIGN.md 3)"
            if args.path_weight == 0.0:
                # ... and for the "honest CPU" line: a structured CPU solver (oracle/structured_oracle.cpp: the same
                # block-tridiagonal LDL^T, plain loops, OpenMP) on the whole batch, all host threads
                buf = np.zeros((B, S, 3, 2 * o))
                _o.struct_solve_batch(o, wp, tm, vel_zero_weight=args.vel_zero_weight, nthreads=_o.max_threads(), out=buf)
                t2c = time.perf_counter()
                _o.struct_solve_batch(o, wp, tm, vel_zero_weight=args.vel_zero_weight, nthreads=_o.max_threads(), out=buf)
                res["cpu_structured"] = {"value": B / (time.perf_counter() - t2c), "unit": "solves/s", "cores": _o.max_threads(),
                                         "kind": "structured CPU solver (not the reference's algorithm)",
                                         "sample": "the whole timed batch", "max_rel_err_vs_hip": synth.rel_err(got_np, buf[:chk])}
            if not args.no_side_records:
                try:
                    res["single_flight"] = single_flight()
                except Exception as e:
                    res["single_flight"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
