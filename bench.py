#!/usr/bin/env python3
"""bench.py -- minimum-snap solves/sec on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (csp_minsnap_solve_batch through the C-ABI) over one
batch of synthetic trajectories already resident in HBM.  Workload = BASELINE config C3 per GPU:
B=65536 independent 3-axis trajectories, 16 segments, order 4 (degree 7), fp64.  Trajectories
are independent, so ranks shard the batch with no data-path collective (weak scaling: every
rank owns 65536 trajectories of one deterministic stream).

One JSON line on rank 0; `roofline` prices the dominant kernel against the 8 TB/s HBM peak by
ALGORITHMIC bytes (3608 B/solve at S=16,o=4,f64 -- SURVEY.md §8d); `cpu_baseline` times the CPU
oracle (a restatement of the reference's dense path, kind "port") on a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def bench_c5(args, csp, dev):
    """Side benchmark (not the driver's line): BASELINE config C5 -- mixed batch, S ~ U{4..64},
    order ~ U{3,4,5}, fp32 storage, bucketed by order and length class on the host BEFORE the timed
    region; one ragged call per bucket."""
    trajs = synth.make_ragged(args.batch)
    buckets = []
    total_bytes = 0
    mixed = _load_mixed(csp)
    for order in (3, 4, 5):
        sel = sorted((t for t in trajs if t[0] == order), key=lambda t: len(t[2]))
        all_lens = np.array([len(t[2]) for t in sel])
        for lo, hi in mixed.length_classes(all_lens):   # one ragged call per power-of-two length class
            sub, lens = sel[lo:hi], all_lens[lo:hi]
            wp = torch.from_numpy(np.concatenate([t[1] for t in sub]).astype(np.float32)).to(dev)
            tm = torch.from_numpy(np.concatenate([t[2] for t in sub]).astype(np.float32)).to(dev)
            off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)).to(dev)
            buckets.append(csp.PreparedSolve(wp, tm, order=order, seg_offsets=off, max_segments=int(lens.max())))
            total_bytes += sum(synth.algorithmic_bytes(int(n), order, 4) for n in lens)

    # a bucket alone cannot fill 1024 SIMDs: four HIP streams, buckets dealt round-robin, longest first
    buckets.sort(key=lambda ps: -ps.tm.numel())
    streams = [torch.cuda.Stream(device=dev) for _ in range(min(4, len(buckets)))]
    main = torch.cuda.current_stream(dev)

    def step():
        for st_ in streams:
            st_.wait_stream(main)
        for i, ps in enumerate(buckets):
            ps.run(streams[i % len(streams)].cuda_stream)
        for st_ in streams:
            main.wait_stream(st_)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    print(json.dumps({"workload": "C5 mixed ragged, fp32 storage / fp64 arithmetic", "batch": args.batch,
                      "solves_per_s": args.batch / (ms * 1e-3), "ms_per_step": ms,
                      "algorithmic_GBps": total_bytes / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": total_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU")
    ap.add_argument("--segments", type=int, default=16)
    ap.add_argument("--order", type=int, default=4)
    ap.add_argument("--path-weight", type=float, default=0.0, help="side benchmark: path-deviation penalty on (row A7)")
    ap.add_argument("--vel-zero-weight", type=float, default=0.0)
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU baseline work")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--segment-major", action="store_true",
                    help="experiment: CSP_FLAG_SEGMENT_MAJOR coefficient layout [S][B][3][2o]")
    ap.add_argument("--no-persistent", action="store_true", help="A/B: CSP_FLAG_NO_PERSISTENT")
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"], help="c5 = side benchmark of the mixed ragged fp32 path")
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
    if use_dist:
        import torch.distributed as dist
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    csp = importlib.import_module("cs-pathplan_amd")
    if args.workload == "c5":
        return bench_c5(args, csp, dev)
    B, S, o = args.batch, args.segments, args.order
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
    if use_dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    if rank == 0:
        bytes_per_solve = synth.algorithmic_bytes(S, o, 8)
        solves_per_s = world * B * args.steps / elapsed
        achieved = B * bytes_per_solve / (kernel_ms * 1e-3) / 1e9  # GB/s, one launch on one GPU
        traffic = None
        tr_file = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tr_file):
            try:
                tr = json.load(open(tr_file))
                if tr.get("kernel") == kernel and tr.get("batch") == B:
                    traffic = tr.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        res = {
            "metric": "minimum-snap solves/sec (16-seg, order-7, 3-axis)",
            "value": solves_per_s, "unit": "solves/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C3: B=%d trajectories/GPU x %d segments, order %d (degree %d), "
                                   "3 axes, fp64, zero boundary vel/acc, %s" % (B, S, o, 2 * o - 1,
                                       "penalties off" if args.path_weight == 0.0 and args.vel_zero_weight == 0.0 else
                                       "path_weight=%g vel_zero_weight=%g (side benchmark)" % (args.path_weight, args.vel_zero_weight)),
                       "batch_per_gpu": B, "segments": S, "order": o, "kernel": kernel,
                       "coeff_layout": "[S][B][3][2o] (CSP_FLAG_SEGMENT_MAJOR)" if args.segment_major else "[B][S][3][2o]",
                       "sharding": "independent trajectories per rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_solve": bytes_per_solve, "kernel": kernel,
                         "kernel_ms": kernel_ms},
        }
        if args.host_path and world == 1:
            csp.solve_batch(wp, tm, order=o, force_generic=args.force_generic)
            t1 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                csp.solve_batch(wp, tm, order=o, force_generic=args.force_generic)
            res["host_path_solves_per_s"] = B * reps / (time.perf_counter() - t1)
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
            res["parity_max_rel_err"] = synth.rel_err(got.cpu().numpy(), ref[:chk])
            import oracle
            nld = 128  # 80-bit long-double build of the oracle as the yardstick for both
            ld, _ = oracle.solve_batch(o, wp[:nld], tm[:nld], path_weight=args.path_weight, vel_zero_weight=args.vel_zero_weight,
                                       nthreads=oracle.max_threads(), long_double=True)
            res["parity_vs_long_double"] = {"hip": synth.rel_err(got[:nld].cpu().numpy(), ld),
                                            "cpu_port_fp64": synth.rel_err(ref[:nld], ld), "trajectories": nld}
            if args.path_weight == 0.0:
                # ... and for the "honest CPU" line: a structured CPU solver (oracle/structured_oracle.cpp: the same
                # block-tridiagonal LDL^T, plain loops, OpenMP) on the whole batch, all host threads
                buf = np.zeros((B, S, 3, 2 * o))
                _o.struct_solve_batch(o, wp, tm, vel_zero_weight=args.vel_zero_weight, nthreads=_o.max_threads(), out=buf)
                t2c = time.perf_counter()
                _o.struct_solve_batch(o, wp, tm, vel_zero_weight=args.vel_zero_weight, nthreads=_o.max_threads(), out=buf)
                res["cpu_structured"] = {"value": B / (time.perf_counter() - t2c), "unit": "solves/s", "cores": _o.max_threads(),
                                         "kind": "structured CPU solver (not the reference's algorithm)",
                                         "sample": "the whole timed batch", "max_rel_err_vs_hip": synth.rel_err(got.cpu().numpy(), buf[:chk])}
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
