"""bench.py's own plumbing on the GPU box: the PreparedSolve fast path, the JSON contract, and the
multi-rank control flow (two ranks rehearsed on ONE GPU over gloo -- CSP_BENCH_SHARE_GPU=1; the real
N>1 run over RCCL is the driver's)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_prepared_solve_equals_solve_batch(csp):
    import torch
    wp, tm = synth.make_batch(3000, 16, config_id=3)
    d_wp, d_tm = torch.from_numpy(wp).cuda(), torch.from_numpy(tm).cuda()
    ref = csp.solve_batch(d_wp, d_tm, order=4).coeffs
    prep = csp.PreparedSolve(d_wp, d_tm, order=4)
    for _ in range(3):
        out = prep.run()
    torch.cuda.synchronize()
    assert prep.kernel == "fixed_o4_s16_f64"
    assert torch.equal(out, ref)


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_json_contract_single_gpu():
    d = _run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--batch", "8192", "--cpu-budget", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-12
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["parity_per_power_rel_err"] < 5e-8 and d["parity_max_rel_err"] < 1e-8   # the gate is the per-power figure


def test_bench_two_ranks_share_one_gpu():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2",
              "--batch", "8192"], env={"CSP_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and "cpu_baseline" not in d


def test_bench_rccl_control_flow_on_one_rank():
    """The driver's N>1 runs go through RCCL (backend "nccl"): init with a device id, barrier inside the
    timed region, MAX all-reduce of the timings, teardown.  One rank under torch.distributed.run with
    CSP_BENCH_FORCE_DIST=1 takes exactly that path on this one-GPU box."""
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                "--master-addr", "127.0.0.1", "--master-port", "29547",
                "bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
               env={"CSP_BENCH_FORCE_DIST": "1"})
    assert res["n_gpus"] == 1 and res["value"] > 1e8
    assert res["roofline"]["kernel"] == "fixed_o4_s16_f64"


def test_bench_strong_scaling_divides_one_batch():
    """--scaling strong: ONE batch of --batch trajectories cut into contiguous balanced chunks over the ranks (two ranks
    rehearsed on one GPU over gloo)."""
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
              "127.0.0.1", "--master-port", "29561", "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2",
              "--batch", "8192", "--scaling", "strong"], env={"CSP_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["batch_per_gpu"] == 4096 and d["value"] > 0


def test_bench_end_to_end_pipeline_on_one_rank():
    """--end-to-end through the RCCL-initialised path with one rank (CSP_BENCH_FORCE_DIST=1): the root pipeline degenerates to
    the chunked local solve and must reproduce the one-call result bit for bit; the record carries the byte counts."""
    res = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                "--master-addr", "127.0.0.1", "--master-port", "29563",
                "bench.py", "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-side-records",
                "--end-to-end", "--e2e-chunks", "3", "--batch", "10000"],
               env={"CSP_BENCH_FORCE_DIST": "1"})
    e = res["end_to_end"]
    assert e["bit_equal_to_one_device"] is True and e["total_batch"] == 10000 and e["solves_per_s"] > 0
    assert e["bytes_scattered_per_step"] == 0 and e["bytes_gathered_per_step"] == 0   # one rank: nothing travels
