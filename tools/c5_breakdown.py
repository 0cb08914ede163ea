"""Per-bucket timing of the C5 mixed batch (bench.py's `c5` record): each (order, length class) bucket alone,
then all of them the way MixedBatch.run() issues them.    python tools/c5_breakdown.py [batch]"""
import importlib
import importlib.util
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tests import synth
import bench

csp = importlib.import_module("cs-pathplan_amd")
spec = importlib.util.spec_from_file_location("csp_mixed", os.path.join(os.path.dirname(csp.__file__), "mixed.py"))
mixed = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mixed)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
mb = mixed.MixedBatch(csp, synth.make_ragged(B), dev, dtype=torch.float32)
tot = 0.0
for ps in mb.buckets:
    ms = bench.timed(ps.run, 20, 3, dev)
    lens = (ps.off[1:] - ps.off[:-1]).cpu().numpy()
    segs = int(lens.sum())
    o = ps.desc.order
    nbytes = 4 * (3 * (segs + len(lens)) + segs) + 4 * 3 * segs * 2 * o
    tot += ms
    print(json.dumps({"kernel": ps.kernel, "trajectories": len(lens), "segments": segs, "us": round(ms * 1e3, 1),
                      "Gseg_per_s": round(segs / ms / 1e6, 2), "GBps": round(nbytes / ms / 1e6, 1)}), flush=True)
print(json.dumps({"sum_of_buckets_us": round(tot * 1e3, 1), "together_us": round(bench.timed(mb.run, 20, 3, dev) * 1e3, 1),
                  "algorithmic_MB": round(mb.algorithmic_bytes / 1e6, 1)}))
