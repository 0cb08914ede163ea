"""rocprofv3 --kernel-trace --stats -- python3 tools/sample_prof.py : the batched sampler's kernels (B=65536 x 16, order 4)."""
import importlib, sys, torch
sys.path.insert(0, ".")
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
B, S = 65536, 16
wp, _ = synth.make_batch(B, S, config_id=21)
d_wp = torch.from_numpy(wp * 4.0).cuda()
plan = csp.plan_batch(d_wp, 5.0, 0.1, order=4, vel_zero_weight=0.02)
bufs = csp.sample_batch(plan.times, plan.coeffs, 0.7, 256)
for _ in range(20):
    csp.sample_batch(plan.times, plan.coeffs, 0.7, 256, out=bufs)
torch.cuda.synchronize()
