import importlib, sys, torch
sys.path.insert(0, ".")
import bench
from tests import synth
csp = importlib.import_module("cs-pathplan_amd")
dev = torch.device("cuda", 0)
B, S, o, nb = 4096, 8, 4, 16
preps = []
for k in range(nb):
    wp, tm = synth.make_batch(B, S, config_id=2, offset=k * B)
    preps.append(csp.PreparedSolve(torch.from_numpy(wp).to(dev), torch.from_numpy(tm).to(dev), order=o))
for p in preps:
    p.run()
torch.cuda.synchronize()
ref = [p.out.clone() for p in preps]
for p in preps:
    p.out.zero_()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s):
        for p in preps:
            p.run(stream=s.cuda_stream)
torch.cuda.synchronize()
g.replay()
torch.cuda.synchronize()
print("graph replay bit-equal:", all(torch.equal(a, p.out) for a, p in zip(ref, preps)))
ms = bench.timed(g.replay, 100, 10, dev)
def loop():
    for p in preps: p.run()
ms2 = bench.timed(loop, 100, 10, dev)
print("graph: %.2f us per batch; plain launches: %.2f us per batch" % (ms * 1e3 / nb, ms2 * 1e3 / nb))
