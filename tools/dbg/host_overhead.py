"""host-side time of one epoch's enqueue calls (headline bench object): how long is the GPU idle while Python prepares launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd"), os.path.join(ROOT, "tests", "golden")]
import torch
import bench
from a2c_amd.parallel import Shard
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
b = bench.Bench("a3c", None, "RMSprop", "host-pinned", "native", 14, Shard(), dev, frame_store=True)
b.step(); b.capture(); b.step(); b.step()
tr, tu, tw = [], [], []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); b.rollout(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    b.update(); t3 = time.perf_counter()
    tr.append(t1 - t0); tw.append(t2 - t1); tu.append(t3 - t2)
import statistics as st
print("rollout enqueue (host) us: median %.0f min %.0f" % (1e6 * st.median(tr), 1e6 * min(tr)))
print("rollout GPU wait after enqueue us: median %.0f" % (1e6 * st.median(tw)))
print("update replay + read-back (host+GPU) us: median %.0f" % (1e6 * st.median(tu)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    b.rollout(); torch.cuda.synchronize(); b.update()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
b.close()
