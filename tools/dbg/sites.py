import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/pytorch-a2c_amd"]
import torch
import bench
from a2c_amd.parallel import Shard
from a2c_amd import ops
dev = torch.device("cuda", 0)
wl = sys.argv[1]
b = bench.Bench(wl, None, "RMSprop", sys.argv[2] if len(sys.argv) > 2 else "host-pinned", "native", 8, Shard(), dev, update_graph=False)
b.step(); b.step()
e, r, u = b.timed(3)
print(wl, "ms/step", round(1e3 * e / 3, 2), "rollout", round(r, 2), "update", round(u, 2))
summ = b.site_timers(2)
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])[:24]:
    print(f"  {k:36s} avg {v['avg_ms']:8.3f} ms x {v['launches'] // 2:4d} = {v['total_ms'] / 2:8.2f} ms")
print("  total sites", round(sum(v["total_ms"] for v in summ.values()) / 2, 2))
b.close()
