import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/pytorch-a2c_amd"]
import torch
import bench
from a2c_amd.parallel import Shard
dev = torch.device("cuda", 0)
if os.environ.get("DEDICATED") == "1":
    torch.cuda.set_stream(torch.cuda.Stream())
wl = sys.argv[1] if len(sys.argv) > 1 else "conv"
b = bench.Bench(wl, None, "RMSprop", "host-pinned", "native", 4, Shard(), dev)
b.step()
print("eager ok", b.info)
b.capture()
print("ugraph", b.ugraph is not None)
for i in range(3):
    b.rollout()
    if os.environ.get('SYNC') == '1': torch.cuda.synchronize()
    d = b.D["dones"].reshape(b.n_envs, b.T)
    try:
        print(b.update())
    except Exception as e:
        print("update failed:", e)
        bufs = b.updater._bufs
        print("err flag", bufs["err"].item(), "udev", b.udev)
        # eager scan on the same data
        from a2c_amd import ops
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        a_, r_ = torch.empty(b.N, device=dev), torch.empty(b.N, device=dev)
        ops.gae_returns(b.D["deltas"], b.D["rewards"], b.D["dones"], .97, .99, b.n_envs, b.T, a_, r_, err=err)
        print("eager scan err", err.item())
        break
b.close()
