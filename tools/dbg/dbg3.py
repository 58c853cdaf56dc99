import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/pytorch-a2c_amd"]
import torch
import bench
from a2c_amd.parallel import Shard
dev = torch.device("cuda", 0)
seq = sys.argv[1]
import a2c_amd.synthetic as syn
_orig = syn.TapeEnv.__init__
def _init(self, *a, **k):
    _orig(self, *a, **k)
    self.frames = (self.frames * int(os.environ.get("FRAMEVAL", "1"))).astype("uint8")
syn.TapeEnv.__init__ = _init
b = bench.Bench("conv", None, "RMSprop", "host-pinned", "native", 8, Shard(), dev)
b.step()
b.capture()
i = 0
for ch in seq:
    if ch == "Y":
        torch.cuda.synchronize(); print("sync")
    elif ch == "E":
        e = torch.cuda.Event(enable_timing=True); e.record(); print("event")
    elif ch == "S":
        b.rollout()
        d = b.D["dones"].reshape(b.n_envs, b.T)
        try:
            info = b.update()
            print("step", i, "ok loss", round(info["Loss"], 5))
        except Exception as ex:
            torch.cuda.synchronize()
            print("step", i, "FAILED", "last dones all 1:", bool((d[:, -1] == 1).all()), "host vec", b.udev.cpu().tolist(), hex(int(b.udev.cpu()[4])))
            break
        i += 1
b.close()
