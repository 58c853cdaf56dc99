"""world-size-1 RCCL run with the collectives forced on: does the update capture into a hipGraph with the
all-reduces inside, and does the replay give the same numbers as the eager update?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, ROOT + "/pytorch-a2c_amd"]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                  A2C_FORCE_COLLECTIVES="1", A2C_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch
import bench
from a2c_amd.parallel import Shard
sh = Shard.from_env()
print("active", sh.active, torch.distributed.get_backend())
dev = torch.device("cuda", 0)
b = bench.Bench("a3c", 64, "RMSprop", "host-pinned", "native", 2, sh, dev)
b.step()
b.capture()
print("update graph captured:", b.ugraph is not None)
for i in range(3):
    print(b.step())
e, r, u = b.timed(10)
print("ms/step", 1e3 * e / 10, "rollout", r, "update", u)
b.close()
torch.distributed.destroy_process_group()
