#!/usr/bin/env python3
"""Phase stamps of the persistent ring rollout kernel (a3c_ring_kernel), headline config, native env threads:
    python tools/ring_timing.py [n_workers] [transport]
prints, for workgroup 0 (wave 1's view), the average microseconds per env step spent in each phase."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd"), os.path.join(ROOT, "tests", "golden")]
import torch  # noqa: E402

import bench  # noqa: E402
from a2c_amd import _lib  # noqa: E402
from a2c_amd.parallel import Shard  # noqa: E402

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 12
transport = sys.argv[2] if len(sys.argv) > 2 else "bits"
frame_store = len(sys.argv) > 3 and sys.argv[3] == "frame_store"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
b = bench.Bench("a3c", None, "RMSprop", "host-pinned", "native", nw, Shard(), dev, transport=transport, frame_store=frame_store)
b.step(); b.capture(); b.step()
lib = _lib.load()
lib.a2c_debug_ring_timing.argtypes = [ctypes.c_void_p]
buf = torch.zeros(10, dtype=torch.int64, device=dev)
lib.a2c_debug_ring_timing(buf.data_ptr())
names = ["wait for the env worker (poll)", "barrier after the poll", "frame over PCIe -> ring (+barrier)", "conv1 newest plane (+barrier)",
         "conv2 + epilogue (2 barriers)", "heads (2 barriers)", "partial sums of the next state", "state row + stash stores (issue)", "TURN-AROUND cmd store -> rec seen",
         "polls per step (count x100)"]
acc = torch.zeros(10, dtype=torch.float64)
n = 10
for _ in range(n):
    b.rollout()
    torch.cuda.synchronize()
    acc += buf.cpu().double()
    b.update()
lib.a2c_debug_ring_timing(None)
e, r_ms, u_ms = b.timed(20)
per = acc / n / (b.T + 1) / 100.0          # 100 MHz ticks -> us per step
out = {k: round(float(v), 3) for k, v in zip(names, per)}
out["sum_us_per_step"] = round(float(per[:8].sum()), 3)
out["rollout_ms_timed"] = round(r_ms, 3)
out["update_ms_timed"] = round(u_ms, 3)
out["env_threads"] = nw
out["transport"] = transport
out["frame_store_lazy_states"] = frame_store
print(json.dumps(out, indent=1))
b.close()
