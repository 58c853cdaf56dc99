#!/usr/bin/env python3
"""Host side of the ingest alone (no GPU): native env threads stepping tape envs behind an UNREGISTERED pool region;
this process plays the device's part (posts the actions of all envs, waits for all rec granules).  Prints the time per
lock-step round and the implied service time per env step and thread.
    python tools/host_service_bench.py [n_envs] [tape_len]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "pytorch-a2c_amd")]
import numpy as np  # noqa: E402
from a2c_amd.hostpool import ROLLOUT, ThreadEnvPool  # noqa: E402
from a2c_amd.synthetic import TapeEnv  # noqa: E402

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 33
for bits in (False, True):
    for nthr in (4, 8, 12, 16):
        envs = [TapeEnv(env_id=j, length=L) for j in range(n_envs)]
        pool = ThreadEnvPool.from_tape_envs(envs, n_threads=nthr, register=False, pong=True, frame_bits=bits)
        pool.start()
        pool.set_phase(ROLLOUT)
        acts = np.zeros(n_envs, np.int64)
        pool.wait_frames(0)
        K = 2000
        for k in range(200):
            pool.post_actions(acts, seq=k)
            pool.wait_frames(k + 1)
        t0 = time.perf_counter()
        for k in range(200, 200 + K):
            pool.post_actions(acts, seq=k)
            pool.wait_frames(k + 1)
        dt = time.perf_counter() - t0
        print(f"transport={'bits' if bits else 'u8'} envs={n_envs} tape={L} threads={nthr}: {dt / K * 1e6:.2f} us per round, "
              f"{dt / K / n_envs * nthr * 1e6:.3f} us per env-step per thread", flush=True)
        pool.close()
