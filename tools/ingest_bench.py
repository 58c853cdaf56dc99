#!/usr/bin/env python3
"""Rollout-only timing of the host-pinned ingest (A3CModel, synthetic TapeEnv workers):
    python tools/ingest_bench.py [n_envs] [workers,workers,...] [zero-copy|memcpy] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd")]
import torch
import a2c_amd
from a2c_amd.hostpool import ProcessEnvPool
from a2c_amd.runner import Runner
from a2c_amd.synthetic import TapeEnv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
workers = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "8,12,14").split(",")]
ingest = sys.argv[3] if len(sys.argv) > 3 else "zero-copy"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
T, A, ss = 128, 3, (4, 84, 84)
hyps = dict(gamma=.99, lambda_=.98, n_tsteps=T, n_rollouts=B, n_envs=B, n_frame_stack=4, action_shift=0, render=False,
            env_type="Pong-synthetic", use_bptt=False, use_nstep_rets=False, norm_advs=True, entr_coef=.005, pi_coef=1.0,
            val_coef=.5, max_norm=.5, lr=1e-4, optim_type="RMSprop", is_discrete=True, h_size=256)
dev = "cuda"
torch.manual_seed(0)
net = a2c_amd.models.A3CModel(list(ss), A, h_size=256)
N = B * T
D = dict(states=torch.zeros(N, *ss, device=dev), deltas=torch.zeros(N, device=dev), rewards=torch.zeros(N, device=dev),
         dones=torch.zeros(N, device=dev), actions=torch.zeros(N, dtype=torch.int64, device=dev))
from a2c_amd.hostpool import ThreadEnvPool
native = os.environ.get("NATIVE", "1") == "1"
for W in workers:
    if native:
        pool = ThreadEnvPool.from_tape_envs([TapeEnv(env_id=j, length=33) for j in range(B)], n_threads=W, pong=True)
    else:
        pool = ProcessEnvPool(TapeEnv, B, env_kwargs=[dict(env_id=j, length=T + 1) for j in range(B)], n_workers=W, pong=True)
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest)
    try:
        t0 = time.perf_counter()
        r.rollout(net, list(range(B)), hyps)
        r.finish()
        t_first = time.perf_counter() - t0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{ingest} envs={B} workers={W}: rollout median {med * 1e3:.2f} ms  min {ts[0] * 1e3:.2f}  max {ts[-1] * 1e3:.2f} "
              f"({med / T * 1e6:.1f} us/step; first {t_first * 1e3:.1f} ms); H2D {B * 7056 * T / med / 1e9:.1f} GB/s", flush=True)
    finally:
        r.close()
