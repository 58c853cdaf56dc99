#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): time the imported reference's Runner.rollout / Updater.update_model beside
the oracle's SlotRunner.rollout / OracleUpdater.update_model on the SAME inputs, to show that bench.py's
`cpu_baseline.kind = "port"` costs what the reference costs.  Prints a markdown table (committed into
BASELINE.md); nothing here is imported by tests, bench.py or the product.

    python tools/time_oracle_vs_reference.py [--threads 8]
"""
import argparse
import os
import queue
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import make_golden as MG  # noqa: E402  (loads the reference by file path under stubs)
from cases import base_hyps, hashf, synth_shared  # noqa: E402
from oracle import a2c_oracle as O  # noqa: E402

R = MG.R


def time_rollout(kind, T, n_slots, A=3):
    """batch-1 forwards, one torch thread (the reference's process-per-env design, training.py:116-121)"""
    torch.set_num_threads(1)
    ss = (4, 84, 84)
    ekw = dict(env_id=3, rew_period=5, done_period=31)
    env_type = "FakePong-v0"
    MG._ENV_SPECS[env_type] = dict(env_kwargs=ekw, n_actions=A)
    hyps = base_hyps(env_type=env_type, n_tsteps=T, n_rollouts=n_slots, action_shift=1)
    N = T * n_slots
    us = hashf(N + 8, 31337)
    out = {}
    for who in ("reference", "oracle"):
        datas = dict(states=torch.zeros(N, *ss), deltas=torch.zeros(N), rewards=torch.zeros(N),
                     actions=torch.zeros(N).long(), dones=torch.zeros(N))
        if kind == "GRUModel":
            datas["h_states"] = torch.zeros(N, 256)
        if who == "reference":
            net = MG.ref_model(kind, ss, A, 256)
            rew_q = queue.Queue(1)
            rew_q.put(-1)
            runner = R["runner"].Runner(datas, hyps, None, None, rew_q)
            runner.net = net
            runner.env = R["runner"].SequentialEnvironment(**hyps)
            runner.state_bookmark = R["utils"].next_state(runner.env, runner.obs_deque, obs=None, reset=True)
            runner.h_bookmark = torch.zeros(1, net.h_size) if net.is_recurrent else None
            runner.ep_rew = 0
            for p in net.parameters():
                p.requires_grad = False
            it = iter(us)
            real_rand = torch.rand
            torch.rand = lambda *shape: torch.tensor([next(it)], dtype=torch.float32).reshape(shape)
            t0 = time.perf_counter()
            for idx in range(n_slots):
                runner.rollout(net, idx, hyps)
            dt = time.perf_counter() - t0
            torch.rand = real_rand
        else:
            onet = O.OracleNet(kind, ss, A, 256)
            it = iter(us)
            sr = O.SlotRunner(O.FakeEnv(**ekw), datas, hyps, uniform_fn=lambda: float(next(it)))
            sr.start(onet)
            t0 = time.perf_counter()
            for idx in range(n_slots):
                sr.rollout(onet, idx)
            dt = time.perf_counter() - t0
        out[who] = (dt, datas)
    same = all(torch.equal(out["reference"][1][k], out["oracle"][1][k]) for k in ("actions", "dones", "states"))
    return out["reference"][0], out["oracle"][0], N, same


def time_update(kind, R_, T, threads, bptt=False, A=3):
    torch.set_num_threads(threads)
    ss, h = (4, 84, 84), 256
    hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type="RMSprop", use_bptt=bptt, h_size=h)
    res = {}
    for who in ("reference", "oracle"):
        recurrent = kind == "GRUModel"
        D = synth_shared(kind, ss, A, h, R_, T, seed=900, recurrent=recurrent)
        if who == "reference":
            net = MG.ref_model(kind, ss, A, h)
            upd = R["updater"].Updater(net, hyps)
        else:
            net = O.OracleNet(kind, ss, A, h)
            upd = O.OracleUpdater(net, hyps)
        upd.update_model(D)                           # warm (allocator, thread pool)
        D = synth_shared(kind, ss, A, h, R_, T, seed=910, recurrent=recurrent)
        t0 = time.perf_counter()
        info = upd.update_model(D)
        res[who] = (time.perf_counter() - t0, {k: float(v) for k, v in info.items()})
    same = all(abs(res["reference"][1][k] - res["oracle"][1][k]) <= 1e-6 + 1e-5 * abs(res["oracle"][1][k]) for k in res["oracle"][1])
    return res["reference"][0], res["oracle"][0], R_ * T, same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    print(f"host: {cpu}, {a.threads} threads for the update, torch {torch.__version__}\n")
    print("| what | samples | reference s | oracle s | oracle / reference | same outputs |")
    print("|---|---|---|---|---|---|")
    for kind, T, n in (("A3CModel", 64, 4), ("GRUModel", 32, 2), ("ConvModel", 16, 2)):
        r, o, N, same = time_rollout(kind, T, n)
        print(f"| `Runner.rollout` {kind}, batch-1 forwards, 1 thread | {N} env-steps | {r:.3f} | {o:.3f} | {o / r:.2f} | {same} |")
    for kind, R_, T, bptt in (("A3CModel", 32, 64, False), ("GRUModel", 8, 32, True), ("ConvModel", 4, 16, False)):
        r, o, N, same = time_update(kind, R_, T, a.threads, bptt)
        print(f"| `Updater.update_model` {kind}{' +BPTT' if bptt else ''}, {a.threads} threads | {N} | {r:.3f} | {o:.3f} | {o / r:.2f} | {same} |")


if __name__ == "__main__":
    main()
