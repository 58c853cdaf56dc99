#!/usr/bin/env python3
"""How far four sharded graphed updates (world 2) drift from four single-process eager ones, per update and info key:
the body of tests/test_gpu_timed_path.py::test_sharded_graphed_update_equals_single_process_eager with the differences printed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"),
                os.path.join(ROOT, "oracle")]
import numpy as np  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def main(kind):
    from a2c_amd.updater import Updater
    from test_gpu_system import _free_port
    from test_gpu_timed_path import _graphed_shard_worker, make_net, base_hyps, synth_shared
    ss, A, h, R, T = (4, 84, 84), 3, 256, 4, 6
    net = make_net(kind, ss, A, h)
    upd = Updater(net, base_hyps(n_tsteps=T, n_rollouts=R, optim_type="RMSprop", h_size=h))
    ref = []
    for u in range(4):
        D = synth_shared(kind, ss, A, h, R, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
        ref.append(upd.update_model({k: v.cuda() for k, v in D.items()}))
    ref_params = [p.detach().cpu().numpy() for p in net.parameters()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphed_shard_worker, args=(r, 2, port, kind, q, ss)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
    rank, infos, params, _, _ = res[0]
    for u in range(4):
        print(u, " ".join(f"{k}={abs(infos[u][k] - ref[u][k]) / (abs(ref[u][k]) + 1e-30):.1e}" for k in ref[u]))
    print("params max abs diff", max(float(np.abs(a - b).max()) for a, b in zip(params, ref_params)),
          "elements further than 4e-6:", sum(int((np.abs(a - b) > 4e-6).sum()) for a, b in zip(params, ref_params)), "of",
          sum(a.size for a in params))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "GRUModel")
