"""per-tensor gradient accuracy of the full-size updates: HIP fp32 and oracle fp32 against the oracle in fp64"""
import copy, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-a2c_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from oracle import a2c_oracle as O
from cases import base_hyps
from test_gpu_models import _datas, make_net
from a2c_amd.hostpool import ThreadEnvPool
from a2c_amd.runner import Runner
from a2c_amd.synthetic import TapeEnv
from a2c_amd.updater import Updater

torch.set_num_threads(16)
kind, B, T, ingest = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
A, ss = 3, (4, 84, 84)
hyps = base_hyps(env_type="Pong-synthetic", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B)
net = make_net(kind, ss, A, 256)
onet = O.OracleNet(kind, ss, A, 256)
D = _datas(B * T, ss, False, actions_on_host=False)
envs = [TapeEnv(env_id=j, length=T + 1, p_done=1.0 / 100) for j in range(B)]
pool = ThreadEnvPool.from_tape_envs(envs, n_threads=4, pong=True, frame_bits=True)
r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest)
try:
    r.rollout(net, list(range(B)), hyps); r.finish()
    Do = {k: v.cpu().clone() for k, v in D.items()}
    upd = Updater(net, hyps)
    info = upd.update_model(D)
finally:
    r.close()
o64 = O.OracleNet(kind, ss, A, 256, state_dict={k: v.double() for k, v in O.formula_state_dict(kind, ss, A, 256).items()})
t0 = time.time()
oinfo, ex = O.OracleUpdater(onet, hyps).update_model(Do, keep=True)
t1 = time.time()
_d = O.discount
O.discount = lambda a, d, f: _d(a, d, f).to(a.dtype)     # same fp32 scans feed both precisions
D64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in Do.items()}
oinfo64, ex64 = O.OracleUpdater(o64, hyps).update_model(D64, keep=True)
t2 = time.time()
print("oracle32 %.1fs oracle64 %.1fs" % (t1 - t0, t2 - t1))
print("info hip", info); print("info o32", oinfo); print("info o64", oinfo64)
for n, p in net.named_parameters():
    g64 = ex64["grads"][n]
    if g64 is None: continue
    g32 = ex["grads"][n].double(); gh = net.G(n).cpu().double()
    rms = float(g64.pow(2).mean().sqrt())
    print("%-24s rms %.3e | hip-64 rms %.2e max %.2e | o32-64 rms %.2e max %.2e | hip-o32 rms %.2e max %.2e  (rel to rms)" % (
        n, rms, float((gh - g64).pow(2).mean().sqrt()) / rms, float((gh - g64).abs().max()) / rms,
        float((g32 - g64).pow(2).mean().sqrt()) / rms, float((g32 - g64).abs().max()) / rms,
        float((gh - g32).pow(2).mean().sqrt()) / rms, float((gh - g32).abs().max()) / rms))
