import sys, os
ROOT = os.getcwd()
sys.path[:0] = [ROOT, ROOT + "/pytorch-a2c_amd", ROOT + "/tests/golden", ROOT + "/tests"]
import numpy as np, torch
from oracle import a2c_oracle as O
from cases import U8FakeEnv, base_hyps, hashf
from test_gpu_models import _datas, make_net
from test_gpu_ingest import _pool, _oracle_rollouts
from a2c_amd.runner import Runner
B, T, A, ss = 5, 6, 3, (4, 84, 84)
kind = "A3CModel"
ekws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + 2 * j) for j in range(B)]
hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, action_shift=0, n_envs=B, env_timeout_s=20.0)
net = make_net(kind, ss, A, 256)
onet = O.OracleNet(kind, ss, A, 256)
D = _datas(B * T, ss, False, actions_on_host=False)
us = torch.from_numpy(hashf(2 * T * B, 901, 0, 1).reshape(2, T, B))
usd = us.to("cuda")
rnd = [0]
pool = _pool(U8FakeEnv, ekws, 2, pong=True)
r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=sys.argv[1], uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
refs = _oracle_rollouts(kind, onet, hyps, ekws, us, 2, B, T, ss)
r.rollout(net, list(range(B)), hyps); r.finish()
S = D["states"].cpu().reshape(B, T, 4, -1); R = refs[0]["states"].reshape(B, T, 4, -1)
eq = (S == R).all(-1)
print(eq.int())
bad = (~eq).nonzero()
for b, t, c in bad[:6].tolist():
    d = (S[b, t, c] != R[b, t, c]).nonzero().flatten()
    print("slot", b, "t", t, "plane", c, "ndiff", len(d), "first", d[:8].tolist(), "last", d[-4:].tolist(), "sum got", S[b,t,c].sum().item(), "ref", R[b,t,c].sum().item())
    # does it equal some other plane of ref?
    for b2 in range(B):
        for t2 in range(T):
            for c2 in range(4):
                if torch.equal(S[b,t,c], R[b2,t2,c2]): print("   == ref", b2, t2, c2)
print("dones", D["dones"].cpu().reshape(B,T))
pass
def ident(f):
    for j in range(8):
        for t in range(0, 10):
            for nr in range(0, 4):
                g = O.formula_frames(1, (1,84,84), seed=j*100003 + t*17 + nr*7919, binary=True)[0].reshape(-1)
                if np.array_equal(g, f): return (j, t, nr)
    return None
print("got   slot0 t0 plane3:", ident(S[0,0,3].numpy()), " ref:", ident(R[0,0,3].numpy()))
print("got   slot0 t1 plane3:", ident(S[0,1,3].numpy()), " ref:", ident(R[0,1,3].numpy()))
print("got   slot1 t0 plane3:", ident(S[1,0,3].numpy()), " ref:", ident(R[1,0,3].numpy()))
r.close()
