import os, sys, subprocess, json
ROOT="/root/repo"
code = r'''
import os, sys, numpy as np, torch
sys.path[:0]=["%s","%s/pytorch-a2c_amd","%s/tests/golden","%s/tests"]
from cases import UPDATE_CASES, base_hyps, synth_shared
from test_gpu_models import make_net
from a2c_amd.updater import Updater
from oracle import a2c_oracle as O
case=[c for c in UPDATE_CASES if c[0]=="gru_bptt_rms"][0]
name, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, n_upd = case
net = make_net(kind, ss, A, h)
hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs, use_nstep_rets=nstep, use_bptt=use_bptt, h_size=h, max_norm=1e9)
D = synth_shared(kind, ss, A, h, R_, T, seed=700, recurrent=net.is_recurrent)
Updater(net, hyps).update_model({k: v.cuda() for k, v in D.items()})
g = {n: net.G(n).cpu().double().numpy() for n, _ in net.named_parameters() if n not in net._unused_params}
o64 = O.OracleNet(kind, ss, A, h, state_dict={k: v.double() for k, v in O.formula_state_dict(kind, ss, A, h).items()})
i64, g64 = O.update_grads_chunked(o64, D, hyps, R_, dtype=torch.float64)
i32, g32 = O.update_grads_chunked(O.OracleNet(kind, ss, A, h), D, hyps, R_, dtype=torch.float32)
out={}
for n in g:
    gx=g64[n].numpy(); rms=np.sqrt((gx**2).mean())
    out[n]=(float(np.sqrt(((g[n]-gx)**2).mean())/rms), float(np.sqrt(((g32[n].double().numpy()-gx)**2).mean())/rms), float(np.abs(g[n]-gx).max()/rms))
np.save(sys.argv[1], g, allow_pickle=True)
print({k:(f"{a:.1e}",f"{b:.1e}",f"{c:.1e}") for k,(a,b,c) in out.items() if "convs" in k})
''' % (ROOT,ROOT,ROOT,ROOT)
open("/tmp/_cmp.py","w").write(code)
for v in ("1","0"):
    env=dict(os.environ, A2C_NO_ODD_BS=v)
    r=subprocess.run([sys.executable,"/tmp/_cmp.py",f"/tmp/g_{v}.npy"],env=env,capture_output=True,text=True)
    print("NO_ODD_BS="+v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
import numpy as np
a=np.load("/tmp/g_1.npy",allow_pickle=True).item(); b=np.load("/tmp/g_0.npy",allow_pickle=True).item()
for n in a:
    d=np.abs(a[n]-b[n]).max(); s=np.abs(a[n]).max()
    if d>0: print(n, "max |generic - odd kernel| / max:", d/s)
