import os, sys
sys.path[:0]=['/root/repo','/root/repo/pytorch-a2c_amd']
import torch
from a2c_amd import ops
dev=torch.device('cuda')
def timeit(fn, reps=300):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for (M,N,K) in [(256,256,2304),(256,576,2000),(32,576,2000)]:
    x=torch.rand(M,K,device=dev)-0.5; W=(torch.rand(N,K,device=dev)-0.5)*0.1; b=torch.rand(N,device=dev)
    out=torch.empty(M,N,device=dev)
    ref=torch.relu(x.double()@W.double().t()+b.double())
    for sk in (ops.pick_splitk(M,N,K), 4, 8, 12, 16, 24, 36):
        ws=torch.empty(max(ops.gemm_ws_bytes(M,N,sk),4)//4,device=dev)
        f=lambda: ops.gemm(0,1,M,N,K,x.data_ptr(),K,W.data_ptr(),K,out.data_ptr(),N,bias=b,relu=True,splitk=sk,ws=ws)
        f(); err=float((out.double()-ref).abs().max())
        print(M,N,K,'splitk',sk,'err %.1e'%err,'%.1f us'%timeit(f), flush=True)
