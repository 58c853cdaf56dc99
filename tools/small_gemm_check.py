#!/usr/bin/env python3
"""The small dense products of the conv-stack models' rollout step through a2c_gemm_f32 (error against fp64, HIP-event time):
    python tools/small_gemm_check.py            # the dispatch as shipped
    A2C_NO_SMALL_GEMM=1 python tools/small_gemm_check.py   # block-tiled kernel + split-K reduce instead of the one-launch kernel"""
import os, sys
sys.path[:0]=['/root/repo','/root/repo/pytorch-a2c_amd']
import torch
from a2c_amd import ops
dev=torch.device('cuda')
def timeit(fn, reps=200):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
for (M,N,K) in [(256,256,2304),(256,768,256),(256,576,2000),(32,576,2000)]:
    x=torch.rand(M,K,device=dev)-0.5; W=(torch.rand(N,K,device=dev)-0.5)*0.1; b=torch.rand(N,device=dev)
    out=torch.empty(M,N,device=dev)
    sk=ops.pick_splitk(M,N,K)
    ws=torch.empty(max(ops.gemm_ws_bytes(M,N,sk),4)//4,device=dev) if sk>1 else None
    f=lambda: ops.gemm(0,1,M,N,K,x.data_ptr(),K,W.data_ptr(),K,out.data_ptr(),N,bias=b,relu=True,splitk=sk,ws=ws)
    f(); ref=torch.relu(x.double()@W.double().t()+b.double())
    err=float((out.double()-ref).abs().max())
    print(M,N,K,'splitk',sk,'err %.1e'%err,'%.1f us'%timeit(f), flush=True)
