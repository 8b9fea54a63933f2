#!/usr/bin/env python3
"""Time mpv_reduce on the BASELINE configs[1] sample volume (181 windows x 10 000 rows x 32 samples = 232 MB) and check it
against a float64 torch reduction."""
import torch, sys
sys.path.insert(0,'.')
from mpv_amd import _native
ctx=_native.Context(torch.device('cuda',0))
x=torch.randn(181,10000,1,32,device='cuda')
std=torch.tensor([2.0],device='cuda'); mean=torch.tensor([0.5],device='cuda')
ctx.reduce(x,std,mean); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): mpv,pm,_=ctx.reduce(x,std,mean)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/20
want=(x.double()*2+0.5).var(dim=-1,unbiased=False).mean(dim=(1,2))
print("reduce 232 MB: %.3f ms = %.2f TB/s; max rel err %.2e"%(ms, x.numel()*4/ms/1e9, float(((mpv.double()-want).abs()/want).max())))
