#!/usr/bin/env python3
"""Time of one ma_ctc_loss_grad_f32 call (log-sum-exp, alpha / beta recursions, dlogits) at the cfg-4 shape (40 x 255 frames, V = 4233)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from mindaudio_amd.train import kernels as K
b,t,V,Vp=40,255,4233,4288
g=torch.Generator().manual_seed(1)
logits=torch.randn(b*t,Vp,generator=g).cuda()
ys=torch.randint(1,V,(b,30),generator=g,dtype=torch.int32).cuda()
hl=torch.full((b,),t,dtype=torch.int32).cuda(); yl=torch.full((b,),30,dtype=torch.int32).cuda()
for _ in range(3): out=K.ctc_loss_grad(logits,V,b,t,ys,hl,yl,1.0)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(50): out=K.ctc_loss_grad(logits,V,b,t,ys,hl,yl,1.0)
torch.cuda.synchronize(); print("ctc_loss_grad us per call: %.1f"%((time.perf_counter()-t0)/50*1e6), float(out[0]))
