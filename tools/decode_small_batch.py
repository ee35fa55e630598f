"""Round 5: the last-stage mask decode at 1 / 2 / 4 frames per launch (kernel rows spread over blockIdx.z below 128 workgroups): time and error vs fp64."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
DEV='cuda:0'
def timeit(fn, iters=100, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
g=torch.Generator().manual_seed(0)
for B in (1,2,4):
    x=torch.randn(B,256,128,256,generator=g).to(DEV); k=torch.randn(B,117,256,generator=g).to(DEV); kb=torch.randn(B,117,generator=g).to(DEV)
    hi,lo=vkn.ops.split_planes(k); out=torch.empty(B,117,128,256,device=DEV)
    t=timeit(lambda: vkn.ops.mask_decode_planes(x,hi,lo,117,kb,out))
    ref=torch.einsum('bnc,bchw->bnhw', k.double(), x.double())+kb.double()[:,:,None,None]
    print(f'decode B={B}: {t:.1f} us   max err vs fp64 {float((out.double()-ref).abs().max()):.2e}')
