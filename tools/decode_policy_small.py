"""Round 6 (debug library): cache policy of the x loads in the last-stage decode at 1 .. 8 frames per launch — default (sc0 nt), plain (cache-allocating),
nt only.  After the gather / fused passes of the same step x (33.5 MB per frame) may still sit in the 256 MiB memory-side cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
vkn._lib.use_debug()
DEV = 'cuda:0'
def timeit(fn, iters=200, warm=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
g = torch.Generator().manual_seed(0)
for B in (1, 2, 4, 8):
    x = torch.randn(B, 256, 128, 256, generator=g).to(DEV); k = torch.randn(B, 117, 256, generator=g).to(DEV); kb = torch.randn(B, 117, generator=g).to(DEV)
    hi, lo = vkn.ops.split_planes(k); out = torch.empty(B, 117, 128, 256, device=DEV)
    row = []
    for abl in (0, 4, 6):
        os.environ['VKN_DECODE_ABL'] = str(abl)
        row.append(timeit(lambda: vkn.ops.mask_decode_planes(x, hi, lo, 117, kb, out)))
    print(f'decode B={B}: sc0|nt {row[0]:.1f} us   plain {row[1]:.1f} us   nt {row[2]:.1f} us')
