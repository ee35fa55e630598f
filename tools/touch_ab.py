"""In-step A/B of the two weight-latency measures of the persistent chain (debug library): VKN_CHAIN_TOUCH (the gather reduction warms the
memory-side cache with the next chain's weights) and the per-unit cache-line prefetch distance (VKN_CHAIN_ABL 11 / 12 / 13 = 0 / 4 / 12 units).
    python tools/touch_ab.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import vkn_import
vkn = vkn_import.load()
vkn._lib.use_debug()
DEV = 'cuda:0'
N, C, H, W = 117, 256, 128, 256
cfg = vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100)
head = vkn.build_head(cfg); torch.manual_seed(0); head.init_weights(); head = head.to(DEV).eval()
g = torch.Generator().manual_seed(1)
B = 32
x = torch.randn(B, C, H, W, generator=g).to(DEV); pf = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
mp = (torch.randn(B, N, H, W, generator=g) * 4).to(DEV); prev = torch.randn(B, N, C, 1, 1, generator=g).to(DEV)
def timeit(fn, iters=30, warm=6):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
with torch.no_grad():
    ref = None
    for rep in range(3):
        for t, abl in ((1, 0), (0, 0)):
            os.environ['VKN_CHAIN_TOUCH'] = str(t)
            os.environ['VKN_CHAIN_ABL'] = str(abl)
            ms = timeit(lambda: head._head_forward(x, pf, mp, prev, want_track=True))
            out = head._head_forward(x, pf, mp, prev, want_track=True)
            if ref is None: ref = out
            same = all(torch.equal(a, b) for a, b in zip(out, ref))
            pfd = 0
            print(f'touch={t} line-prefetch {pfd:2d} units ahead: {ms:.3f} ms per 32-frame step -> {B / ms * 1e3:.0f} frames/s  identical outputs: {same}')
