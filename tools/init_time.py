"""GPU: time the kernel-initialisation pass (vkn_kernel_init_f32) at cfg2 size, one-pass form against the round-5 form; bit-compare."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import
vkn = vkn_import.load()
B, C, H, W, P0 = int(os.environ.get('B', 32)), 256, 128, 256, 100
g = torch.Generator().manual_seed(3)
loc = torch.randn(B, C, H, W, generator=g).cuda(); sem = torch.roll(loc, 1, 0)
iw = (torch.randn(P0, C, 1, 1, generator=g) * 0.05).cuda(); sw = (torch.randn(19, C, 1, 1, generator=g) * 0.05).cuda(); sb = (torch.randn(19, generator=g) * 0.1).cuda()
def run(flags, seg=True):
    return vkn.ops.kernel_init(loc, sem, iw, sw, sb, 2, True, True, want_seg_preds=seg, flags=flags)
a = run(0); b_ = run(vkn.ops.FLAG_INIT_SEPARATE)
for nm, u, v in zip(('proposal_feats', 'x_feats', 'mask_preds', 'seg_preds'), a, b_):
    print(nm, 'bit-identical' if torch.equal(u, v) else f'DIFFERENT max {float((u - v).abs().max()):.3e}')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for tag, fl in (('one pass', 0), ('round-5 form', vkn.ops.FLAG_INIT_SEPARATE)):
    for _ in range(3): run(fl, False)
    e0.record()
    for _ in range(10): run(fl, False)
    e1.record(); torch.cuda.synchronize()
    print(f'kernel init pass 0, B={B}, {tag}: {e0.elapsed_time(e1) / 10:.3f} ms')
