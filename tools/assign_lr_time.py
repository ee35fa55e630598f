"""GPU: the assignment costs of one cfg3 training stage (4 images, 100 kernels, 128x256 -> x4) — the low-res kernel against the three-pass form on the up-scaled tensor."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
import oracle.synth as synth
dev = 'cuda:0'
B, N, Ns, h, w, S, ncls = 4, 100, 117, 128, 256, 4, 19
Gs = [int(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else (20, 20, 20, 20))]
B = len(Gs)
g = torch.Generator().manual_seed(1)
low = (torch.randn(B, Ns, h, w, generator=g) * 3).to(dev)
gts = [(torch.rand(G, S * h, S * w, generator=g) > 0.7).float().to(dev) for G in Gs]
cls = [torch.randn(N, ncls, generator=g).to(dev) for _ in Gs]
labs = [torch.randint(0, ncls, (G,), generator=g).to(dev) for G in Gs]
up = vkn.ops.upsample_bilinear(low, S)
lr = lambda: vkn.ops.assign_costs_lowres_batch([low[b][:N] for b in range(B)], S, cls, gts, labs)
old = lambda: vkn.ops.assign_costs_batch([up[b][:N] for b in range(B)], cls, gts, labs)
a, b_ = lr(), old()
torch.cuda.synchronize()
print('max |lowres - three-pass|', max(float((x - y).abs().max()) for x, y in zip(a, b_)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for tag, f in (('low-res kernel', lr), ('three-pass form', old)):
    for _ in range(3): f()
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per stage ({B} images, G = {Gs})')
