"""GPU: time vkn_mask_losses_bwd_lowres_f32 alone at the cfg3 training size (B = 4, Ns = 117, 128x256 -> x4) against the pair it replaces."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
from importlib import import_module
L = vkn._lib.lib(); ops = vkn.ops
B, Ns, h, w, S, K = 4, 117, 128, 256, 4, 80
dev = 'cuda:0'
g = torch.Generator().manual_seed(1)
low = (torch.randn(B, Ns, h, w, generator=g) * 3).to(dev)
scaled = ops.upsample_bilinear(low, S)
P = S * h * S * w
bank = (torch.rand(K, S * h, S * w, generator=g) > 0.5).float().to(dev)
rowk = torch.full((B * Ns,), -1, dtype=torch.int32); tgt = torch.full((B * Ns,), -1, dtype=torch.int32)
pos = torch.randperm(B * Ns, generator=g)[:K].sort()[0]
rowk[pos] = torch.arange(K, dtype=torch.int32); tgt[pos] = torch.arange(K, dtype=torch.int32)
rowk, tgt = rowk.to(dev), tgt.to(dev)
a = torch.rand(K, generator=g).to(dev) * 1e5; bc = (torch.rand(K, generator=g) * 1e5 + 1e5).to(dev)
lse = torch.logsumexp(scaled, 1).contiguous(); top = torch.randint(-1, Ns, (B, S * h, S * w), generator=g).int().to(dev)
one = torch.ones(1, device=dev)
out_lr = torch.empty_like(low); gs = torch.empty_like(scaled)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def lr():
    assert L.vkn_mask_losses_bwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), p(a), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K, p(lse), p(top), B, Ns, h, w, S, 1, p(out_lr), st) == 0
def pair():
    assert L.vkn_mask_losses_bwd_bank_f32(p(scaled), p(bank), p(tgt), p(rowk), p(a), p(bc), p(one), p(one), p(one), 1.0, 4.0, 0.1, K, p(lse), p(top), B, Ns, P, 1, p(gs), st) == 0
    return ops.upsample_bilinear_bwd(gs, S)
ref = pair(); lr(); torch.cuda.synchronize()
print('max |lowres - pair| / max', float((out_lr - ref).abs().max() / ref.abs().max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for tag, f in (('low-res kernel', lr), ('bwd_bank + upsample adjoint', pair)):
    for _ in range(3): f()
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us')
