"""GPU: vkn_mask_losses_fwd_lowres_f32 at the cfg3 training size (B = 4, Ns = 117, 128x256 -> x4) against what it replaces
(k_upsample_s + vkn_mask_losses_fwd_bank_f32 on the up-scaled tensor): values and time."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
L = vkn._lib.lib(); ops = vkn.ops
B, Ns, h, w, S, K = 4, 117, 128, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 4, 80
dev = 'cuda:0'
g = torch.Generator().manual_seed(1)
low = (torch.randn(B, Ns, h, w, generator=g) * 3).to(dev)
P = S * h * S * w
bank = (torch.rand(K, S * h, S * w, generator=g) > 0.5).float().to(dev)
rowk = torch.full((B * Ns,), -1, dtype=torch.int32); tgt = torch.full((B * Ns,), -1, dtype=torch.int32)
pos = torch.randperm(B * Ns, generator=g)[:K].sort()[0]
rowk[pos] = torch.arange(K, dtype=torch.int32); tgt[pos] = torch.arange(K, dtype=torch.int32)
rowk, tgt, posd = rowk.to(dev), tgt.to(dev), pos.to(dev)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nch, nbl, ncl = L.vkn_mask_losses_chunks(P), L.vkn_mask_losses_blocks(P), L.vkn_mask_losses_lowres_chunks(h, w)
rp0, rk0 = torch.zeros(K, nch, 4, device=dev), torch.zeros(B, nbl, device=dev)
lse0, top0 = torch.zeros(B, P, device=dev), torch.zeros(B, P, dtype=torch.int32, device=dev)
rp1, rk1 = torch.zeros(K, ncl, 4, device=dev), torch.zeros(B, ncl, device=dev)
lse1, top1 = torch.zeros(B, P, device=dev), torch.zeros(B, P, dtype=torch.int32, device=dev)
def old():
    scaled = ops.upsample_bilinear(low, S)
    assert L.vkn_mask_losses_fwd_bank_f32(p(scaled), p(bank), p(tgt), p(posd), p(rowk), K, B, Ns, P, 1, p(rp0), p(lse0), p(top0), p(rk0), st) == 0
def new():
    assert L.vkn_mask_losses_fwd_lowres_f32(p(low), p(bank), p(tgt), p(rowk), K, B, Ns, h, w, S, 1, p(rp1), p(lse1), p(top1), p(rk1), st) == 0
old(); new(); torch.cuda.synchronize()
a, b_ = rp0.double().sum(1), rp1.double().sum(1)
print('row sums: max rel diff', float(((a - b_).abs() / a.abs().clamp(min=1.0)).max()))
print('lse: max abs diff', float((lse0 - lse1).abs().max()), ' top equal:', bool(torch.equal(top0, top1)))
print('rank loss sum', float(rk0.double().sum()), float(rk1.double().sum()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for tag, f in (('low-res forward', new), ('upsample + fwd_bank', old)):
    for _ in range(3): f()
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us')
