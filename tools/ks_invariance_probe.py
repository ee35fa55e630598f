"""Is the few-row chain bit-invariant under its tile shape?  (debug library: VKN_KS_WGCAP forces fatter / thinner tiles.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkn_import
vkn = vkn_import.load()
vkn._lib.build_debug(); vkn._lib.use_debug()
DEV = 'cuda:0'
N, C, H, W = 117, 256, 16, 32
cfg = vkn.configs.roi_head_cfg(True, C=C, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=1, up=1, nprop=100)
head = vkn.build_head(cfg); torch.manual_seed(0); head.init_weights(); head = head.to(DEV).eval()
g = torch.Generator().manual_seed(3)
for B in (3, 6, 8):
    dims = head.mask_head[0].make_dims(B, N, H, W)
    pack = head.mask_head[0].stage_pack(torch.device(DEV))
    xf = (torch.randn(B, N, C, generator=g) * 50).to(DEV); ob = torch.randn(B, N, C, generator=g).to(DEV)
    outs = {}
    for cap in (100000, 768, 256, 64):
        os.environ['VKN_KS_WGCAP'] = str(cap)
        outs[cap] = vkn.ops.stage_chain(dims, pack, xf, ob, flags=vkn.ops.FLAG_CHAIN_KSPLIT)
    ref = outs[100000]
    for cap in (768, 256, 64):
        d = [float((a - b).abs().max()) for a, b in zip(outs[cap], ref)]
        print(f'B={B} cap {cap} vs all-(1,1): max abs diff cls/kern/kb/obj = {d}')
