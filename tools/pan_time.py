"""GPU: time vkn_panoptic_joint_f32 at cfg2 geometry on bench.py's panoptic inputs (B = 32)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vkn_import
import bench
vkn = vkn_import.load()
B, N, P0 = int(os.environ.get('B', 32)), 117, 100
pc, pl = bench.panoptic_inputs(B, N, P0, 19, 128, 256, 'cuda:0')
full = (1024, 2048)
pan = lambda: vkn.ops.panoptic_joint(pc, pl, P0, 2, P0, 0.25, 0.6, full, full, full, upsample_stride=4)
for _ in range(3): pan()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): pan()
e1.record(); torch.cuda.synchronize()
print(f'panoptic_joint B={B}: {e0.elapsed_time(e1) / 10:.3f} ms')
def timed(tag, cls_, lg_):
    f = lambda: vkn.ops.panoptic_joint(cls_, lg_, P0, 2, P0, 0.25, 0.6, full, full, full, upsample_stride=4)
    for _ in range(2): f()
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{tag}: {e0.elapsed_time(e1) / 5:.3f} ms')
if os.environ.get('PAN_VARIANTS'):
    lg = torch.full_like(pl, -10.0); lg[:, 0] = 10.0
    timed('one survivor per tile', pc, lg)
    lg = torch.randn_like(pl)
    timed('all 117 survive (iid noise)', pc, lg)
