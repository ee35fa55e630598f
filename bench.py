#!/usr/bin/env python3
"""bench.py — frames/s of the Video K-Net kernel-update head on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: re-executes itself under torch.distributed.run, one rank per GPU;
                                                            also runs when the driver already launched it that way)

A "step" = one pass of the hot path over one clip of `--frames` synthetic 1024x2048 frames per GPU:
`VideoKernelIterHead` (S=3 stages of gather -> kernel update + interaction -> decode, N = 100 proposals + 17 stuff kernels,
C = 256) + the last stage's tracking link (previous_type="ffn") + the x4 bilinear upsample of the final logits
(`_mask_forward`), i.e. everything `simple_test_mask_preds_plus_previous` does per frame; inputs are
resident in HBM before the timed region.  Frames of a clip are sharded over the ranks in contiguous blocks (weak scaling:
`--frames` per GPU); the only cross-rank data is the [N x C] kernel set of a block's last frame, which the next rank's first
frame needs for its tracking embedding (RCCL all_gather, 120 KB per rank, inside the timed region).

`--clip T` = STRONG scaling, the shape BASELINE cfg3 words (one clip of T = 8 frames over 8 GPUs, one frame per GPU, as the reference
trains — samples_per_gpu = 1 — and infers, one frame per call): the SAME T-frame clip in contiguous blocks of T / N frames per rank,
`"scaling": "strong"`, value = T frames x steps / time.  At N = 1 the `breakdown` additionally carries the measured step time at
T/2, T/4, ... 1 frames per call (per-rank compute on ONE GPU, labelled as such — not a scaling result).

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (mask-decode kernel, HIP-event timed, algorithmic bytes / time
vs 8 TB/s HBM), "cpu_baseline" (the torch CPU oracle timed on this host, rank 0 / N=1 only), "breakdown" (incl. the whole step at
1 / 8 / 32 frames per call: the reference walks a video one frame per call).

`--train` (BASELINE cfg3, not the headline): one step = `forward_train_with_previous` of the same head on `--frames` frames per GPU
(losses, Hungarian assignment, backward through the HIP kernels' autograd wrappers), bucketed RCCL all-reduce of the head's
gradients overlapped with backward, SGD update; prints training frames/s.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CFG2 = dict(C=256, heads=8, ffn=2048, ncls=19, n_thing=2, n_stuff=17, S=3, up=4, nprop=100, N=117, H=128, W=256)
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
PMC_SIDECAR = 'profiles/r06_pmc.json'   # HBM counters of this same command (tools/gpu_profile.sh); `roofline.traffic` is read from it


def build_head(vkn, device, seed=0, link='ffn'):
    # link='update': the `*_joint_update` KITTI-STEP configs (previous_link='update_dynamic_cov', previous_type='update') — the
    # last stage of frame t depends on frame t-1's final kernels (bench.py --head update; never the headline line)
    over = dict(previous_link='update_dynamic_cov', previous_type='update') if link == 'update' else None
    head = vkn.build_head(vkn.configs.roi_head_cfg(True, C=CFG2['C'], heads=CFG2['heads'], ffn=CFG2['ffn'], ncls=CFG2['ncls'],
                               n_thing=CFG2['n_thing'], n_stuff=CFG2['n_stuff'], S=CFG2['S'], up=CFG2['up'],
                               nprop=CFG2['nprop'], mask_over=over))
    torch.manual_seed(seed)
    head.init_weights()                      # xavier-uniform, fc_cls.bias = -log 99 (reference init; no checkpoints offline)
    return head.to(device).eval()


def synth_inputs(B, device, seed):
    g = torch.Generator(device='cpu').manual_seed(1234 + seed)
    x = torch.randn(B, CFG2['C'], CFG2['H'], CFG2['W'], generator=g)
    pf = torch.randn(B, CFG2['N'], CFG2['C'], 1, 1, generator=g)
    mp = torch.randn(B, CFG2['N'], CFG2['H'], CFG2['W'], generator=g) * 4.0
    return x.to(device), pf.to(device), mp.to(device)


def panoptic_inputs(B, N, Np, ncls, H, W, device, seed=7):
    """cls probabilities and segmentation-like mask logits for the post-head timing: one soft elliptic blob per thing kernel,
    one horizontal band per stuff kernel, plus noise (i.i.d. noise would be rejected wholesale by the merge)."""
    g = torch.Generator(device='cpu').manual_seed(4321 + seed)
    cls = torch.rand(B, N, ncls, generator=g) * 0.96 + 0.02
    cx, cy = torch.rand(B, N, 1, 1, generator=g), torch.rand(B, N, 1, 1, generator=g)
    r0 = (0.5 / Np) ** 0.5
    rx, ry = (torch.rand(B, N, 1, 1, generator=g) + 0.5) * r0, (torch.rand(B, N, 1, 1, generator=g) + 0.5) * r0
    ys = ((torch.arange(H) + 0.5) / H).view(1, 1, H, 1)
    xs = ((torch.arange(W) + 0.5) / W).view(1, 1, 1, W)
    logits = (6.0 * (1.0 - torch.sqrt(((xs - cx) / rx) ** 2 + ((ys - cy) / ry) ** 2))).clamp(-6.0, 6.0)
    ns = N - Np
    for j in range(ns):
        inside = ((ys >= j / ns) & (ys < (j + 1) / ns)).expand(B, 1, H, W)[:, 0]
        logits[:, Np + j] = torch.where(inside, torch.tensor(4.0), torch.tensor(-4.0))
    logits = logits + 0.7 * torch.randn(B, N, H, W, generator=g)
    return cls.to(device), logits.to(device)


def _cpu_identity():
    """(CPU model string, physical core count) of this host from /proc/cpuinfo — BASELINE.md §3 asks for both beside the thread
    count the baseline actually used."""
    model, cores = 'unknown', set()
    try:
        phys = core = None
        with open('/proc/cpuinfo') as f:
            for line in f:
                k, _, v = line.partition(':')
                k, v = k.strip(), v.strip()
                if k == 'model name' and model == 'unknown':
                    model = v
                elif k == 'physical id':
                    phys = v
                elif k == 'core id':
                    core = v
                elif not k and phys is not None and core is not None:
                    cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or None)


def cpu_baseline(head_sd, sample_frames=1, runs=10):
    """The CPU oracle (same ATen op sequence as the reference) on this host's cores — kind 'port'."""
    from oracle.knet_oracle import HeadCfg, iter_head_mask_preds
    cfg = HeadCfg(num_stages=CFG2['S'], in_channels=CFG2['C'], num_heads=CFG2['heads'], num_classes=CFG2['ncls'],
                  mask_upsample_stride=CFG2['up'], feat_channels=CFG2['C'], previous_type='ffn',
                  extra=dict(feedforward_channels=CFG2['ffn']))
    ncpu = os.cpu_count() or 1
    sd = {k: v.detach().float().cpu() for k, v in head_sd.items()}
    g = torch.Generator().manual_seed(99)
    x = torch.randn(sample_frames, CFG2['C'], CFG2['H'], CFG2['W'], generator=g)
    pf = torch.randn(sample_frames, CFG2['N'], CFG2['C'], 1, 1, generator=g)
    mp = torch.randn(sample_frames, CFG2['N'], CFG2['H'], CFG2['W'], generator=g) * 4.0
    prev = torch.randn(sample_frames, CFG2['N'], CFG2['C'], 1, 1, generator=g)
    # pick the intra-op thread count that runs this workload fastest on this host (all cores is NOT it on a
    # 256-thread box: 19 s/frame); one probe run per candidate, then `runs` timed runs at the best
    best, best_t = 1, float('inf')
    with torch.no_grad():
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=prev)
            t0 = time.perf_counter()
            iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=prev)
            t = time.perf_counter() - t0
            if t < best_t:
                best, best_t = nt, t
    torch.set_num_threads(best)
    ts = []
    with torch.no_grad():
        for i in range(2 + runs):
            t0 = time.perf_counter()
            iter_head_mask_preds(sd, x, pf, mp, cfg, previous_obj_feats=prev)
            if i >= 2:
                ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    # the same head at BASELINE cfg1 size (512x1024 frame -> 64x128 features), for the record
    x1, mp1 = x[:, :, :64, :128].contiguous(), mp[:, :, :64, :128].contiguous()
    t1 = []
    with torch.no_grad():
        for i in range(5):
            t0 = time.perf_counter()
            iter_head_mask_preds(sd, x1, pf, mp1, cfg, previous_obj_feats=prev)
            if i >= 1:
                t1.append(time.perf_counter() - t0)
    t1.sort()
    model, phys = _cpu_identity()
    return dict(value=round(sample_frames / med, 4), unit='frames/s', cores=torch.get_num_threads(), kind='port',
                cpu_model=model, physical_cores=phys, logical_cores=ncpu, threads_used=torch.get_num_threads(),
                cfg1_size_value=round(sample_frames / t1[len(t1) // 2], 4),
                sample=f'{runs} timed runs (2 warm-up) of {sample_frames} frame(s) of the same workload, fp32, median, '
                       f'{best} intra-op threads (fastest of 8/16/32/64 on {ncpu} logical CPUs); '
                       f'min {sample_frames / ts[-1]:.3f} max {sample_frames / ts[0]:.3f} frames/s')


def _max_over_ranks(dt, steps, device, dist_on):
    """(MAX over ranks of the timed region, ms per step of every rank, the rank count the process group reports)."""
    if not dist_on:
        return dt, [round(dt / steps * 1e3, 4)], 1
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    ts = [float(p.item()) for p in parts]
    return max(ts), [round(v / steps * 1e3, 4) for v in ts], dist.get_world_size()


def train_main(args, vkn, vkn_dist, device, world, rank, dist_on=False):
    """BASELINE cfg3: the clip's frames sharded over the ranks, head trained data-parallel (see the module docstring)."""
    B = args.frames if args.frames != 32 else 4            # frames per GPU per step (the inference default of 32 is not a training batch)
    # x4: mask_upsample_stride of the shipped KITTI-STEP video config (configs/det/video_knet_kitti_step/...link_ffn_joint_train.py:102;
    # mask_assign_stride=2, :22) -> 512x1024 loss / assignment masks for a 1024x2048 frame.  Rounds 2-5 timed x2 (`--train-up 2`).
    N, C, H, W, up = CFG2['N'], CFG2['C'], CFG2['H'], CFG2['W'], args.train_up
    cfg = vkn.configs.roi_head_cfg(True, C=C, heads=CFG2['heads'], ffn=CFG2['ffn'], ncls=CFG2['ncls'], n_thing=CFG2['n_thing'],
                                   n_stuff=CFG2['n_stuff'], S=CFG2['S'], up=up, nprop=CFG2['nprop'],
                                   train_cfg=vkn.configs.rcnn_train_cfg(CFG2['S']))
    head = vkn.build_head(cfg)
    torch.manual_seed(0)                                   # identical replicas
    head.init_weights()
    head = head.to(device).train()
    reducer = vkn_dist.BucketedGradAllReducer(head, force_collectives=dist_on)
    torch_chain = getattr(args, 'torch_chain', False)
    if torch_chain:
        # A/B: the round-3 chain — torch autograd on the BLAS libraries' GEMMs.  Their default heuristic runs the 468-row problems of a
        # 4-frame step on a 256x256 macro-tile (8 workgroups, 114 us per call), so that arm lets PyTorch's TunableOp pick per shape.
        for stage in head.mask_head:
            stage.enable_device_chain(False)
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(True)
        torch.cuda.tunable.set_max_tuning_duration(10)
        torch.cuda.tunable.set_max_tuning_iterations(10)
        torch.cuda.tunable.set_filename(os.path.join('/tmp', 'vkn_tunableop_%d.csv' % rank))
    if not getattr(args, 'no_chain_graphs', False):
        head.enable_chain_graphs()                        # every stage's [B*N, C] chain, forward and backward, as captured hipGraphs
    if getattr(args, 'train_torch_sgd', False) or getattr(args, 'train_foreach_sgd', False):     # A/B: torch's optimizer over ~300 tensors
        try:
            opt = torch.optim.SGD(head.parameters(), lr=1e-4, momentum=0.9, fused=not getattr(args, 'train_foreach_sgd', False))
        except (TypeError, RuntimeError, ValueError):
            opt = torch.optim.SGD(head.parameters(), lr=1e-4, momentum=0.9)
    else:        # the same update rule in one pass per gradient bucket over flat (parameter, gradient, momentum) ranges (dist.FlatSGD)
        opt = vkn_dist.FlatSGD(reducer, lr=1e-4, momentum=0.9)
    x, pf, mp = synth_inputs(B, device, rank)
    x.requires_grad_(True)                                 # gradients flow on into the backbone in the real model
    g = torch.Generator(device='cpu').manual_seed(4321 + rank)
    Hs, Ws = H * up, W * up
    gt_masks, gt_labels, gt_sem_seg, gt_sem_cls = [], [], [], []
    ys, xs = torch.arange(Hs).view(1, Hs, 1), torch.arange(Ws).view(1, 1, Ws)
    for _ in range(B):                                     # 8 rectangular "things" + the 17 stuff bands per frame
        cy, cx = torch.rand(8, 1, 1, generator=g) * Hs, torch.rand(8, 1, 1, generator=g) * Ws
        hh, ww = torch.rand(8, 1, 1, generator=g) * Hs / 4 + 4, torch.rand(8, 1, 1, generator=g) * Ws / 4 + 4
        gt_masks.append((((ys - cy).abs() < hh) & ((xs - cx).abs() < ww)).float().to(device))
        gt_labels.append(torch.randint(0, CFG2['n_thing'], (8,), generator=g).to(device))
        band = (ys * CFG2['n_stuff'] // Hs).expand(1, Hs, Ws)
        gt_sem_cls.append((torch.arange(CFG2['n_stuff']) + CFG2['n_thing']).to(device))
        gt_sem_seg.append((band == torch.arange(CFG2['n_stuff']).view(-1, 1, 1)).float().to(device))
    prev = torch.randn(B, N, C, 1, 1, generator=g).to(device)
    metas = [dict() for _ in range(B)]

    # the step runs on a stream of its own: the chain graphs are captured on the stream the step runs on, so no parameter gradient
    # crosses streams in backward (KernelUpdateHead.enable_chain_graphs); gradients arrive as tensors and fill their bucket with one
    # multi-tensor copy (BucketedGradAllReducer.zero_grad(set_to_none=True)) instead of one add_ per parameter
    torch.cuda.synchronize()
    train_stream = torch.cuda.Stream(device=device)

    def step():
        if getattr(args, 'train_default_stream', False):    # A/B
            return step_()
        with torch.cuda.stream(train_stream):
            return step_()

    def step_():
        reducer.zero_grad(set_to_none=not getattr(args, 'train_add_grads', False))
        if x.grad is not None:
            x.grad = None
        out = head.forward_train_with_previous(x, pf, mp, None, metas, gt_masks, gt_labels, gt_sem_seg=gt_sem_seg,
                                               gt_sem_cls=gt_sem_cls, previous_obj_feats=prev)
        loss = sum(v for k, v in out[0].items() if 'loss' in k) + 1e-3 * (out[5] ** 2).mean()
        loss.backward()                                    # per-stage buckets are all-reduced (RCCL) as backward produces them
        reducer.finalize()
        opt.step()
        return loss

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 3)):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    dt, per_rank_ms, ranks_seen = _max_over_ranks(dt, args.steps, device, dist_on)
    if rank == 0:
        nparam = sum(p.numel() for p in head.parameters())
        print(json.dumps(dict(metric='training frames/sec (S=3, N=100, 1024x2048, head only)', value=round(world * B * args.steps / dt, 2),
                              unit='frames/s', n_gpus=world, rccl_ranks=(ranks_seen if dist_on else None), per_rank_ms_per_step=per_rank_ms,
                              steps=args.steps, warmup=max(args.warmup, 3),
                              ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True, scaling='weak', vs_baseline=None,
                              dtype='f32', data='synthetic',
                              config=dict(workload='cfg3 video_knet_s3_r50 head training: forward_train_with_previous (losses, GPU '
                                                   'cost matrices + device LSAP), backward through the HIP gather / decode kernels, '
                                                   'the [B*N, C] chains '
                                                   + ('as torch autograd on library GEMMs picked by TunableOp (A/B)' if torch_chain else
                                                      'on the library\'s own kernels in both directions (chain_train.py, vkn_train.hip)')
                                                   + (', launched eagerly' if getattr(args, 'no_chain_graphs', False)
                                                      else ', captured as hipGraphs (forward + backward)')
                                                   + ', per-stage bucketed RCCL gradient all-reduce overlapped with backward, SGD step (momentum 0.9) over the flat gradient buckets',
                                          frames_per_gpu_per_step=B, parallelism=f'frame-sharded dp{world}',
                                          mask_upsample_stride=up, loss_mask_size=[Hs, Ws], fused_loss_tail=bool(getattr(head, '_last_tail_fused', False)),
                                          head_parameters=nparam, last_loss=round(float(loss), 4)),
                              # the parity pin of THIS step at THIS size: the reference's own forward_train_with_previous on the CPU,
                              # same head / channel / kernel counts / feature size / x4, two frames (oracle/gen_golden.py)
                              parity_witness=('tests/golden/train_video_cfg3.npz via tests/test_gpu_train.py::'
                                              'test_forward_train_at_the_benchmarked_cfg3_size_vs_reference_golden' if up == 4 else None))))
    if dist_on:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--settle', type=int, default=300, help='untimed settle steps before the warm-up (profiling runs use few)')
    ap.add_argument('--frames', type=int, default=32,
                    help='frames of the clip per GPU per step (throughput at 8 / 16 / 32 / 64: 5.3k / 6.7k / 7.7k / 8.4k frames/s)')
    ap.add_argument('--clip', type=int, default=0,
                    help='STRONG scaling (BASELINE cfg3 shape): ONE clip of this many frames split into contiguous blocks of clip / N frames '
                         'per rank (clip 8 on 8 GPUs = one frame per GPU, as the reference trains and infers); overrides --frames.  '
                         'Default 0 = weak scaling at --frames per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the extra data points of `breakdown` that launch other batch sizes / several clips (profiling runs)')
    ap.add_argument('--no-upsample', action='store_true', help='skip the x4 upsample output (diagnostic)')
    ap.add_argument('--train-up', type=int, default=4, choices=[1, 2, 4],
                    help='--train: mask_upsample_stride (4 = the shipped video KITTI-STEP config: 512x1024 loss masks; 2 = what rounds 2-5 timed)')
    ap.add_argument('--train-torch-sgd', action='store_true', help='--train A/B: torch.optim.SGD(fused=True) instead of dist.FlatSGD')
    ap.add_argument('--train-foreach-sgd', action='store_true', help='--train A/B: torch.optim.SGD on its foreach path instead of fused=True')
    ap.add_argument('--train-default-stream', action='store_true', help='--train A/B: run the step on the default stream (chain graphs captured on a side stream)')
    ap.add_argument('--train-add-grads', action='store_true', help='--train A/B: zero the gradient buckets and add into their views instead of set_to_none + one batched copy')
    ap.add_argument('--streams', type=int, default=1,
                    help='frame groups of the clip processed on separate HIP streams (measured: no gain, 1 is fastest)')
    ap.add_argument('--x-storage', default='fp32', choices=['fp32', 'fp16', 'bf16'],
                    help='storage type of the feature map x (the head computes in fp32 either way; fp32 = the parity-exact headline)')
    ap.add_argument('--train', action='store_true', help='training step (cfg3) instead of the inference headline; see the docstring')
    ap.add_argument('--torch-chain', action='store_true', help='--train A/B: the [B*N, C] chains as torch autograd on library GEMMs (TunableOp) instead of the library\'s own kernels')
    ap.add_argument('--no-chain-graphs', action='store_true',
                    help='--train: run the [B*N, C] chains as eager torch ops instead of captured hipGraphs (A/B)')
    ap.add_argument('--head', default='ffn', choices=['ffn', 'update'],
                    help="'update': the previous_link heads (video_knet_s3_swin*_joint_update): a rank's block runs in three phases "
                         "around one receive / one send (dist.linked_block_forward); extra measurement, not the BASELINE metric")
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise the RCCL process group and take the multi-rank code path even with ONE rank (tests/test_gpu_rccl.py: '
                         'the distributed step on a 1-GPU box)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or args.force_dist):
        # `python bench.py --gpus N` starts its own N ranks (the reference's launcher does the same from one command:
        # tools/dist_train.sh:7-9): re-exec under torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1
        import socket
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            port = s_.getsockname()[1]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
                                  '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]])
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus N` (self-launching) or with '
                         f'torch.distributed.run --nproc-per-node equal to --gpus')
    if args.clip:
        if args.clip % world != 0:
            raise SystemExit(f'--clip {args.clip} does not split into equal contiguous blocks over {world} ranks')
        args.frames = args.clip // world          # strong scaling: the clip is fixed, the per-rank block shrinks with N
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', device_id=device, rank=rank, world_size=world)

    import vkn_import
    vkn = vkn_import.load()
    from importlib import import_module
    vkn_dist = import_module('video_k_net_amd.dist')
    if args.train:
        return train_main(args, vkn, vkn_dist, device, world, rank, dist_on)
    head = build_head(vkn, device, link=args.head)
    B = args.frames
    x, pf, mp = synth_inputs(B, device, rank)
    XDT = {'fp32': torch.float32, 'fp16': torch.float16, 'bf16': torch.bfloat16}
    x = x.to(XDT[args.x_storage])
    xeb = x.element_size()
    N, C = CFG2['N'], CFG2['C']
    last = head.mask_head[-1]
    dims = last.make_dims(B, N, CFG2['H'], CFG2['W'])
    packs = [h.stage_pack(device) for h in head.mask_head]
    first_prev = torch.zeros(1, N, C, device=device)
    up = 1 if args.no_upsample else CFG2['up']

    NS = max(1, min(args.streams, B))
    bounds = [(B * i) // NS for i in range(NS + 1)]
    groups = [(bounds[i], bounds[i + 1]) for i in range(NS)]
    streams = [torch.cuda.Stream(device=device) for _ in range(NS)] if NS > 1 else [torch.cuda.current_stream(device)]
    gdims = [last.make_dims(b1 - b0, N, CFG2['H'], CFG2['W']) for b0, b1 in groups]
    xs = [x[b0:b1] for b0, b1 in groups]
    pfs = [pf[b0:b1].reshape(b1 - b0, N, C) for b0, b1 in groups]
    mps = [mp[b0:b1] for b0, b1 in groups]
    for p in packs:
        p.ensure_prepared(dims)

    dims1 = last.make_dims(1, N, CFG2['H'], CFG2['W'])

    def step(events=None):
        if args.head == 'update':
            # previous_link heads: phase A (stages 0..S-2 + last gather) on every rank at once, the frame-sequential last-stage
            # chains (phase B) handed from rank to rank with ONE 120 KB receive / send per boundary, phase C (decode, upsample,
            # tracking link) overlapping the next rank's chains
            out = vkn_dist.linked_block_forward(head.linked_block_phases(x, pf, mp), first_prev)
            return out, out[4]
        if dist_on and NS == 1:
            # one process per GPU, contiguous blocks of the clip: the whole block — S stages, x4 upsample, the tracking link of
            # frames 1 .. B-1 to their predecessors (VKN_FLAG_CLIP_LINK) — is ONE C-ABI call, as on one GPU; only frame 0 of the
            # block links across ranks: one neighbour hand-over of the previous rank's last [N x C] kernels (120 KB, point to
            # point) and a one-frame link.  Rank 0's frame 0 links to the clip's `first_prev` inside the call.
            out = vkn.ops.head_forward(dims, packs, x, pfs[0], mp, None, up, clip_first_prev=first_prev, decode_events=events)
            p0 = vkn_dist.neighbour_last_kernels(out[0])
            if p0 is not None:
                # (pinned to the form the block's in-call link ran: more than 16 row tiles = one launch per GEMM — vkn_track_link_flags_f32)
                out[4][0:1].copy_(vkn.ops.track_link(dims1, packs[-1], out[0][0:1], p0,
                                                     flags=vkn.ops.FLAG_CHAIN_LAUNCHES if (B * N + 31) // 32 > 16 else 0))
            return out, out[4]
        if world == 1 and NS == 1:
            # single process: the whole clip step — S stages, x4 upsample, tracking link of every frame to its predecessor — is
            # ONE C-ABI call (VKN_FLAG_CLIP_LINK), so the host side of a step is a handful of allocations
            out = vkn.ops.head_forward(dims, packs, x, pfs[0], mp, None, up, clip_first_prev=first_prev, decode_events=events)
            return out, out[4]
        # every group of frames: S stages + upsample (one C-ABI call per group, each on its own stream) ...
        main = torch.cuda.current_stream(device)
        outs = []
        for gi in range(NS):
            st = streams[gi]
            if NS > 1:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(vkn.ops.head_forward(gdims[gi], packs, xs[gi], pfs[gi], mps[gi], None, up,
                                                 decode_events=events if gi == 0 else None))
        if NS > 1:
            for st in streams:
                main.wait_stream(st)
        cur = torch.cat([o[0] for o in outs], 0) if NS > 1 else outs[0][0]
        # ... then the tracking link over the whole block: prev[b] = obj[b-1]; frame 0 takes the previous rank's last frame
        # (one small RCCL all_gather, 120 KB per rank — the only cross-rank traffic of the clip)
        prev = vkn_dist.previous_kernels_for_block(cur, first_previous=first_prev, all_nonempty=True)
        track = vkn.ops.track_link(dims, packs[-1], cur, prev)
        return outs, track

    def barrier():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # untimed settle phase before the W warm-up steps: a step is a few ms, so a handful of warm-up steps alone would leave
        # the caching allocator, the clocks and the lazily loaded code objects cold on a fresh box (observed: the first ~0.5 s
        # of a fresh process can run 1.5x slower).  Batches of `--settle` / 6 steps are repeated for at least 2 s AND until two
        # consecutive batches agree within 2 % (at most 24 batches); the stop decision is all-reduced so that every rank runs the same number of
        # steps (step() holds a collective when world > 1).
        batch = max(1, args.settle // 6)
        prev_t, stable, t_settle = None, 0, time.perf_counter()
        for _ in range(24 if args.settle > 0 else 0):
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for _ in range(batch):
                out = step()  # keep the result alive exactly like the timed loop does: both generations of output buffers
                              # (1.96 GB x frames/8 each) must exist before timing — a first hipMalloc of that size inside the
                              # timed region cost up to 20 % on a fresh box
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            stable = stable + 1 if (prev_t is not None and abs(tb - prev_t) < 0.02 * tb) else 0
            prev_t = tb
            long_enough = (time.perf_counter() - t_settle) >= (2.0 if args.settle >= 300 else 0.0)
            done = torch.tensor([1 if (stable >= 2 and long_enough) else 0], device=device, dtype=torch.int32)
            if dist_on:
                dist.all_reduce(done, op=dist.ReduceOp.MIN)
            if int(done.item()):
                break
        for _ in range(args.warmup):
            out = step()
        # one HIP-event pair per timed step, recorded by the library around the LAST stage's mask-decode launch on the launch
        # stream (vkn_head_forward_prof_f32): the roofline kernel is timed live, inside the timed region
        dec_events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for e0_, e1_ in dec_events:      # the first record creates the underlying hipEvent_t
            e0_.record(streams[0])
            e1_.record(streams[0])
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = step(dec_events[i])
        barrier()
        dt = time.perf_counter() - t0
        dec_live_ms = sorted(e0_.elapsed_time(e1_) for e0_, e1_ in dec_events)
    dt, per_rank_ms, ranks_seen = _max_over_ranks(dt, args.steps, device, dist_on)
    frames = world * B * args.steps
    ms_per_step = dt / args.steps * 1e3

    extra = {}
    if rank == 0:
        with torch.no_grad():
            # ---- roofline of the dominant x-streaming kernel with an HBM-bound design: k_decode_mfma (last stage), timed LIVE
            # inside the timed steps above (mean of the per-step HIP-event pairs); `isolated_*` = the same kernel in a
            # back-to-back loop of 50 launches (sustained HBM load: lower clocks, what round 1 reported)
            P = CFG2['H'] * CFG2['W']
            kern = torch.randn(B, N, C, device=device)
            hi, lo = vkn.ops.split_planes(kern)
            kb = torch.randn(B, N, device=device)
            outm = torch.empty(B, N, CFG2['H'], CFG2['W'], device=device)
            for _ in range(10):
                vkn.ops.mask_decode_planes(x, hi, lo, N, kb, outm)
            reps = 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                vkn.ops.mask_decode_planes(x, hi, lo, N, kb, outm)
            e1.record()
            torch.cuda.synchronize()
            dec_iso_ms = e0.elapsed_time(e1) / reps
            Bl = B // NS if NS > 1 else B                      # frames of the launch the events bracket
            dec_ms = sum(dec_live_ms) / len(dec_live_ms)
            if args.head == 'update':      # the phased call records no decode events: the isolated loop stands in (extra measurement)
                dec_ms, dec_live_ms = dec_iso_ms, [dec_iso_ms]
            alg = Bl * P * (C * xeb + N * 4)                   # read x once + write the logits once
            ach = alg / (dec_ms * 1e-3) / 1e9
            # HBM bytes per launch from the committed PMC profile of this same command (tools/gpu_profile.sh ->
            # profiles/r01_pmc.json); null when the sidecar is absent or was taken at another batch size
            traffic, mfma_util = None, None
            try:
                side = json.load(open(os.path.join(ROOT, PMC_SIDECAR)))
                if Bl == side.get('_frames_per_launch') and xeb == 4:
                    traffic = side['k_decode_mfma']['hbm_bytes_per_launch']
                    mfma_util = side['k_decode_mfma'].get('mfma_util')
            except Exception:  # noqa: BLE001
                pass
            extra['roofline'] = dict(kernel='k_decode_mfma', bound='hbm', achieved=round(ach, 1), peak=HBM_PEAK_GBS,
                                     unit='GB/s', frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                                     # NOT measured in this run: HBM counters need their own rocprofv3 --pmc passes (never combined with timing);
                                     # the sidecar is the committed counter profile of this same command (tools/gpu_profile.sh), its box / date inside
                                     traffic_source=(('sidecar ' + PMC_SIDECAR + ' (' + str(side.get('_collected', 'date not recorded')) + ')')
                                                     if traffic is not None else None),
                                     mfma_util=mfma_util,   # matrix-pipe busy fraction of this kernel, same sidecar
                                     algorithmic_bytes_per_launch=alg, avg_launch_ms=round(dec_ms, 4),
                                     min_launch_ms=round(dec_live_ms[0], 4), max_launch_ms=round(dec_live_ms[-1], 4),
                                     timed='HIP events recorded by the library around the launch, one pair per timed step',
                                     isolated_loop_launch_ms=round(dec_iso_ms, 4),
                                     isolated_loop_frac=round(B * P * (C * xeb + N * 4) / (dec_iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     frames_per_launch=Bl)
            # the fused decode(s) -> gather(s+1) pass alone (k_fused_il + reduce): it reads x once; reported both against the
            # bytes it actually moves and against the algorithmic bytes of the two ops it replaces (decode: x + logits written,
            # gather: x + logits read — SURVEY.md §8(d)); co-bound by the matrix pipe, see DESIGN.md
            for _ in range(5):
                vkn.ops.decode_gather(x, hi, lo, N, kb)
            e0.record()
            for _ in range(reps):
                vkn.ops.decode_gather(x, hi, lo, N, kb)
            e1.record()
            torch.cuda.synchronize()
            fu_ms = e0.elapsed_time(e1) / reps
            alg = B * P * (C * xeb + N * 4)
            dec_ms = dec_iso_ms                                 # the breakdown below lists isolated-loop timings
            # gather kernel (+ its partial reduce), same accounting: read x once + read the logits once
            for _ in range(3):
                vkn.ops.mask_gather(x, mp)
            e0.record()
            for _ in range(reps):
                vkn.ops.mask_gather(x, mp)
            e1.record()
            torch.cuda.synchronize()
            ga_ms = e0.elapsed_time(e1) / reps
            # head without the x4 upsample output, and the upsample alone
            e0.record()
            for _ in range(5):
                vkn.ops.head_forward(dims, packs, x, pf.reshape(B, N, C), mp, None, 1)
            e1.record()
            torch.cuda.synchronize()
            head_ms = e0.elapsed_time(e1) / 5
            e0.record()
            for _ in range(5):
                vkn.ops.upsample_bilinear(outm, CFG2['up'])
            e1.record()
            torch.cuda.synchronize()
            up_ms = e0.elapsed_time(e1) / 5
            # the two widened rows, timed for the record; they are NOT part of `value`
            P0 = CFG2['N'] - 17
            loc = x if xeb == 4 else x.float()   # the kernel-initialisation pass reads fp32 features
            sem = torch.roll(loc, 1, 0)
            iw = torch.randn(P0, C, 1, 1, device=device) * 0.05
            sw, sb = torch.randn(19, C, 1, 1, device=device) * 0.05, torch.zeros(19, device=device)
            for _ in range(2):
                vkn.ops.kernel_init(loc, sem, iw, sw, sb, 2, True, True)
            e0.record()
            for _ in range(5):
                vkn.ops.kernel_init(loc, sem, iw, sw, sb, 2, True, True)
            e1.record()
            torch.cuda.synchronize()
            init_ms = e0.elapsed_time(e1) / 5
            full = (CFG2['H'] * 8, CFG2['W'] * 8)
            # post-head: structured logits (blobs / bands) — the arg-max kernel's footprint pruning is data dependent
            pc, pl = panoptic_inputs(B, N, P0, 19, CFG2['H'], CFG2['W'], device)
            pan = lambda: vkn.ops.panoptic_joint(pc, pl, P0, 2, P0, 0.25, 0.6, full, full, full, upsample_stride=CFG2['up'])  # noqa: E731
            for _ in range(2):
                pan()
            e0.record()
            for _ in range(5):
                pan()
            e1.record()
            torch.cuda.synchronize()
            pan_ms = e0.elapsed_time(e1) / 5
            # END TO END, what the reference's `simple_test` runs per frame (knet/det/knet.py:161-190; video:
            # knet/video/knet_quansi_dense_embed_fc_joint_train.py:505-560): kernel initialisation (pass 0: the two 1x1 decodes, x = loc + sem,
            # the kernel-init gather) -> the S-stage head on pass 0's OWN outputs (+ the tracking link; no x4 tensor: the post-head
            # kernels resample the low-res logits themselves) -> panoptic merge to the 1024x2048 id map.  Chained on one stream,
            # inputs resident; the head's masks here are whatever the random-init head makes of pass 0's output (timing, not parity —
            # the arg-max pruning is data dependent: `panoptic_joint_1024x2048_ms` above is on segmentation-like logits).
            # The arg-max kernel's work is DATA DEPENDENT (exact footprint pruning: a handful of kernels survive per tile of a segmentation-like
            # map, all 117 of noise) and a random-init head turns pass 0's output into noise, so the merge is timed both ways:
            # `pipeline_ms`: the merge consumes the head's class scores and the segmentation-like logits of `panoptic_joint_1024x2048_ms`
            # (same kernels, same stream order, what a trained head hands over); `pipeline_noise_logits_ms`: the merge on the random-init
            # head's own logits — every kernel everywhere, the worst case of the merge.
            def pipeline(own_logits):
                prop_, xf_, mp_, _ = vkn.ops.kernel_init(loc, sem, iw, sw, sb, 2, True, True, want_seg_preds=False)
                o_ = vkn.ops.head_forward(dims, packs, xf_, prop_, mp_, None, CFG2['up'], want_scaled=False, clip_first_prev=first_prev)
                return vkn.ops.panoptic_joint(o_[1], o_[2] if own_logits else pl, P0, 2, P0, 0.25, 0.6, full, full, full, upsample_stride=CFG2['up'])
            pipe = {False: None, True: None}
            for own in ((False, True) if args.head == 'ffn' else ()):   # (the previous_link head needs its link packs: headline step only)
                for _ in range(2):
                    pipeline(own)
                e0.record()
                for _ in range(5):
                    pipeline(own)
                e1.record()
                torch.cuda.synchronize()
                pipe[own] = e0.elapsed_time(e1) / 5
            pipe_ms, pipe_noise_ms = pipe[False], pipe[True]
            # the whole step (S stages + link + x4 upsample, one C call) at 1 / 8 frames per call — the reference walks a video one
            # frame per call; `value` above is at `--frames` per call
            per_call = {}
            # --clip T at N = 1: the step at T/2, T/4, ... 1 frames per call = what ONE rank of a 2 / 4 / ... / T-GPU run of the same clip
            # computes per step (per-rank compute measured on ONE GPU — NOT a scaling result: no hand-over, no other rank)
            sizes = sorted({max(1, args.clip >> k) for k in range(1, 8)} | {1}) if args.clip else (1, 8)
            for b_ in sizes:
                if b_ >= B or args.head != 'ffn':     # (the previous_link head needs its link packs: only its headline step is timed)
                    continue
                dims_b = last.make_dims(b_, N, CFG2['H'], CFG2['W'])
                xb, pfb, mpb = x[:b_], pf[:b_].reshape(b_, N, C), mp[:b_]
                for _ in range(5):
                    vkn.ops.head_forward(dims_b, packs, xb, pfb, mpb, None, up, clip_first_prev=first_prev)
                e0.record()
                for _ in range(30):
                    vkn.ops.head_forward(dims_b, packs, xb, pfb, mpb, None, up, clip_first_prev=first_prev)
                e1.record()
                torch.cuda.synchronize()
                per_call[f'frames_per_s_at_{b_}_frames_per_call'] = round(b_ / (e0.elapsed_time(e1) / 30 * 1e-3), 1)
                if args.clip:
                    per_call[f'per_rank_step_ms_at_{b_}_frames_ONE_gpu'] = round(e0.elapsed_time(e1) / 30, 4)
                if world == 1 and NS == 1 and not args.no_extras:
                    # the same small calls with FOUR independent videos in flight (call i on HIP stream i % 4): a few-frame call is a
                    # chain of ~45 latency-bound launches that leaves most of the chip idle — another video's call fills it.  For the
                    # record only (throughput of a GPU serving several streams; one call's latency is the line above)
                    sts_b = [torch.cuda.Stream(device=device) for _ in range(4)]
                    for st in sts_b:
                        st.wait_stream(torch.cuda.current_stream(device))
                    keep_b = [None] * 4

                    def run_b(n):
                        for i_ in range(n):
                            with torch.cuda.stream(sts_b[i_ % 4]):
                                keep_b[i_ % 4] = vkn.ops.head_forward(dims_b, packs, xb, pfb, mpb, None, up, clip_first_prev=first_prev)
                    run_b(16)
                    torch.cuda.synchronize()
                    tb_ = time.perf_counter()
                    run_b(80)
                    torch.cuda.synchronize()
                    tb_ = (time.perf_counter() - tb_) / 80
                    per_call[f'frames_per_s_at_{b_}_frames_per_call_4_videos_in_flight'] = round(b_ / tb_, 1)
                    del keep_b, sts_b
            per_call[f'frames_per_s_at_{B}_frames_per_call'] = round(frames / dt, 1)
            if world == 1 and NS == 1 and not args.no_extras:
                # several independent clips in flight (step i on HIP stream i % 4, e.g. one video per stream): the latency-bound update
                # chain of one clip runs while another clip's HBM-bound kernels stream (tools/inflight_probe.py: 1 / 3 / 4 / 6 streams ->
                # 8.66 / 9.13 / 9.62 / 9.64 k frames/s).  For the record only: `value` is ONE step at a time on one stream, and the
                # roofline kernel is timed without a concurrent clip.
                sts = [torch.cuda.Stream(device=device) for _ in range(4)]
                keep = [None] * 4
                for st in sts:
                    st.wait_stream(torch.cuda.current_stream(device))

                def run_inflight(n):
                    for i_ in range(n):
                        with torch.cuda.stream(sts[i_ % 4]):
                            keep[i_ % 4] = vkn.ops.head_forward(dims, packs, x, pfs[0], mp, None, up, clip_first_prev=first_prev)
                run_inflight(12)
                torch.cuda.synchronize()
                th = time.perf_counter()
                run_inflight(40)
                torch.cuda.synchronize()
                th = (time.perf_counter() - th) / 40
                per_call[f'frames_per_s_at_{B}_frames_per_call_4_clips_in_flight'] = round(B / th, 1)
                del keep, sts
                torch.cuda.empty_cache()
            if world == 1 and NS == 1 and xeb == 4 and B == 32 and not args.no_extras:
                # ... and at twice the clip length per call (the update chain is latency-bound in M = B x N rows): for the record only
                try:
                    x2, pf2, mp2 = synth_inputs(2 * B, device, 1)
                    dims2 = last.make_dims(2 * B, N, CFG2['H'], CFG2['W'])
                    pf2 = pf2.reshape(2 * B, N, C)
                    for _ in range(3):
                        o2 = vkn.ops.head_forward(dims2, packs, x2, pf2, mp2, None, up, clip_first_prev=first_prev)
                    e0.record()
                    for _ in range(10):
                        o2 = vkn.ops.head_forward(dims2, packs, x2, pf2, mp2, None, up, clip_first_prev=first_prev)
                    e1.record()
                    torch.cuda.synchronize()
                    per_call[f'frames_per_s_at_{2 * B}_frames_per_call'] = round(2 * B / (e0.elapsed_time(e1) / 10 * 1e-3), 1)
                    del x2, pf2, mp2, o2
                    torch.cuda.empty_cache()
                except RuntimeError:   # out of memory on a shared box: skip the extra point
                    pass
            # the same step with x STORED as fp16 / bf16 (VKN_FLAG_X_F16 / _BF16: fp32 compute, half the x bytes; bit-identical to
            # the fp32 path on the rounded x, tests/test_gpu_xhalf.py) — reported next to the fp32 headline, never as `value`
            variants = {}
            if xeb == 4 and world == 1 and NS == 1 and args.head == 'ffn':
                del loc, sem
                for nm in ('fp16', 'bf16'):
                    xh = x.to(XDT[nm])
                    for _ in range(5):
                        o_ = vkn.ops.head_forward(dims, packs, xh, pfs[0], mp, None, up, clip_first_prev=first_prev)
                    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
                    for a_, b_ in ev:
                        a_.record()
                        b_.record()
                    torch.cuda.synchronize()
                    th = time.perf_counter()
                    for a_, b_ in ev:
                        o_ = vkn.ops.head_forward(dims, packs, xh, pfs[0], mp, None, up, clip_first_prev=first_prev,
                                                  decode_events=(a_, b_))
                    torch.cuda.synchronize()
                    th = (time.perf_counter() - th) / len(ev)
                    dh = sum(a_.elapsed_time(b_) for a_, b_ in ev) / len(ev)
                    algh = B * P * (C * 2 + N * 4)
                    variants[nm] = dict(ms_per_step=round(th * 1e3, 4), frames_per_s=round(B / th, 1), decode_launch_ms=round(dh, 4),
                                        decode_frac_of_hbm_peak=round(algh / (dh * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        decode_algorithmic_bytes=algh)
                    del xh, o_
            per_call['x_storage_variants'] = variants
            if xeb == 4 and world == 1 and NS == 1 and args.head == 'ffn' and up > 1:
                # opt-in: the x4 up-scaled logits stored as fp16 (VKN_FLAG_SCALED_F16: the fp32 interpolation rounded once at the store,
                # |error| <= 2^-11 |logit|, tests/test_gpu_xhalf.py) — half of the 245 MB per frame the step's largest kernel writes.
                # Reported here only, never as `value` (the reference returns fp32 scaled_mask_preds).
                for _ in range(5):
                    o_ = vkn.ops.head_forward(dims, packs, x, pfs[0], mp, None, up, clip_first_prev=first_prev, flags=vkn.ops.FLAG_SCALED_F16)
                torch.cuda.synchronize()
                th = time.perf_counter()
                for _ in range(20):
                    o_ = vkn.ops.head_forward(dims, packs, x, pfs[0], mp, None, up, clip_first_prev=first_prev, flags=vkn.ops.FLAG_SCALED_F16)
                torch.cuda.synchronize()
                th = (time.perf_counter() - th) / 20
                per_call['scaled_output_fp16_variant'] = dict(ms_per_step=round(th * 1e3, 4), frames_per_s=round(B / th, 1),
                                                              tolerance='|error| <= 2^-11 |logit| vs the fp32 output (bit-identical to it rounded to fp16)')
                del o_
            fused_traffic = None
            try:   # counter traffic of the fused kernel from the same sidecar as roofline.traffic (x once + partials written)
                side = json.load(open(os.path.join(ROOT, PMC_SIDECAR)))
                if Bl == side.get('_frames_per_launch') and xeb == 4:
                    fused_traffic = side['k_fused_il']['hbm_bytes_per_launch']
            except Exception:  # noqa: BLE001
                pass
            extra['breakdown'] = dict(**per_call, fused_decode_gather_ms=round(fu_ms, 4),
                                      fused_x_GBps=round(B * P * C * xeb / (fu_ms * 1e-3) / 1e9, 1),
                                      fused_frac_of_hbm_peak=(round(fused_traffic / (fu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                                              if fused_traffic else None),   # counter bytes / isolated-loop time
                                      fused_replaces_algorithmic_GBps=round(2 * alg / (fu_ms * 1e-3) / 1e9, 1),
                                      decode_ms=round(dec_ms, 4), gather_plus_reduce_ms=round(ga_ms, 4),
                                      kernel_init_pass0_ms=round(init_ms, 4), panoptic_joint_1024x2048_ms=round(pan_ms, 4),
                                      pipeline_ms=pipe_ms and round(pipe_ms, 4), pipeline_frames_per_s=pipe_ms and round(B / (pipe_ms * 1e-3), 1),
                                      pipeline_noise_logits_ms=pipe_noise_ms and round(pipe_noise_ms, 4),
                                      pipeline='kernel init (pass 0) -> S-stage head + tracking link -> panoptic merge (1024x2048 id map), chained as simple_test does; the merge on segmentation-like logits (pipeline_ms) / on the random-init head\'s own noise logits (pipeline_noise_logits_ms)',
                                      gather_GBps=round(alg / (ga_ms * 1e-3) / 1e9, 1),
                                      head_3stages_no_upsample_ms=round(head_ms, 4),
                                      upsample_x4_ms=round(up_ms, 4),
                                      upsample_write_GBps=round(B * N * P * 16 * 4 / (up_ms * 1e-3) / 1e9, 1),
                                      frames_per_s_no_upsample=round(B / (head_ms * 1e-3), 1))
        if world == 1 and not args.no_cpu_baseline:
            extra['cpu_baseline'] = cpu_baseline(head.state_dict())

    if rank == 0:
        line = dict(metric='frames/sec (S=3, N=100, 1024x2048)', value=round(frames / dt, 2), unit='frames/s',
                    n_gpus=world, rccl_ranks=(ranks_seen if dist_on else None), per_rank_ms_per_step=per_rank_ms,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 4),
                    higher_is_better=True, scaling='strong' if args.clip else 'weak', vs_baseline=None, dtype='f32', data='synthetic',
                    config=dict(workload=('cfg3 shape: ONE clip of %d frames in contiguous blocks of %d per GPU; ' % (args.clip, B) if args.clip else '')
                                         + 'cfg2 video_knet_s3_r50: VideoKernelIterHead S=3, N=100 proposals + 17 stuff = 117 '
                                         'kernels, C=256, 1024x2048 frame -> 128x256 stride-8 features, '
                                         + ('ffn tracking link, ' if args.head == 'ffn' else
                                            'previous_link=update_dynamic_cov + previous_type=update (NOT the BASELINE head: the last '
                                            'stage is frame-sequential, phases A/B/C per rank), ')
                                         + 'x4 bilinear upsample of the final logits' + (' [SKIPPED]' if args.no_upsample else ''),
                                frames_per_gpu_per_step=B, clip_frames=(args.clip or None), streams_per_gpu=NS,
                                parallelism=f'frame-sharded dp{world}', x_storage=args.x_storage,
                                arithmetic=('fp32 storage;' if xeb == 4 else args.x_storage + ' storage of x, fp32 everything else;') + ' gather/decode on f16 hi+lo split MFMA, [N x C] GEMMs on the same two-term f16 split in the persistent form (>= 22 row tiles: this workload at >= 6 frames per call) and on bf16x3 split MFMA in the few-row / launch-per-GEMM forms, '
                                           'fp32 accumulate everywhere (fp32-class accuracy, DESIGN.md §3); random-init weights'),
                    **extra)
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
