"""GPU (-m gpu): the RCCL path on ONE GPU.  `backend='nccl'` IS RCCL on ROCm; the multi-GPU runs are the driver's, but a process
group of world_size 1 already goes through RCCL's communicator setup, its stream ordering against the compute stream, the async
all-reduce handles of `BucketedGradAllReducer` and the batched point-to-point step of `neighbour_last_kernels` — everything except
the xGMI transport itself.  Also: `bench.py` launched the way the driver launches the N-GPU runs (torch.distributed.run, one rank)
prints the same metric as the plain run."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope='module')
def rccl_group():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
    yield
    dist.destroy_process_group()


def test_bucketed_reducer_through_rccl(vkn, rccl_group):
    """Per-stage buckets all-reduced asynchronously from the backward hooks over RCCL (sum over ONE rank = identity): gradients of
    the head's own torch chain are unchanged, the collectives really were issued, and the step order zero_grad -> backward ->
    finalize -> optimizer works with torch's default zero_grad."""
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    _updator_torch = import_module('video_k_net_amd.kernel_update_head')._updator_torch
    torch.manual_seed(0)
    C, rows = 64, 48
    net = torch.nn.ModuleDict({'mask_head': torch.nn.ModuleList(
        [vkn.KernelUpdator(in_channels=C, feat_channels=C, out_channels=C) for _ in range(3)])}).to(DEV)
    red = d.BucketedGradAllReducer(net, force_collectives=True)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    u, k = torch.randn(rows, C, device=DEV), torch.randn(rows, 1, C, device=DEV)

    def loss_of():
        h = k
        for m in net['mask_head']:
            h = _updator_torch(m, u, h)
        return (h ** 2).mean()

    ref = torch.autograd.grad(loss_of(), list(net.parameters()))
    for zero in (red.zero_grad, opt.zero_grad, red.zero_grad):
        zero()
        loss_of().backward()
        launched = [b['handle'] is not None for b in red.buckets]
        red.finalize()
        opt.step()
        for p, g in zip(net.parameters(), ref):
            assert torch.equal(p.grad, g)
    assert all(launched), 'from the second step on every stage bucket starts its all-reduce inside backward'
    t = torch.ones(1 << 20, device=DEV)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert float(t.sum()) == float(1 << 20)


def test_neighbour_exchange_through_rccl(vkn, rccl_group):
    """The 120 KB hand-over of a block's last kernels as ONE batched isend / irecv — with one rank the neighbour is the rank
    itself (RCCL pairs the two within the group) — followed by the one-frame link on the compute stream."""
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    N, C = 117, 256
    block = torch.randn(4, N, C, device=DEV)
    assert d.neighbour_last_kernels(block) is None            # world 1: no neighbour
    recv = torch.empty(N, C, device=DEV)
    d.exchange(block[-1].contiguous(), recv, send_to=0, recv_from=0)
    torch.cuda.synchronize()
    assert torch.equal(recv, block[-1])
    prev = d.previous_kernels_for_block(block, first_previous=torch.zeros(1, N, C, device=DEV))
    assert torch.equal(prev[1:], block[:-1]) and float(prev[0].abs().max()) == 0.0


def test_previous_link_block_in_phases_through_rccl(vkn, rccl_group):
    """`dist.linked_block_forward` on a previous_link head with a real RCCL process group (one rank: no neighbour, but the phases run
    between initialised-communicator calls on the GPU) equals `clip_forward`; the hand-over message itself — the block's last kernels
    as ONE batched isend / irecv, here to the rank itself — arrives intact and phase B started from it reproduces the clip."""
    from importlib import import_module
    from helpers import load_golden
    from test_gpu_parity import _build_head
    d = import_module('video_k_net_amd.dist')
    g, case = load_golden('video_upd_cfg')
    head, _ = _build_head(vkn, case)
    T, N, C, H, W = 4, case['N'], case['C'], case['H'], case['W']
    gen = torch.Generator().manual_seed(3)
    xs = torch.randn(T, C, H, W, generator=gen).to(DEV)
    pfs = torch.randn(T, N, C, 1, 1, generator=gen).to(DEV)
    mps = (torch.randn(T, N, H, W, generator=gen) * 4).to(DEV)
    first = torch.randn(1, N, C, generator=gen).to(DEV)
    with torch.no_grad():
        whole = head.clip_forward(xs, pfs, mps, first_previous_obj_feats=first.reshape(1, N, C, 1, 1))
        out = d.linked_block_forward(head.linked_block_phases(xs, pfs, mps), first)
        for k in range(5):
            assert torch.equal(out[k].reshape(whole[k].shape), whole[k]), k
        # two blocks, the hand-over travelling through RCCL (send to / receive from rank 0 = this rank)
        run0 = head.linked_block_phases(xs[:2], pfs[:2], mps[:2])
        run0('A', None)
        k0 = run0('B', first)
        recv = torch.empty(1, N, C, device=DEV)
        d.exchange(k0[-1:].contiguous().reshape(1, N, C), recv, send_to=0, recv_from=0)
        out0 = run0('C', first)
        run1 = head.linked_block_phases(xs[2:], pfs[2:], mps[2:])
        run1('A', None)
        run1('B', recv)
        out1 = run1('C', recv)
        for k in range(5):
            assert torch.equal(torch.cat([out0[k], out1[k]], 0).reshape(whole[k].shape), whole[k]), k


def _bench(args, launcher=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('MASTER_PORT', None)
    cmd = [sys.executable, *launcher, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '40', '--warmup', '10', '--settle', '60',
           '--no-cpu-baseline', '--no-extras', *args]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])


def test_bench_under_torchrun_one_rank_matches_plain_run():
    """`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 --force-dist`: RCCL initialised, the multi-rank
    step (clip link in the call + neighbour exchange + barrier + max-over-ranks all-reduce of the time) — same value as the plain
    single-process run within run-to-run noise."""
    plain = _bench([])
    port = _free_port()
    dist_line = _bench(['--force-dist'], launcher=['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
                                                 '127.0.0.1', '--master-port', str(port)])
    assert dist_line['n_gpus'] == 1 and dist_line['metric'] == plain['metric']
    ratio = dist_line['value'] / plain['value']
    print(f'plain {plain["value"]} frames/s, torchrun + RCCL (1 rank) {dist_line["value"]} frames/s, ratio {ratio:.3f}')
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'r03_rccl_one_rank.json'), 'w') as f:
        json.dump(dict(plain=plain, torchrun_force_dist=dist_line, ratio=ratio), f, indent=1)
    assert 0.93 < ratio < 1.07, (plain['value'], dist_line['value'])

    train = _bench(['--train', '--force-dist', '--steps', '6', '--warmup', '3'],
                   launcher=['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                             '--master-port', str(_free_port())])
    assert train['n_gpus'] == 1 and train['value'] > 0


def test_bench_previous_link_head_under_torchrun_one_rank():
    """`bench.py --head update` (the previous_link heads: phases A / B / C around one receive / one send per rank boundary) as the driver
    would launch it on N GPUs — here one rank, RCCL initialised: it runs and reports the metric of the phased clip step."""
    port = _free_port()
    line = _bench(['--force-dist', '--head', 'update', '--frames', '8'],
                  launcher=['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                            '--master-port', str(port)])
    assert line['n_gpus'] == 1 and line['value'] > 0 and 'previous_link=update_dynamic_cov' in line['config']['workload']
    assert line['config']['frames_per_gpu_per_step'] == 8



def test_bench_clip_of_8_strong_scaling_under_torchrun_one_rank():
    """`bench.py --clip 8` (BASELINE cfg3 shape: ONE clip of 8 frames in contiguous blocks of 8 / N per rank, strong scaling) launched
    the way the driver launches N-GPU runs — here one rank, RCCL initialised: the line says "strong", carries the whole clip on the one
    rank, and the N = 1 breakdown holds the per-rank step times at 4 / 2 / 1 frames per call (what a rank of a 2 / 4 / 8-GPU run computes)."""
    port = _free_port()
    line = _bench(['--force-dist', '--clip', '8'],
                  launcher=['-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                            '--master-port', str(port)])
    assert line['scaling'] == 'strong' and line['n_gpus'] == 1 and line['value'] > 0
    assert line['config']['clip_frames'] == 8 and line['config']['frames_per_gpu_per_step'] == 8
    bd = line['breakdown']
    for b in (4, 2, 1):
        assert bd[f'per_rank_step_ms_at_{b}_frames_ONE_gpu'] > 0
    assert bd['per_rank_step_ms_at_1_frames_ONE_gpu'] < bd['per_rank_step_ms_at_4_frames_ONE_gpu']


def test_bench_self_launches_its_ranks_from_the_bare_command():
    """`python bench.py --gpus 1 --force-dist` with no launcher and no WORLD_SIZE in the environment: bench.py re-executes itself
    under torch.distributed.run (the path `python bench.py --gpus 8` takes; reference: one command starts every rank,
    tools/dist_train.sh:7-9), RCCL is initialised, and the one JSON line reports the rank count the process group saw and every
    rank's own step time.  Also `--clip 8` and `--train` through the same path."""
    env_keys = [k for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR') if k in os.environ]
    saved = {k: os.environ.pop(k) for k in env_keys}
    try:
        line = _bench(['--force-dist'])
        assert line['n_gpus'] == 1 and line['rccl_ranks'] == 1 and len(line['per_rank_ms_per_step']) == 1 and line['value'] > 0
        assert abs(line['per_rank_ms_per_step'][0] - line['ms_per_step']) < 1e-3 * line['ms_per_step'] + 1e-3
        clip = _bench(['--force-dist', '--clip', '8'])
        assert clip['scaling'] == 'strong' and clip['rccl_ranks'] == 1
        train = _bench(['--force-dist', '--train', '--steps', '6', '--warmup', '3'])
        assert train['rccl_ranks'] == 1 and train['value'] > 0 and train['parity_witness']
    finally:
        os.environ.update(saved)


def test_flat_sgd_equals_torch_sgd(vkn):
    """`dist.FlatSGD` (one `vkn_sgd_momentum_f32` pass per gradient bucket over flat parameter / gradient / momentum ranges) against
    `torch.optim.SGD(momentum, weight_decay)` on a copy of the same module over four steps: the same parameters to fp32 rounding, the
    parameters still views of the flat buffers, a parameter without a gradient treated as a zero gradient."""
    import copy
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.LayerNorm(53), torch.nn.Linear(53, 11, bias=False)).to('cuda:0')
    ref = copy.deepcopy(net)
    red = vkn.dist.BucketedGradAllReducer(net, bucket_of=lambda name: 'a' if name.startswith('0') else 'b')
    opt = vkn.dist.FlatSGD(red, lr=0.05, momentum=0.9, weight_decay=1e-3)
    topt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3)
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.equal(a, b)
    for step in range(4):
        x = torch.randn(19, 37, device='cuda:0')
        red.zero_grad(set_to_none=True)
        topt.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        ref(x).square().mean().backward()
        red.finalize()
        opt.step()
        topt.step()
        for a, b in zip(net.parameters(), ref.parameters()):
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), step
    ranges = [(st['param'].data_ptr(), st['param'].data_ptr() + 4 * st['param'].numel()) for st in opt.state]
    assert all(any(lo <= p.data_ptr() < hi for lo, hi in ranges) for p in net.parameters())
    # a member without a gradient: zero gradient (momentum decays, weight decay applies) — not skipped
    red.zero_grad(set_to_none=True)
    before = [p.detach().clone() for p in net.parameters()]
    net[0](torch.randn(5, 37, device='cuda:0')).square().mean().backward()        # only the first Linear takes part
    red.finalize()
    opt.step()
    assert not torch.equal(before[-1], list(net.parameters())[-1])
