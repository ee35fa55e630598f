"""CPU, world_size 2 on gloo: the frame sharding + neighbour exchange of the clip path."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vkn_import
    vkn = vkn_import.load()
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    N, C = 5, 8
    frames = torch.arange(T, dtype=torch.float32).reshape(T, 1, 1).expand(T, N, C) + 100.0      # kernels of frame t == 100 + t
    b0, b1 = d.shard_bounds(T, world, rank)
    first_prev = torch.full((1, N, C), -1.0)
    prev = d.previous_kernels_for_block(frames[b0:b1].contiguous(), first_previous=first_prev)
    if b1 > b0 and T >= world:   # every rank owns frames: the sync-free variant must give the same answer
        prev2 = d.previous_kernels_for_block(frames[b0:b1].contiguous(), first_previous=first_prev, all_nonempty=True)
        assert torch.equal(prev, prev2)
    q.put((rank, b0, b1, prev[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()
    assert vkn is not None


def _run(T, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    return out


def test_clip_is_sharded_contiguously_and_previous_kernels_cross_ranks():
    out = _run(T=8)
    covered = []
    for rank, b0, b1, prev in out:
        covered += list(range(b0, b1))
        want = [(-1.0 if t == 0 else 100.0 + t - 1) for t in range(b0, b1)]          # prev of frame t = kernels of frame t-1
        assert prev == want, (rank, prev, want)
    assert covered == list(range(8))


def test_uneven_and_tiny_clips():
    out = _run(T=3)
    assert [o[1:3] for o in out] == [(0, 1), (1, 3)]
    assert out[1][3] == [100.0, 101.0]
    out = _run(T=1)                      # rank 0 owns nothing, rank 1 owns frame 0 and must fall back to first_previous
    assert out[0][1:3] == (0, 0) and out[1][1:3] == (0, 1) and out[1][3] == [-1.0]


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vkn_import
    vkn = vkn_import.load()
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    _updator_torch = import_module('video_k_net_amd.kernel_update_head')._updator_torch
    torch.manual_seed(0)                                    # identical replicas
    C, rows = 32, 16
    net = torch.nn.ModuleDict({'mask_head': torch.nn.ModuleList(
        [vkn.KernelUpdator(in_channels=C, feat_channels=C, out_channels=C) for _ in range(3)])})
    red = d.BucketedGradAllReducer(net)
    assert sorted(b['key'] for b in red.buckets) == ['stage0', 'stage1', 'stage2']
    g = torch.Generator().manual_seed(7)
    u, k = torch.randn(world * rows, C, generator=g), torch.randn(world * rows, 1, C, generator=g)

    def loss_of(u_, k_):
        h = k_
        for m in net['mask_head']:                           # the head's own torch chain (kernel_update_head._updator_torch), CPU
            h = _updator_torch(m, u_, h)
        return (h ** 2).mean()

    for _ in range(2):                                       # two steps: zero_grad re-arms the hooks
        red.zero_grad()
        loss_of(u[rank * rows:(rank + 1) * rows], k[rank * rows:(rank + 1) * rows]).backward()
        red.finalize()
    got = {n: p.grad.clone() for n, p in net.named_parameters()}
    if rank == 0:                                            # single-process reference on the whole batch
        ref_net = torch.nn.ModuleDict({'mask_head': torch.nn.ModuleList(
            [vkn.KernelUpdator(in_channels=C, feat_channels=C, out_channels=C) for _ in range(3)])})
        ref_net.load_state_dict(net.state_dict())
        h = k
        for m in ref_net['mask_head']:
            h = _updator_torch(m, u, h)
        (h ** 2).mean().backward()
        err = max(float((got[n] - p.grad).abs().max() / (p.grad.abs().max() + 1e-12)) for n, p in ref_net.named_parameters())
        q.put(err)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_allreduce_equals_full_batch_gradients():
    """2 ranks x half the rows, per-stage buckets reduced asynchronously from backward hooks == 1 rank x all rows."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert err < 1e-5, err


def _robust_worker(rank, world, port, q):
    """torch's default `zero_grad()` (set_to_none) must not cut the reducer off from the gradients; gradient
    accumulation under `no_sync()`; buckets with never-used members overlap from the second step on; misuse raises."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vkn_import
    vkn_import.load()
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    torch.manual_seed(0)

    class Stage(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.used = torch.nn.Linear(8, 8)
            self.unused = torch.nn.Linear(8, 8)          # like the link modules of the non-last video stages

    net = torch.nn.ModuleDict({'mask_head': torch.nn.ModuleList([Stage(), Stage()])})
    red = d.BucketedGradAllReducer(net)
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    g = torch.Generator().manual_seed(3)
    data = torch.randn(4, world * 6, 8, generator=g)       # 4 micro-batches of world * 6 rows

    def loss_of(xb):
        h = xb
        for st in net['mask_head']:
            h = torch.tanh(st.used(h))
        return (h ** 2).sum()

    def full_grads(batches):
        ref = {n: torch.zeros_like(p) for n, p in net.named_parameters()}
        for xb in batches:
            gs = torch.autograd.grad(loss_of(xb), [p for n, p in net.named_parameters() if '.used.' in n])
            for (n, _), gi in zip([(n, p) for n, p in net.named_parameters() if '.used.' in n], gs):
                ref[n] += gi
        return {n: v / world for n, v in ref.items()}

    res = {}
    mine = lambda xb: xb[rank * 6:(rank + 1) * 6]  # noqa: E731
    # step 1: torch's DEFAULT zero_grad (drops the .grad views), one backward
    opt.zero_grad()
    assert all(p.grad is None for p in net.parameters())
    loss_of(mine(data[0])).backward()
    red.finalize()
    ref = full_grads([data[0]])
    res['default_zero_grad'] = max(float((p.grad - ref[n]).abs().max()) for n, p in net.named_parameters() if p.grad is not None)
    assert all((p.grad is None) == ('.unused.' in n) for n, p in net.named_parameters())
    assert all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for b in red.buckets for p, v in zip(b['params'], b['views']))
    # step 2: buckets now know that `unused` never fires -> their all-reduce starts inside backward
    red.zero_grad()
    loss_of(mine(data[1])).backward()
    res['overlap'] = all(b['handle'] is not None for b in red.buckets) if world > 1 else True
    red.finalize()
    ref = full_grads([data[1]])
    res['step2'] = max(float((p.grad - ref[n]).abs().max()) for n, p in net.named_parameters())
    # step 3: accumulation of two micro-batches, the first under no_sync(); default zero_grad again
    opt.zero_grad()
    with red.no_sync():
        loss_of(mine(data[2])).backward()
    loss_of(mine(data[3])).backward()
    red.finalize()
    ref = full_grads([data[2], data[3]])
    res['accumulate'] = max(float((p.grad - ref[n]).abs().max()) for n, p in net.named_parameters() if p.grad is not None)
    # misuse: a second synchronised backward before finalize() must raise, not corrupt an in-flight bucket
    red.zero_grad()
    loss_of(mine(data[0])).backward()
    try:
        loss_of(mine(data[1])).backward()
        res['misuse_raises'] = world == 1
    except RuntimeError:
        res['misuse_raises'] = True
    red.finalize()
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_survives_default_zero_grad_accumulation_and_unused_members():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_robust_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res['default_zero_grad'] < 1e-5 and res['step2'] < 1e-5 and res['accumulate'] < 1e-5, res
    assert res['overlap'] and res['misuse_raises'], res


def _bulk_worker(rank, world, port, q):
    """Parameter gradients delivered OUTSIDE autograd's accumulation nodes (what a captured chain graph does: `_ChainGraphRunner`)
    mixed with eager ones in the same bucket, one parameter reporting both ways — through `BucketedGradAllReducer.params_ready`:
    averaged gradients equal the plain computation, in both zero_grad modes, and the buckets are launched inside backward from the
    second step on (report counts were learned in the first)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vkn_import
    vkn_import.load()
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    torch.manual_seed(0)

    class Outside(torch.autograd.Function):          # y = x @ w.T + b with w, b NOT inputs of the function
        @staticmethod
        def forward(ctx, x, stage):
            ctx.stage = stage
            ctx.save_for_backward(x)
            return x @ stage.w.detach().t() + stage.b.detach()

        @staticmethod
        def backward(ctx, gy):
            (x,) = ctx.saved_tensors
            st = ctx.stage
            for p, g in ((st.w, gy.t() @ x), (st.b, gy.sum(0))):
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            st.on_param_grads([st.w, st.b])
            return gy @ st.w.detach(), None

    class Stage(torch.nn.Module):
        on_param_grads = None

        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(8, 8) * 0.3)
            self.b = torch.nn.Parameter(torch.zeros(8))
            self.eager = torch.nn.Linear(8, 8)

        def forward(self, x):
            h = torch.tanh(Outside.apply(x, self))
            return torch.tanh(self.eager(h)) + 0.1 * (h @ self.w.t())      # `w` is ALSO used eagerly: it reports twice

        def plain(self, x):
            h = torch.tanh(x @ self.w.t() + self.b)
            return torch.tanh(self.eager(h)) + 0.1 * (h @ self.w.t())

    net = torch.nn.ModuleDict({'mask_head': torch.nn.ModuleList([Stage(), Stage()])})
    red = d.BucketedGradAllReducer(net)
    assert all(st.on_param_grads is not None for st in net['mask_head'])
    g = torch.Generator().manual_seed(5)
    data = torch.randn(4, world * 6, 8, generator=g)

    def run(xb, plain):
        h = xb.clone().requires_grad_(True)          # (a function none of whose inputs needs a gradient is never differentiated: the
        for st in net['mask_head']:                  #  chain graphs fall back to the eager chain in that case, KernelUpdateHead._chain)
            h = st.plain(h) if plain else st(h)
        return (h ** 2).sum()

    res = dict(err=0.0, overlap=True)
    for step, mode in enumerate((False, True, True, False)):
        params = [p for _, p in net.named_parameters()]
        ref = [gi / world for gi in torch.autograd.grad(run(data[step], True), params)]
        red.zero_grad(set_to_none=mode)
        run(data[step][rank * 6:(rank + 1) * 6], False).backward()
        if step >= 1:
            res['overlap'] = res['overlap'] and all(b['handle'] is not None for b in red.buckets)
        red.finalize()
        res['err'] = max([res['err']] + [float((p.grad - r).abs().max()) for p, r in zip(params, ref)])
        assert all(p.grad.data_ptr() == v.data_ptr() for b in red.buckets for p, v in zip(b['params'], b['views']))
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_takes_gradients_delivered_outside_autograd():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bulk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res['err'] < 1e-5 and res['overlap'], res


# ---- previous_link heads: the phase split of a sharded clip
def _stub_phases(x_block, log):
    """A stand-in for `VideoKernelIterHead.linked_block_phases` with the same data flow, in plain torch on the CPU:
       A: y[t] = tanh(x[t] W)                                  (per frame, no cross-frame input)
       B: k[t] = tanh(y[t] + 0.5 * roll(k[t-1])) , k[-1] = prev (the frame-sequential last-stage chain)
       C: out[t] = k[t] * y[t] + k[t-1]                         (per frame + the tracking link to the previous kernels)"""
    torch.manual_seed(5)
    Wm = torch.randn(x_block.shape[-1], x_block.shape[-1]) * 0.3
    st = {}

    def run(name, prev):
        log.append(name)
        if name == 'A':
            assert prev is None
            st['y'] = torch.tanh(x_block @ Wm)
            return None
        if name == 'B':
            ks, k = [], prev[0]
            for t in range(x_block.shape[0]):
                k = torch.tanh(st['y'][t] + 0.5 * torch.roll(k, 1, 0))
                ks.append(k)
            st['k'] = torch.stack(ks)
            return st['k']
        before = torch.cat([prev, st['k'][:-1]])
        return st['k'] * st['y'] + before
    return run


def _linked_worker(rank, world, port, T, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import vkn_import
    vkn_import.load()
    from importlib import import_module
    d = import_module('video_k_net_amd.dist')
    N, C = 5, 8
    g = torch.Generator().manual_seed(77)
    x = torch.randn(T, N, C, generator=g)
    first = torch.randn(1, N, C, generator=g)
    b0, b1 = d.shard_bounds(T, world, rank)
    log = []
    out = d.linked_block_forward(_stub_phases(x[b0:b1], log), first)
    q.put((rank, b0, b1, out.numpy(), log))     # (by value: a torch tensor travels as a file descriptor the child must outlive)
    dist.barrier()
    dist.destroy_process_group()


def test_previous_link_clip_sharded_in_phases_equals_the_whole_clip():
    """`dist.linked_block_forward` over 2 and 3 ranks (gloo) with a stub chain of the head's data flow: the sharded clip — phase A on
    every rank at once, the sequential phase B handed from rank to rank with ONE receive and ONE send per boundary, phase C —
    equals the single-process clip, and every rank ran A, B, C in that order."""
    from importlib import import_module
    sys.path.insert(0, ROOT)
    import vkn_import
    vkn_import.load()
    d = import_module('video_k_net_amd.dist')
    T, N, C = 7, 5, 8
    g = torch.Generator().manual_seed(77)
    x = torch.randn(T, N, C, generator=g)
    first = torch.randn(1, N, C, generator=g)
    whole = d.linked_block_forward(_stub_phases(x, []), first)       # no process group: one rank
    for world in (2, 3):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_linked_worker, args=(r, world, port, T, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        got = torch.cat([torch.from_numpy(r[3]) for r in res])
        assert [r[1] for r in res] == [(T * k) // world for k in range(world)] and res[-1][2] == T
        assert torch.equal(got, whole), f'world {world}: max err {float((got - whole).abs().max())}'
        assert all(r[4] == ['A', 'B', 'C'] for r in res)
