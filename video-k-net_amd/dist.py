"""Frame-sharded data parallelism for a clip (one process per GPU, `torch.distributed`; backend "nccl" = RCCL on ROCm).

Masks, class scores and kernels of frame t do not depend on frame t-1; only the tracking embedding of
the last stage does (knet/video/kernel_iter_head.py:544-546, knet/video/kernel_update_head.py:394-415).  So the frames of a clip
are split into contiguous blocks, every rank runs the whole head on its block with NO collective on the mask path, and a single
small exchange hands each rank the final kernels of the frame just before its block (120 KB for N=117, C=256).
"""
import torch
import torch.distributed as dist


def shard_bounds(num_frames: int, world: int, rank: int):
    """Contiguous block [begin, end) of a clip's frames owned by `rank` (sizes differ by at most 1)."""
    return (num_frames * rank) // world, (num_frames * (rank + 1)) // world


def exchange(send, recv, send_to=None, recv_from=None, group=None):
    """One batched point-to-point step: `send` -> group rank `send_to`, `recv` <- group rank `recv_from` (either may be None).
    P2POp peers are GLOBAL ranks: group-local ranks are translated (identical for the default group).  A rank may name itself
    (RCCL pairs a send and a receive posted in one group): that is how a 1-GPU box exercises this path."""
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    ops = []
    if send_to is not None:
        ops.append(dist.P2POp(dist.isend, send, peer(send_to), group))
    if recv_from is not None:
        ops.append(dist.P2POp(dist.irecv, recv, peer(recv_from), group))
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()


def neighbour_last_kernels(block_kernels: torch.Tensor, group=None):
    """Every rank owns >= 1 frame: hand this rank's LAST frame's kernels [N, C] to rank + 1 and return the previous rank's
    ([1, N, C]; None on rank 0 or without a process group).  ONE point-to-point send / receive (120 KB at N = 117, C = 256) —
    xGMI is point-to-point, a single send rides one link, no ring, no host synchronisation."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world <= 1:
        return None
    N, C = block_kernels.shape[1:]
    last = block_kernels[-1].contiguous()
    recv = torch.empty_like(last)
    exchange(last, recv, rank + 1 if rank + 1 < world else None, rank - 1 if rank > 0 else None, group)
    return recv.reshape(1, N, C) if rank > 0 else None


def previous_kernels_for_block(block_kernels: torch.Tensor, first_previous=None, group=None, all_nonempty=False):
    """`block_kernels` [T_r, N, C] = final kernels of this rank's frames.  Returns prev [T_r, N, C] with
    prev[i] = kernels of the frame BEFORE frame i of the block: the previous rank's last frame for i = 0
    (rank 0: `first_previous` [1,N,C], or its own frame 0 when the clip starts the video), own frame i-1 otherwise.
    `all_nonempty=True` (every rank is known to own frames, e.g. an even split — what bench.py and training use): ONE
    point-to-point send to rank + 1 / receive from rank - 1, no host synchronisation.  Otherwise (uneven splits, ranks without
    frames): one all_gather of [N, C] + a has-frames flag per rank and one host read of the flags; ranks with an empty block are
    skipped by their successor."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = block_kernels.shape[0]
    N, C = block_kernels.shape[1:]
    if world > 1 and all_nonempty:
        p0 = neighbour_last_kernels(block_kernels, group)
    elif world > 1:
        last = block_kernels[-1].contiguous() if T > 0 else block_kernels.new_zeros(N, C)
        has = torch.tensor([1.0 if T > 0 else 0.0], device=block_kernels.device)
        payload = torch.cat([last.reshape(-1), has])             # one message per rank: [N*C kernels | has-frames flag]
        got = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(got, payload, group=group)
        p0 = None
        flags = torch.stack([g_[-1] for g_ in got]).cpu()   # ONE device -> host read for all flags (uneven splits only)
        for r in range(rank - 1, -1, -1):        # nearest predecessor that owns at least one frame
            if float(flags[r]) > 0:
                p0 = got[r][:-1].reshape(1, N, C)
                break
    else:
        p0 = None
    if p0 is None:
        p0 = first_previous if first_previous is not None else block_kernels[:1]
    if T == 0:
        return block_kernels
    return torch.cat([p0.reshape(1, N, C), block_kernels[:-1]], dim=0)


def linked_block_forward(run_phase, first_previous, group=None):
    """One rank's share of a clip for the heads whose LAST stage rewrites its incoming kernels from the previous frame's final
    kernels (`previous_link = "update_dynamic_cov" | "link_atten"`, knet/video/kernel_update_head.py:324-372,
    knet/video/kernel_iter_head.py:454-456).  There the masks of frame t depend on frame t-1 — through [N x C]-sized state only,
    and only in the last stage — so a rank cannot simply run its block and patch frame 0 afterwards (`neighbour_last_kernels`
    does that for the heads without previous_link).  The block runs in three phases around ONE receive and ONE send:

        A  `run_phase('A', None)`     stages 0 .. S-2 and the last stage's gather for all frames of the block: no cross-frame input,
                                      every rank runs it at once (the x-streaming work);
           receive the final kernels of the frame before the block from rank - 1 (rank 0: `first_previous`);
        B  `run_phase('B', prev)`     the last stage's [N x C] chains, frame by frame, frame 0 linked to `prev` -> returns the
                                      block's final kernels [T_r, N, C];
           send the LAST frame's kernels to rank + 1 — before the HBM-bound tail, so the next rank's chains start while this
           rank decodes and upsamples;
        C  `run_phase('C', prev)`     the batched last decode, the upsample, the tracking link -> the block's outputs.

    The last-stage chains of a clip therefore serialise across the ranks (~0.13 ms per frame at one frame per launch) while
    phases A and C of all ranks overlap; the cross-rank traffic is one 120 KB point-to-point message per rank boundary.
    `run_phase(name, prev)` is `VideoKernelIterHead.linked_block_phases(...)` on the GPU, any callable with that contract in tests.
    Every rank must own at least one frame.  -> what phase C returns."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    run_phase('A', None)
    prev = first_previous
    if world > 1 and rank > 0:
        prev = torch.empty_like(first_previous)
        exchange(None, prev, None, rank - 1, group)
    kernels = run_phase('B', prev)
    if world > 1 and rank + 1 < world:
        exchange(kernels[-1:].contiguous().reshape(first_previous.shape), None, rank + 1, None, group)
    return run_phase('C', prev)


class BucketedGradAllReducer:
    """Data-parallel gradient averaging for the head (BASELINE cfg3: frames of a clip sharded over the GPUs of a node, RCCL
    all-reduce over xGMI; the reference gets this from mmcv's `MMDistributedDataParallel`, external/train.py:53-61).

    Parameters are grouped into buckets (default: one per `mask_head.{s}` stage + one for the rest).  Every parameter's `.grad` is
    a VIEW into its bucket's flat fp32 buffer, so a bucket is reduced in place with ONE collective and nothing is copied.  A
    post-accumulate-grad hook marks the bucket's parameters as backward produces them; the bucket's `all_reduce(async_op=True)`
    is launched the moment the last gradient it EXPECTS has landed — backward runs the stages in reverse, so stage S-1's 13 MB
    travel while stages S-2 .. 0 are still being differentiated.  `finalize()` launches what is left, waits, and divides by the
    world size.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large buckets keep every link busy with long
    messages instead of hundreds of per-tensor rings.  With world_size 1 everything degenerates to plain local gradients.

    Contract (the DDP one):
    * one synchronised backward per step: `zero_grad -> backward -> finalize -> optimizer.step`.  Gradient accumulation: run the
      earlier backward passes under `with reducer.no_sync():` — nothing is launched there, gradients just add up in the buckets.
      A second backward outside `no_sync()` before `finalize()` would add to a buffer that is already being reduced: it raises.
    * ANY `zero_grad` is fine.  `reducer.zero_grad()` zeroes the flat buffers: backward then ADDS every gradient into its view
      (one `add_` launch per parameter).  `reducer.zero_grad(set_to_none=True)` — or torch's default `optimizer.zero_grad()` /
      `module.zero_grad()` — DROPS the `.grad` views instead: autograd then hands each parameter the gradient tensor itself (no
      kernel), the hook notes it, and the moment a bucket is complete its gradients are copied into the flat buffer by ONE
      multi-tensor launch (`torch._foreach_copy_`) and the views are re-attached, so the collective always reduces the real
      gradients.  This is the cheap mode for a step with one backward (~150 launches and dispatcher calls less per step of the
      K-Net head); the gradient tensors may alias a captured graph's static buffers (`enable_chain_graphs`), which is fine because
      they are consumed before the next replay.  Under `no_sync()` a dropped view is filled and re-attached at once.
    * which parameters take part in backward (the video head builds its link modules in every stage but uses the last stage's
      only), and how often each one reports per step, is learned during the first synchronised step (reduced in `finalize()`):
      from the second step on a bucket is launched the moment its expected reports are in, so stage buckets with unused members
      also overlap with backward.  When the set of used parameters changes
      (e.g. switching between `forward_train` and `forward_train_with_previous`), call `reset_usage()`; a gradient arriving in a
      bucket that is already in flight raises instead of being silently lost."""

    def __init__(self, module, bucket_of=None, group=None, force_collectives=False):
        import re
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        # force_collectives: issue the all-reduces even in a ONE-rank group (a 1-GPU box still exercises the RCCL path)
        self._coll = self.world > 1 or (force_collectives and dist.is_available() and dist.is_initialized())
        self._sync = True
        if bucket_of is None:
            def bucket_of(name):
                m = re.search(r'mask_head\.(\d+)\.', name)
                return f'stage{m.group(1)}' if m else 'rest'
        groups = {}
        seen = set()
        self._hook_of = {}
        for name, p in module.named_parameters():
            if not p.requires_grad or id(p) in seen:        # shared parameters (recursive heads) are bucketed once
                continue
            seen.add(id(p))
            groups.setdefault((bucket_of(name), p.device, p.dtype), []).append(p)
        self.buckets = []
        for (key, dev, dt), params in groups.items():
            # every slot starts on a 256-byte boundary (the padding stays zero): a view is then as aligned as a tensor of its own —
            # the kernels' 16-byte accesses, and `FlatSGD`'s parameter views, which the chain kernels read in place
            flat = torch.zeros(sum(self._slot(p.numel()) for p in params), device=dev, dtype=dt)
            b = dict(key=key, flat=flat, params=params, views=[], fired={}, nfired=0, expect=None, expect_total=0, handle=None, grew=False, pending=set())
            off = 0
            for i, p in enumerate(params):
                v = flat[off:off + p.numel()].view_as(p)
                b['views'].append(v)
                p.grad = v
                off += self._slot(p.numel())
                hook = self._make_hook(b, i)
                self._hook_of[id(p)] = hook
                p.register_post_accumulate_grad_hook(hook)
            self.buckets.append(b)
        # heads whose captured chain graphs deliver parameter gradients in bulk (KernelUpdateHead.enable_chain_graphs) report here
        for m in module.modules():
            if hasattr(m, 'on_param_grads'):
                m.on_param_grads = self.params_ready

    @staticmethod
    def _slot(n):
        return (n + 63) // 64 * 64

    def params_ready(self, params):
        """The gradients of `params` have been written to their `.grad` outside autograd's accumulation nodes (a captured backward
        graph): the same bookkeeping as the post-accumulate hook."""
        for p in params:
            hook = self._hook_of.get(id(p))
            if hook is not None:
                hook(p)

    @staticmethod
    def _flush(b):
        """Gradients that arrived as tensors of their own (dropped views): into the flat buffer with one multi-tensor copy."""
        if b['pending']:
            idx = sorted(b['pending'])
            torch._foreach_copy_([b['views'][i] for i in idx], [b['params'][i].grad for i in idx])
            for i in idx:
                b['params'][i].grad = b['views'][i]
            b['pending'] = set()

    def _launch(self, b):
        self._flush(b)
        # members without a gradient in this step (`zero_grad(set_to_none=True)` and no use): their slots of the flat buffer still
        # hold a previous step's averaged gradient — they must contribute ZERO to the sum, as DDP's unused parameters do.
        # Parameter usage is expected to be the same on every rank (the launch points are learned per rank: rank-divergent usage
        # would order the collectives differently); with identical usage this zeroing is what keeps the slot clean everywhere.
        stale = [v for p, v in zip(b['params'], b['views']) if p.grad is None]
        if stale:
            torch._foreach_zero_(stale)
        b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _make_hook(self, b, i):
        def hook(p):
            view = b['views'][i]
            g = p.grad
            if g is not view and (g is None or g.data_ptr() != view.data_ptr()):
                # `zero_grad(set_to_none=True)` or an external `p.grad = ...` dropped the view
                if g is not None and self._sync:
                    b['pending'].add(i)              # stays `p.grad` (a second use of the parameter adds into it) until the bucket is flushed
                else:
                    if g is not None:
                        view.copy_(g)
                    p.grad = view
            if b['handle'] is not None:
                raise RuntimeError(f"BucketedGradAllReducer: a gradient arrived in bucket '{b['key']}' while its all-reduce is in "
                                   'flight (a second backward before finalize() — wrap the earlier ones in no_sync() — or the set '
                                   'of used parameters changed — call reset_usage())')
            if not self._sync:
                return
            # a parameter may report more than once per backward (a captured chain delivers its share in bulk, an eager use of the same
            # parameter reports through autograd later): the bucket is complete when every member has reported as often as it did
            # in the steps seen so far.  The first synchronised step only learns (everything is reduced in finalize()).
            n = b['fired'][i] = b['fired'].get(i, 0) + 1
            b['nfired'] += 1
            exp = b['expect']
            if exp is None:
                return
            if n > exp.get(i, 0):
                b['grew'] = True                                  # more than we expected: reduce this bucket in finalize()
            if self._coll and not b['grew'] and b['nfired'] == b['expect_total']:
                self._launch(b)
        return hook

    class _NoSync:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev, self.owner._sync = self.owner._sync, False

        def __exit__(self, *exc):
            self.owner._sync = self.prev
            return False

    def no_sync(self):
        """Context manager for the non-final backward passes of a gradient-accumulation step (DDP's `no_sync`)."""
        return self._NoSync(self)

    def reset_usage(self):
        """Forget which parameters take part in backward (re-learned during the next synchronised step)."""
        for b in self.buckets:
            b['expect'] = None

    def zero_grad(self, set_to_none=False):
        """Re-arm the hooks and either zero the flat buffers and (re-)attach the parameters' `.grad` views (gradients are added
        into the buckets), or — `set_to_none=True` — drop the views (gradients arrive as tensors, one batched copy per bucket)."""
        for b in self.buckets:
            if set_to_none:
                for p in b['params']:
                    p.grad = None
            else:
                b['flat'].zero_()
                for p, v in zip(b['params'], b['views']):
                    if p.grad is not v:
                        p.grad = v
            b['fired'], b['nfired'], b['handle'], b['grew'], b['pending'] = {}, 0, None, False, set()

    def finalize(self):
        """Launch the collectives of the buckets that are not in flight yet (first step, unused members, accumulation), wait for
        all of them, then average.  Call after the step's last `backward()` and before `optimizer.step()`."""
        for b in self.buckets:
            if b['handle'] is None:
                self._flush(b)
            late = []
            for p, v in zip(b['params'], b['views']):
                # `p.grad is None`: the parameter took no part in this step after a set_to_none zero_grad — it stays None (the
                # optimizer skips it; its slot of the flat buffer is zeroed before the reduction, see _launch)
                if p.grad is not None and p.grad is not v and p.grad.data_ptr() != v.data_ptr():
                    if b['handle'] is not None:   # a gradient assigned behind the hooks' back
                        raise RuntimeError(f"BucketedGradAllReducer: bucket '{b['key']}' was reduced without a member's gradient")
                    late.append((p, v))
            if late:
                torch._foreach_copy_([v for _, v in late], [p.grad for p, _ in late])
                for p, v in late:
                    p.grad = v
        if self._coll:
            for b in self.buckets:
                if b['handle'] is None:
                    self._launch(b)
            for b in self.buckets:
                b['handle'].wait()
                b['flat'].div_(self.world)
        for b in self.buckets:
            if self._sync and b['fired']:
                exp = b['expect'] if b['expect'] is not None else {}
                for i, n in b['fired'].items():
                    exp[i] = max(exp.get(i, 0), n)
                b['expect'], b['expect_total'] = exp, sum(exp.values())
            b['fired'], b['nfired'], b['handle'], b['grew'] = {}, 0, None, False


class FlatSGD:
    """SGD with momentum over the FLAT buckets of a `BucketedGradAllReducer` (torch.optim.SGD's rule, dampening 0, no nesterov): the
    parameters of a bucket are moved into one flat buffer laid out like the bucket's gradient buffer (every `p.data` becomes a view of
    it — values unchanged), and a step is ONE pass per bucket over (parameters, gradients, momentum) on the library's kernel
    (`vkn_sgd_momentum_f32`) instead of torch's multi-tensor walk over ~300 tensors (0.54 ms -> 0.05 ms per step of the cfg3 head).
    Call order: `reducer.zero_grad -> backward -> reducer.finalize -> opt.step`.
    A parameter without a gradient in a step counts as a ZERO gradient (its momentum still decays and is applied) — what
    DistributedDataParallel + SGD do with unused parameters; torch.optim.SGD alone would skip it.  `reducer_divides=True` (default):
    `finalize()` has already divided by the world size."""

    def __init__(self, reducer, lr, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0):
        if nesterov or dampening:
            raise NotImplementedError('FlatSGD: nesterov / dampening are not provided')
        self.reducer, self.lr, self.momentum, self.weight_decay = reducer, float(lr), float(momentum), float(weight_decay)
        self.state = []
        for b in reducer.buckets:
            flat = b['flat']
            if not flat.is_cuda or flat.dtype != torch.float32:
                raise RuntimeError('FlatSGD runs on the GPU kernels: fp32 CUDA parameters only (no CPU fallback)')
            pflat = torch.empty_like(flat)
            pflat.zero_()
            off = 0
            for p in b['params']:
                n = p.numel()
                pflat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = pflat[off:off + n].view_as(p)
                off += reducer._slot(n)
            self.state.append(dict(param=pflat, mom=torch.zeros_like(flat)))

    def zero_grad(self, set_to_none=True):
        self.reducer.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self):
        from . import _lib
        L = _lib.lib()
        for b, st in zip(self.reducer.buckets, self.state):
            # after finalize() every gradient that exists is a view of the flat buffer; a member without one contributes zero
            stale = [v for p, v in zip(b['params'], b['views']) if p.grad is None]
            if stale:
                torch._foreach_zero_(stale)
            with torch.cuda.device(b['flat'].device):
                _lib.check(L.vkn_sgd_momentum_f32(st['param'].data_ptr(), b['flat'].data_ptr(), st['mom'].data_ptr(), b['flat'].numel(),
                                                  self.lr, self.momentum, self.weight_decay, 1.0,
                                                  torch.cuda.current_stream(b['flat'].device).cuda_stream))

