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
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, last, rank + 1, group))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, recv, rank - 1, group))
    for req in (dist.batch_isend_irecv(ops) if ops else []):
        req.wait()
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


class BucketedGradAllReducer:
    """Data-parallel gradient averaging for the head (BASELINE cfg3: frames of a clip sharded over the GPUs of a node, RCCL
    all-reduce over xGMI; the reference gets this from mmcv's `MMDistributedDataParallel`, external/train.py:53-61).

    Parameters are grouped into buckets (default: one per `mask_head.{s}` stage + one for the rest).  Every parameter's `.grad` is
    a VIEW into its bucket's flat fp32 buffer, so a bucket is reduced in place with ONE collective and nothing is copied.  A
    post-accumulate-grad hook counts the bucket's parameters as backward produces them; the bucket's `all_reduce(async_op=True)`
    is launched the moment its last gradient lands — backward runs the stages in reverse, so stage S-1's 13 MB travel while stages
    S-2 .. 0 are still being differentiated.  `finalize()` waits for the collectives and divides by the world size.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large buckets keep every link busy with long messages instead of
    hundreds of per-tensor rings.  With world_size 1 (or no process group) everything degenerates to plain local gradients."""

    def __init__(self, module, bucket_of=None, group=None):
        import re
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if bucket_of is None:
            def bucket_of(name):
                m = re.search(r'mask_head\.(\d+)\.', name)
                return f'stage{m.group(1)}' if m else 'rest'
        groups = {}
        seen = set()
        for name, p in module.named_parameters():
            if not p.requires_grad or id(p) in seen:        # shared parameters (recursive heads) are bucketed once
                continue
            seen.add(id(p))
            groups.setdefault((bucket_of(name), p.device, p.dtype), []).append(p)
        self.buckets = []
        self._handles = []
        for (key, dev, dt), params in groups.items():
            flat = torch.zeros(sum(p.numel() for p in params), device=dev, dtype=dt)
            b = dict(key=key, flat=flat, params=params, ready=0, handle=None)
            off = 0
            for p in params:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._make_hook(b))
            self.buckets.append(b)

    def _make_hook(self, b):
        def hook(_p):
            b['ready'] += 1
            if b['ready'] == len(b['params']) and self.world > 1:
                b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return hook

    def zero_grad(self):
        """Zero the flat buffers (the parameters' `.grad` views stay attached) and re-arm the hooks."""
        for b in self.buckets:
            b['flat'].zero_()
            b['ready'], b['handle'] = 0, None

    def finalize(self):
        """Wait for the in-flight collectives (buckets whose parameters did not all take part in this backward are reduced
        now), then average.  Call after `loss.backward()` and before `optimizer.step()`."""
        if self.world > 1:
            for b in self.buckets:
                if b['handle'] is None:
                    b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for b in self.buckets:
                b['handle'].wait()
                b['flat'].div_(self.world)
        for b in self.buckets:
            b['ready'] = 0
            b['handle'] = None
