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


def previous_kernels_for_block(block_kernels: torch.Tensor, first_previous=None, group=None, all_nonempty=False):
    """`block_kernels` [T_r, N, C] = final kernels of this rank's frames.  Returns prev [T_r, N, C] with
    prev[i] = kernels of the frame BEFORE frame i of the block: the previous rank's last frame for i = 0
    (rank 0: `first_previous` [1,N,C], or its own frame 0 when the clip starts the video), own frame i-1 otherwise.
    One all_gather of [N, C] per rank; ranks with an empty block contribute zeros and are skipped by their successor.
    `all_nonempty=True` (every rank is known to own frames, e.g. an even split): the predecessor is simply rank - 1 and no
    device -> host read of the has-frames flags is needed (keeps the step free of host synchronisation)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    T = block_kernels.shape[0]
    N, C = block_kernels.shape[1:]
    if world > 1:
        last = block_kernels[-1].contiguous() if T > 0 else block_kernels.new_zeros(N, C)
        has = torch.tensor([1.0 if T > 0 else 0.0], device=block_kernels.device)
        payload = torch.cat([last.reshape(-1), has])             # one message per rank: [N*C kernels | has-frames flag]
        got = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(got, payload, group=group)
        p0 = None
        if all_nonempty:
            if rank > 0:
                p0 = got[rank - 1][:-1].reshape(1, N, C)
        else:
            for r in range(rank - 1, -1, -1):        # nearest predecessor that owns at least one frame
                if float(got[r][-1]) > 0:
                    p0 = got[r][:-1].reshape(1, N, C)
                    break
    else:
        p0 = None
    if p0 is None:
        p0 = first_previous if first_previous is not None else block_kernels[:1]
    if T == 0:
        return block_kernels
    return torch.cat([p0.reshape(1, N, C), block_kernels[:-1]], dim=0)
