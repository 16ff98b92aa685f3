"""Multi-GPU plumbing: one process per GPU, samples of every pixel split across ranks, one RCCL reduce at the end.

The reference has no distributed code (one OpenMP loop, 03_FullTracer/main.c:137).  The path shards naturally by sample
index: rank r renders samples [begin_r, end_r) of EVERY pixel into its own float4 accumulator with disjoint Philox
counters, so no data-path collective is needed until the single sum of the accumulators (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist


def shard_samples(total_spp, rank, world_size, base=0):
    """Contiguous, disjoint, exhaustive split of [base, base+total_spp) -- earlier ranks take the remainder."""
    if world_size < 1 or not (0 <= rank < world_size) or total_spp < 0:
        raise ValueError("bad shard request")
    q, r = divmod(total_spp, world_size)
    begin = base + rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def reduce_accum(accum, dst=0, group=None):
    """Sum the (H, W, 4) accumulators of all ranks onto `dst` (radiance sums AND sample counts add up)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return accum


def new_accum(height, width, device):
    return torch.zeros((height, width, 4), dtype=torch.float32, device=device)


def render_into(scene, cam, settings, accum):
    """Enqueue rtc_render_device on torch's current stream of accum's device; returns immediately."""
    if not accum.is_cuda or accum.dtype != torch.float32 or not accum.is_contiguous():
        raise ValueError("accum must be a contiguous float32 CUDA/HIP tensor of shape (H, W, 4)")
    s = settings.c
    if tuple(accum.shape) != (s.height, s.width, 4):
        raise ValueError(f"accum shape {tuple(accum.shape)} does not match settings {(s.height, s.width, 4)}")
    stream = torch.cuda.current_stream(accum.device).cuda_stream
    scene.render_device(cam, settings, accum.data_ptr(), stream)
    return stream
