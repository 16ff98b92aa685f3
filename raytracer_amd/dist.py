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


def reduce_accum_async(accum, dst=0, group=None):
    """reduce_accum that returns at once: the collective runs on the backend's own stream behind the work already enqueued on
    the current one; the handle's wait() makes the current stream wait for it (None when there is nothing to reduce)."""
    if dist.is_available() and dist.is_initialized():    # (a one-rank group too: the backend's own no-op, same call sequence)
        return dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


class FramePipeline:
    """Frames through TWO accumulators, so that the reduce of frame i overlaps the render of frame i + 1.

    With one accumulator a rank cannot start its next render before the collective has read the buffer: every frame pays
    render + reduce back to back.  Here frame i renders into buffer i % 2 and its reduce is started without waiting; the
    buffer is only needed again for frame i + 2, and by then its reduce has been waited for -- when frame i + 1 was submitted,
    which is also when `on_frame(accum, i)` runs (rank `dst` holds the sum of all ranks in `accum` at that point; the other
    ranks get their own, spent buffer).  flush() finishes what is still in flight.  Nothing here depends on the backend:
    RCCL on a node (bench.py), gloo in the CPU tests."""

    def __init__(self, height, width, device, on_frame=None, dst=0, group=None):
        self.acc = [new_accum(height, width, device), new_accum(height, width, device)]
        self.work = [None, None]
        self.on_frame, self.dst, self.group = on_frame, dst, group
        self.submitted = 0
        self.finished = 0

    def _finish_upto(self, n):
        while self.finished < n:
            k = self.finished % 2
            if self.work[k] is not None:
                self.work[k].wait()
                self.work[k] = None
            if self.on_frame is not None:
                self.on_frame(self.acc[k], self.finished)
            self.finished += 1

    def submit(self, render):
        """render(accum) enqueues this frame's render into the zeroed accumulator and returns whatever the caller wants back."""
        i = self.submitted
        k = i % 2
        self._finish_upto(i - 1)                      # frame i - 2 used this buffer (a no-op unless submit was called out of step)
        self.acc[k].zero_()
        out = render(self.acc[k])
        self.work[k] = reduce_accum_async(self.acc[k], dst=self.dst, group=self.group)
        self.submitted = i + 1
        self._finish_upto(i)                          # frame i - 1: its reduce ran while this frame was being enqueued / rendered
        return out

    def flush(self):
        self._finish_upto(self.submitted)


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
