"""CPU, gloo, world_size 2: the sample-shard + single-reduce pattern of the multi-GPU path."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_samples_partitions_exactly():
    from raytracer_amd.dist import shard_samples
    for total in (0, 1, 7, 256, 16384, 1000003):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_samples(total, r, world, base=5) for r in range(world)]
            assert spans[0][0] == 5 and spans[-1][1] == 5 + total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_samples(10, 2, 2)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from raytracer_amd.dist import reduce_accum, shard_samples
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, spp = 6, 5, 9
    b, e = shard_samples(spp, rank, world)
    # stand-in for the renderer: sample s of pixel p contributes a known value, so the reduced sum is checkable
    s = torch.arange(b, e, dtype=torch.float32).view(1, 1, -1)
    p = torch.arange(H * W, dtype=torch.float32).view(H, W, 1)
    acc = torch.zeros(H, W, 4)
    acc[..., 0] = (p * 0.5 + s).sum(-1)
    acc[..., 1] = (s * s).sum(-1).expand(H, W)
    acc[..., 3] = float(e - b)
    reduce_accum(acc, dst=0)
    dist.barrier()
    if rank == 0:
        torch.save(acc, out)
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_rank(tmp_path):
    out = str(tmp_path / "acc.pt")
    mp.spawn(_worker, args=(2, 29731, out), nprocs=2, join=True)
    acc = torch.load(out)
    H, W, spp = 6, 5, 9
    s = np.arange(spp)
    p = np.arange(H * W).reshape(H, W)
    assert np.allclose(acc[..., 0].numpy(), (p[..., None] * 0.5 + s).sum(-1))
    assert np.allclose(acc[..., 1].numpy(), (s * s).sum())
    assert (acc[..., 3] == spp).all()


def _pipeline_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from raytracer_amd.dist import FramePipeline, shard_samples
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, spp, frames = 4, 3, 10, 5
    got = []
    pipe = FramePipeline(H, W, "cpu", on_frame=lambda acc, idx: got.append((idx, acc.clone())))
    for f in range(frames):
        b, e = shard_samples(spp * world, rank, world, base=f * spp * world)

        def render(acc, b=b, e=e):                     # stand-in renderer: sample s adds s to channel 0 and 1 to the count
            assert float(acc.abs().sum()) == 0.0       # the buffer comes zeroed, two frames after its last use
            acc[..., 0] += float(sum(range(b, e)))
            acc[..., 3] += float(e - b)
            return f
        assert pipe.submit(render) == f
        assert len(got) == f                           # frame f - 1 has been delivered, frame f is in flight
        if f == 2:
            pipe.flush()                               # a flush in the middle (bench.py: the fence between warm-up and timed steps)
            assert len(got) == 3
    pipe.flush(); pipe.flush()
    dist.barrier()
    if rank == 0:
        torch.save([(i, a) for i, a in got], out)
    dist.destroy_process_group()


def test_frame_pipeline_delivers_every_frame_in_order(tmp_path):
    """raytracer_amd.dist.FramePipeline (two accumulators, the reduce of a frame overlapping the next frame's render), gloo, world 2:
    every frame arrives once, in order, with the sum over both ranks' sample shards."""
    out = str(tmp_path / "frames.pt")
    mp.spawn(_pipeline_worker, args=(2, 29741, out), nprocs=2, join=True)
    got = torch.load(out)
    spp, world = 10, 2
    assert [i for i, _ in got] == [0, 1, 2, 3, 4]
    for f, acc in got:
        lo = f * spp * world
        assert float(acc[0, 0, 0]) == float(sum(range(lo, lo + spp * world)))
        assert (acc[..., 3] == spp * world).all()
