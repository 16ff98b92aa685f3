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
