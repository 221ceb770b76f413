"""World-size-2 gloo test of the N > 1 path on CPU: weight broadcast, contiguous image sharding, label-map gather.
The per-rank compute is done by the oracle here (this is a test; the product engine needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from miunet import shard, synth
from miunet.spec import UNetSpec, pack_weights


def test_shard_range_covers_everything_once():
    for n in (0, 1, 7, 16, 513):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_images, out_path):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import oracle_lib as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = UNetSpec(1, 16, 2, 3)
    nbytes = spec.n_params() * 4 + 36
    blob = pack_weights(spec, synth.make_weights(spec, 42)) if rank == 0 else None
    blob = shard.broadcast_blob(blob, nbytes, torch.device("cpu"))
    imgs = synth.make_images(n_images, 16, 16, 1, 0x77)          # every rank can regenerate the global batch
    lo, hi = shard.shard_range(n_images, rank, world)
    if hi > lo:
        _, labels = orc.unet_forward(blob, imgs[lo:hi], nthreads=1)
    else:
        labels = np.zeros((0, 16, 16), np.uint8)
    counts = [b - a for a, b in (shard.shard_range(n_images, r, world) for r in range(world))]
    got = shard.gather_labels(torch.from_numpy(labels), counts)
    if rank == 0:
        _, want = orc.unet_forward(blob, imgs, nthreads=1)
        np.save(out_path, np.array([int(np.array_equal(got.numpy(), want)), got.shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [5, 2])
def test_broadcast_shard_gather_world2(tmp_path, n_images):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), n_images, out), nprocs=2, join=True)
    ok, n = np.load(out)
    assert ok == 1 and n == n_images
