"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding + the one all-gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import sd_amd
from sd_amd import distributed
from conftest import ROOT


def test_shard_ranges_cover_and_are_contiguous():
    for total in (1, 7, 8, 64, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [distributed.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        distributed.shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import sd_amd  # noqa: F401
    from sd_amd import distributed as d
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    d.init_from_env(backend="gloo")
    lo, hi = d.shard_range(total, rank, world)
    # "decode" of image i = a constant image with value i: per-image work independent of the shard
    imgs = torch.stack([torch.full((4, 4, 3), i, dtype=torch.uint8) for i in range(lo, hi)])
    # count every collective torch.distributed issues during the gather: SURVEY.md §8e says exactly one
    import torch.distributed as dist
    calls = []
    for name in ("all_gather", "all_gather_into_tensor", "all_reduce", "broadcast", "all_to_all", "gather", "all_gather_object"):
        orig = getattr(dist, name)
        setattr(dist, name, (lambda o, n: lambda *a, **k: (calls.append(n), o(*a, **k))[1])(orig, name))
    out = d.all_gather_images(imgs, total)
    assert calls == ["all_gather_into_tensor"], calls
    worst = d.max_over_ranks(float(rank + 1), torch.device("cpu"))
    d.barrier()
    q.put((rank, out.numpy(), worst))


def test_config4_shard_sizes():
    """BASELINE config 4: global batch 64 strong-scaled over 1/2/4/8 GPUs = 64/N images per rank."""
    for world in (1, 2, 4, 8):
        sizes = [hi - lo for lo, hi in (distributed.shard_range(64, r, world) for r in range(world))]
        assert sizes == [64 // world] * world


@pytest.mark.parametrize("total", [8, 5, 64])
def test_all_gather_returns_rank_ordered_images(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.full((4, 4, 3), i, dtype=np.uint8) for i in range(total)])
    for rank, out, worst in results:
        assert np.array_equal(out, want), rank
        assert worst == 2.0
