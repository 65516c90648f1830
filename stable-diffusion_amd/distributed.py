"""Multi-GPU scaling of generate(): independent image-batch sharding, one collective.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for tests).  Rank r owns images [r*B/N, (r+1)*B/N); every image has its own seed, so an image
does not depend on N (shard invariance).  The only communication on the path is one all-gather of
the decoded uint8 images at the end of generate: no tensor/sequence parallelism.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise from torchrun's RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*; returns (rank, world, local device index).

    Rehearsal on a box with fewer GPUs than ranks (never the measured configuration): SDAMD_SHARE_GPU=1 maps rank r
    to device r % device_count and SDAMD_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU); the
    image all-gather then goes through host memory."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SDAMD_SHARE_GPU") == "1" and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()
    backend = backend or os.environ.get("SDAMD_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of ``total`` images owned by ``rank``; sizes differ by at most 1."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_images(images, total=None):
    """images uint8 [b_local, H, W, 3] on every rank -> [total, H, W, 3] in rank order, with exactly ONE collective
    (SURVEY.md §8e): every rank knows every shard's size from ``shard_range(total, r, world)`` -- no size exchange.
    ``total`` = images over all ranks (default: every rank holds ``images.shape[0]``).  Ragged shards (sizes differ
    by at most 1) are padded to the largest inside the same single ``all_gather_into_tensor``."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        if total is not None and images.shape[0] != total:
            raise RuntimeError(f"gathered {images.shape[0]} images, expected {total}")
        return images
    world, rank = dist.get_world_size(), dist.get_rank()
    images = images.contiguous()
    if images.is_cuda and dist.get_backend() == "gloo":      # rehearsal only (see init_from_env): gather through the host
        return all_gather_images(images.cpu(), total).to(images.device)
    if total is None:
        total = world * images.shape[0]
    sizes = [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]
    if images.shape[0] != sizes[rank]:
        raise RuntimeError(f"rank {rank} holds {images.shape[0]} images, shard_range({total}, {rank}, {world}) says {sizes[rank]}")
    mx = max(sizes)
    tail = tuple(images.shape[1:])
    send = images
    if sizes[rank] != mx:
        send = torch.zeros((mx,) + tail, dtype=images.dtype, device=images.device)
        send[:sizes[rank]] = images
    out = torch.empty((world * mx,) + tail, dtype=images.dtype, device=images.device)
    dist.all_gather_into_tensor(out, send)
    if min(sizes) != mx:
        out = torch.cat([out[r * mx:r * mx + n] for r, n in enumerate(sizes)])
    return out


def max_over_ranks(value, device):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
