"""Multi-GPU plumbing: one process per GPU, image-parallel, ONE RCCL broadcast of the
packed weight arena at start-up, no collective in the hot path.  `torch.distributed` carries only the
128-byte RCCL id and the timing reduction; backend "nccl" is RCCL on ROCm, "gloo" is used by the CPU
tests."""
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None, device_index: Optional[int] = None):
    """Joins the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (no-op when
    WORLD_SIZE is 1)."""
    rank, local_rank, world = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
    kwargs = {}
    if backend == "nccl":
        kwargs["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_images(n_images: int, rank: int, world: int) -> List[int]:
    """image i -> GPU i mod G: indices of the images this rank processes"""
    return list(range(rank, n_images, world))


def broadcast_bytes(payload: Optional[bytes], src: int = 0) -> bytes:
    """ships a small byte string (the RCCL unique id) from `src` to every rank"""
    if not dist.is_initialized():
        return payload
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized():
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def distribute_weights(ctx, state_dict_fn, rank: int, world: int, native: Optional[bool] = None):
    """rank 0 loads (state_dict_fn() -> name -> tensor) and finalizes; then ONE broadcast of the packed
    arena (1.9 GB at f16) reaches the other ranks over xGMI.

    Default: the broadcast runs on torch.distributed's existing RCCL communicator (backend "nccl"), on
    the arena wrapped zero-copy as a uint8 tensor — one RCCL instance and one bootstrap per process.
    native=True (or ME_NATIVE_RCCL=1) uses the library's own communicator instead (me_bcast_weights:
    the path a non-Python caller takes)."""
    if native is None:
        native = os.environ.get("ME_NATIVE_RCCL", "0") == "1"
    if rank == 0:
        ctx.load_state_dict(state_dict_fn())
    if world <= 1:
        if native:   # one rank: the communicator is still created, the in-place broadcast runs (every RCCL call of the path)
            ctx.bcast_weights(ctx.rccl_unique_id(), 0, 1)
        return
    # every rank must lay its arena out as rank 0 does (dtype, geometry, split_operands): compared before any payload
    # moves, on every rank, so that nobody hangs in the collective (me_bcast_weights does the same check natively)
    layouts = [None] * world
    dist.all_gather_object(layouts, (ctx.weight_arena_layout(), ctx.weight_arena_bytes()))
    if len(set(layouts)) != 1:
        from ._lib import MatrixEyesError
        raise MatrixEyesError(1, f"weight arena layouts differ between ranks: {layouts}")
    if native:
        uid = broadcast_bytes(ctx.rccl_unique_id() if rank == 0 else None, 0)
        ctx.bcast_weights(uid, rank, world)
        return
    arena = ctx.weight_arena_tensor()
    dist.broadcast(arena, src=0)
    torch.cuda.synchronize()
    if rank != 0:
        ctx.adopt_weights()
