"""-m gpu: the N > 1 start-up and sharding path with two processes on the one GPU of the test box.
RCCL refuses two ranks on one device, so torch.distributed runs on gloo here (it moves CUDA tensors
through the host); everything else is the production path: rank 0 packs the weights, the arena is
broadcast as one uint8 tensor, rank 1 adopts it, images are dealt i mod N and every rank's depth must
equal the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_IMAGES = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import torch
    import torch.distributed as dist
    import matrix_eyes_amd as m
    from matrix_eyes_amd import distributed as D
    from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images
    torch.cuda.set_device(0)
    D.init("gloo")
    cfg = m.ModelConfig.tiny()
    ctx = m.Context(0, "f16", cfg)
    loads = []

    def make():
        loads.append(rank)
        return synthetic_checkpoint(cfg)

    D.distribute_weights(ctx, make, rank, world, native=False)
    rgb = synthetic_images(N_IMAGES, cfg.img_size, "structured", seed=99)
    mine = D.shard_images(N_IMAGES, rank, world)
    depth, fov = ctx.extract_depth(rgb[mine], None, want_fov=True)
    slowest = D.max_over_ranks(float(rank))
    D.barrier()
    q.put((rank, loads, mine, depth, fov, slowest))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_share_one_packed_arena():
    import matrix_eyes_amd as m
    from matrix_eyes_amd.synthetic import synthetic_images
    from util import loaded_ctx
    world, port = 2, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0] and res[1][1] == []          # only rank 0 touched the checkpoint
    assert res[0][2] == [0, 2] and res[1][2] == [1, 3]
    assert all(r[5] == 1.0 for r in res)
    ctx = loaded_ctx("tiny", "f16")
    rgb = synthetic_images(N_IMAGES, ctx.cfg.img_size, "structured", seed=99)
    want, want_fov = ctx.extract_depth(rgb, None, want_fov=True)
    for _, _, mine, depth, fov, _ in res:
        assert np.array_equal(depth, want[mine]) and np.array_equal(fov, want_fov[mine])
