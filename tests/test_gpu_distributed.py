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


@pytest.mark.timeout(900)
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the script starts its two ranks itself (child processes created
    before anything touches the GPU), rank 0 builds the checkpoint, the packed arena reaches rank 1 by ONE broadcast,
    both ranks time the same steps and rank 0 prints one JSON line with n_gpus 2, the aggregate rate and both ranks'
    step times.  On this one-GPU box the ranks share the device, so torch.distributed runs on gloo (ME_DIST_BACKEND;
    RCCL refuses two ranks on one device) -- the driver's multi-GPU runs take the same entry with the default, nccl."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ME_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=800, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    per_rank = d["config"]["per_rank_ms_per_step"]
    assert len(per_rank) == 2 and all(v > 0 for v in per_rank)
    assert abs(d["ms_per_step"] - max(per_rank)) / max(per_rank) < 0.25        # MAX over ranks (barrier skew aside)
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 0.02     # two images per step, whole job
    assert "gloo" in d["config"]["weights_broadcast"]
