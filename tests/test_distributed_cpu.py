"""CPU test of the N > 1 plumbing with the gloo backend, world_size 2: image sharding, the byte
broadcast that carries the RCCL id, and the max-over-ranks timing reduction (bench.py's contract)."""
import os
import socket

import pytest
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import matrix_eyes_amd  # noqa: F401
    from matrix_eyes_amd import distributed as D
    r, lr, w = D.init("gloo")
    uid = D.broadcast_bytes(bytes(range(128)) if r == 0 else None, 0)
    slowest = D.max_over_ranks(1.0 + r)
    mine = D.shard_images(64, r, w)
    D.barrier()
    q.put((r, w, uid == bytes(range(128)), slowest, mine))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_plumbing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1] and all(r[1] == 2 and r[2] for r in res)
    assert all(r[3] == 2.0 for r in res)                       # MAX over ranks
    assert res[0][4] == list(range(0, 64, 2)) and res[1][4] == list(range(1, 64, 2))
    assert sorted(res[0][4] + res[1][4]) == list(range(64))    # every image exactly once


def test_single_process_is_a_no_op():
    from matrix_eyes_amd import distributed as D
    assert D.broadcast_bytes(b"x") == b"x" and D.max_over_ranks(3.5) == 3.5
    assert D.shard_images(5, 0, 1) == [0, 1, 2, 3, 4]
    D.barrier()
