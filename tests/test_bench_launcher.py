"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run) on the CPU, with stub
children: a rank that fails ends the others and its code is returned, nobody is left behind, a hung rank is timed out,
every child gets the rank environment torch.distributed.run would give it."""
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the library at module level)

STUB = r"""
import os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
open(os.path.join(sys.argv[1], f"pid{rank}"), "w").write(str(os.getpid()))
mode = sys.argv[2]
if mode == "ok":
    print(f"rank {rank} of {world}")
    sys.exit(0)
if mode == "rank1_fails":
    if rank == 1:
        time.sleep(0.3)
        sys.exit(7)
    time.sleep(60)          # "waiting in a collective" for the rank that died
if mode == "hang":
    time.sleep(60)
"""


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie of ours would have been reaped by launch_ranks' wait(); anything still signalable is alive
    try:
        return open(f"/proc/{pid}/stat").read().split()[2] != "Z"
    except OSError:
        return False


def _run(tmp_path, mode, n=3, timeout_s=None):
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    args = types.SimpleNamespace(gpus=n)
    t0 = time.time()
    rc = bench.launch_ranks(args, child_cmd=[sys.executable, str(stub), str(tmp_path), mode], timeout_s=timeout_s)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(n)]
    return rc, pids, time.time() - t0


def test_all_ranks_succeed(tmp_path):
    rc, pids, _ = _run(tmp_path, "ok")
    assert rc == 0 and not any(_alive(p) for p in pids)


def test_a_failing_rank_ends_the_others(tmp_path):
    rc, pids, took = _run(tmp_path, "rank1_fails")
    assert rc == 7                                   # the failing rank's code, not the terminated ranks' -15
    assert took < 20 and not any(_alive(p) for p in pids)


def test_hung_ranks_are_timed_out(tmp_path):
    rc, pids, took = _run(tmp_path, "hang", n=2, timeout_s=1.0)
    assert rc == 124 and took < 20 and not any(_alive(p) for p in pids)


def test_interrupting_the_launcher_ends_the_ranks(tmp_path):
    """SIGTERM to `python bench.py --gpus 2` itself (a driver's timeout): the children go with it."""
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    code = (f"import sys, types; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks(types.SimpleNamespace(gpus=2), child_cmd=[sys.executable, {str(stub)!r}, {str(tmp_path)!r}, 'hang']))")
    p = subprocess.Popen([sys.executable, "-c", code])
    deadline = time.time() + 20
    while time.time() < deadline and not all((tmp_path / f"pid{r}").exists() for r in range(2)):
        time.sleep(0.05)
    time.sleep(0.2)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(2)]
    p.terminate()
    assert p.wait(timeout=20) == 130
    time.sleep(0.2)
    assert not any(_alive(q) for q in pids)


RANK_STUB = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import bench
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
steps, B = 20, 8
my_step_ms = 170.0 + 1.5 * rank                  # every rank its own pace; rank 7 is the slowest
elapsed, per_rank = bench.reduce_over_ranks(my_step_ms * steps / 1e3, my_step_ms, world, "cpu")
if rank == 0:
    value = bench.whole_job_rate(world, B, steps, elapsed)
    print(json.dumps({"n_gpus": world, "value": value, "per_rank_ms_per_step": per_rank, "elapsed": elapsed,
                      "chain": bench.chain_ceiling_note(world, value)}), flush=True)
dist.barrier()
dist.destroy_process_group()
"""


def test_eight_ranks_aggregate_like_the_contract_says(tmp_path):
    """what `python bench.py --gpus 8` reduces over its ranks, rehearsed on the CPU (gloo) through
    bench.py's own launcher and its own reduction code: eight per-rank step times, `value` = the images of ALL ranks
    / the SLOWEST rank's time, and the configs[4] note that says when the node's file system is the bound."""
    import json
    stub = tmp_path / "rank.py"
    stub.write_text(RANK_STUB)
    out = tmp_path / "out.json"
    code = (f"import sys, types; sys.path.insert(0, {ROOT!r}); import bench; "
            f"sys.exit(bench.launch_ranks(types.SimpleNamespace(gpus=8), child_cmd=[sys.executable, {str(stub)!r}, {ROOT!r}], timeout_s=300))")
    r = subprocess.run([sys.executable, "-c", code], stdout=open(out, "w"), stderr=subprocess.PIPE, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(out.read_text().strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and len(line["per_rank_ms_per_step"]) == 8
    assert line["per_rank_ms_per_step"] == [170.0 + 1.5 * r for r in range(8)]
    slowest_s = (170.0 + 1.5 * 7) * 20 / 1e3
    assert abs(line["elapsed"] - slowest_s) < 1e-9 and abs(line["value"] - 8 * 8 * 20 / slowest_s) < 1e-6
    # 8 ranks x 8 images / 0.1805 s = 355 files/s: past the node's 178 -> the line explains its own plateau
    assert line["chain"]["at_ceiling"] and "bound by the host" in line["chain"]["note"]
    assert not bench.chain_ceiling_note(1, 36.7)["at_ceiling"] and "ranks at this per-GPU rate" in bench.chain_ceiling_note(1, 36.7)["note"]


def test_calibration_object_restates_the_rate_for_the_reference_box():
    """bench.py's `calibration`: the medians of the five passes, the reference box's rates, and `value_normalised` = value /
    (0.8 x MFMA-loop ratio + 0.2 x copy ratio): a box exactly like the reference leaves the value alone, a box whose MFMA loop
    is 5 % slower raises it by 1 / 0.96, and a dead calibration (zeros) gives None instead of a division by zero."""
    ref = bench.CAL_REFERENCE
    cal = {"mfma_tflops": ref["mfma_tflops"], "mfma_clock_ghz": 1.9, "copy_gbs": ref["copy_gbs"], "mfma_loop_ms": 22.0,
           "copy_ms": 0.22, "cus": 256, "mfma_runs": [1.0, 2.0, 3.0, 4.0, 5.0], "copy_runs": [5.0, 4.0, 3.0, 2.0, 1.0]}
    out = bench.calibration_object(cal, 44.0)
    assert out["value_normalised"] == 44.0 and out["box_speed_vs_reference"] == 1.0
    assert out["mfma_loop_tflops_passes"] == [1.0, 2.0, 3.0, 4.0, 5.0] and out["copy_gbs_passes"] == [5.0, 4.0, 3.0, 2.0, 1.0]
    assert out["reference"]["mfma_tflops"] == ref["mfma_tflops"] and "calibrate.hip" in out["loops"]
    slow = dict(cal, mfma_tflops=0.95 * ref["mfma_tflops"])
    out = bench.calibration_object(slow, 44.0)
    assert abs(out["box_speed_vs_reference"] - (bench.CAL_MFMA_SHARE * 0.95 + (1 - bench.CAL_MFMA_SHARE))) < 1e-4
    assert abs(out["value_normalised"] - 44.0 / out["box_speed_vs_reference"]) < 1e-3
    dead = dict(cal, mfma_tflops=0.0)
    out = bench.calibration_object(dead, 44.0)
    assert out["value_normalised"] is None and out["box_speed_vs_reference"] is None
