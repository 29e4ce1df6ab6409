#!/usr/bin/env python3
"""PMC counters of the kernels INSIDE the step (roofline.traffic comes from stand-alone probes of the step's
shapes, tools/pmc_collect.py): rocprofv3 --pmc passes over `bench.py --steps 3 --warmup 1 --no-cpu-baseline` itself, one counter
group per pass, aggregated per kernel name.

    python3 tools/pmc_in_step.py out.json

Started on the GPU box from a process that has not touched the GPU (this one only spawns children); the interpreter itself sits
after `--` (no launcher hop).  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as in pmc_collect.py."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]]


def short(name):
    name = re.sub(r"\(me::GemmParams\)|void |me::|\(anonymous namespace\)::", "", name)
    return name[:96]


def main():
    out_json = sys.argv[1]
    work = os.path.join(ROOT, "gpurun_out", "pmc_step")
    os.makedirs(work, exist_ok=True)
    agg = {}
    for counters in PASSES:
        out = os.path.join(work, "_".join(counters)[:60])
        cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=600, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"rocprofv3 pass {counters} failed with code {r.returncode}:\n{(r.stdout or '')[-2000:]}")
        durs = {}
        for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                durs.setdefault(short(row["Kernel_Name"]), []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                e = agg.setdefault(short(row["Kernel_Name"]), {})
                e.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in durs.items():
            agg.setdefault(k, {}).setdefault("duration_us", []).extend(v)
        print("pass", counters, "done", flush=True)
    result = {}
    for k, e in agg.items():
        if "calib_" in k or "rocclr" in k:
            continue
        r = {"dispatches": len(e.get("FETCH_SIZE", e.get("duration_us", [])))}
        for c, v in e.items():
            r[c] = sum(v) / len(v)
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["hbm_bytes_per_launch"] = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
        if r.get("GRBM_GUI_ACTIVE"):
            r["mfma_util"] = r.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (r["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        result[k] = r
    sys.path.insert(0, ROOT)
    import bench
    result["_meta"] = {"source_sha": bench.kernel_source_sha(), "command": "bench.py --steps 3 --warmup 1 --no-cpu-baseline under rocprofv3 --pmc",
                       "note": "per-launch averages over every dispatch of the kernel in the profiled run (warm-up, timed and end-to-end steps)"}
    json.dump(result, open(out_json, "w"), indent=1)
    rows = sorted(((k, v) for k, v in result.items() if k != "_meta" and "hbm_bytes_per_launch" in v),
                  key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["dispatches"])
    for k, v in rows[:14]:
        print(f"{k[:70]:70s} x{v['dispatches']:4d}  {v['hbm_bytes_per_launch'] / 1e6:8.1f} MB  {v.get('duration_us', 0):7.1f} us  mfma {v.get('mfma_util', 0):.3f}")


if __name__ == "__main__":
    main()
