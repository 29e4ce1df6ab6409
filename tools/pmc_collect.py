#!/usr/bin/env python3
"""Collects rocprofv3 PMC counters for the dominant kernels, one counter group per pass (TCC can hold
FETCH_SIZE or WRITE_SIZE, not both; SQ groups kept small), on tools/gemm_probe.py runs of one shape each.

    python3 tools/pmc_collect.py out.json [op:cfg ...]

Must be started on the GPU box from a process that has not touched the GPU (this one only spawns
children).  Output: per op the per-dispatch average of every counter, the kernel's average duration
under the profiler and the derived figures bench.py / DESIGN.md quote:
  hbm_bytes   = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950: FETCH_SIZE reads half of a wide stream)
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
                (GRBM_GUI_ACTIVE comes back summed over the 8 XCDs; MFMA_BUSY is 16 cycles per 16x16x32 MFMA)
  clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# TCC_HIT / TCC_MISS (round 3): where the operand over-fetch of FETCH_SIZE is served from -- L2 hit rate =
# TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2); a pass whose counters this rocprofv3 does not
# know is skipped
PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
          ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCC_REQ_sum", "TCC_EA0_RDREQ_sum"]]
# the launches of one step at one image: qkv / fc1 / proj / fc2 on the 352-row tile (config 10: three / four / one exact
# rounds, pipeline.hip tall_tile_wins); the 3x3 convolutions on the halo tile (config 9).
# The round-3 short-tail launches (ME_GEMM_TALL=0) stay selectable: fc1_tail:7 proj_tail:7 fc2_tail:7 with fc1:0 ...
DEFAULT = ["qkv_tall:10", "fc1_tall:10", "proj_tall:10", "fc2_tall:10", "conv768:9", "attn:0"]
CFG_NAMES = {0: "256x256x64/8w-pp", 1: "128x128x64/4w", 2: "64x64x64/4w", 3: "160x128x64/4w", 4: "64x64x64/4w-ring6",
             5: "192x256x64/8w-pp", 6: "256x256x64/8w-8ph", 7: "96x256x64/8w-pp", 8: "128x256x64/8w-ring3",
             9: "16x16px-x256x64/8w-halo", 10: "352x256x64/8w-pp"}
# op -> (gemm_probe.py op, N, K, rows): the rows each launch of the step works on
SHAPES = {"qkv": ("qkv", 3072, 1024, 21760), "fc1": ("fc1", 4096, 1024, 20480), "fc1_tail": ("fc1", 4096, 1024, 1280),
          "proj": ("proj", 1024, 1024, 16384), "proj_tail": ("proj", 1024, 1024, 5376),
          "fc2": ("fc2", 1024, 4096, 16384), "fc2_tail": ("fc2", 1024, 4096, 5376),
          "qkv_tall": ("qkv", 3072, 1024, 21760), "fc1_tall": ("fc1", 4096, 1024, 21760), "proj_tall": ("proj", 1024, 1024, 21760),
          "fc2_tall": ("fc2", 1024, 4096, 21760)}
SHAPES8 = {"qkv8": (3072, 1024), "fc1_8": (4096, 1024), "fc2_8": (1024, 4096)}   # MX fp8 operands
PROBE_M = 21760


def describe(op, cfg):
    """(kernel name as bench.py's profile reports it, algorithmic bytes per launch)"""
    if op in SHAPES:
        probe, n, k, rows = SHAPES[op]
        resid = probe in ("proj", "fc2")
        out = rows * n * (8 if resid else 2)       # f32 read-modify-write, or one 16-bit store
        if resid and cfg == 10:
            out += rows * n * 2 + n * 8            # + the normalised 16-bit rows and the LayerNorm weights (gemm_probe.py runs the fused form)
        bytes_ = rows * k * 2 + n * k * 2 + out + n * 4 * (2 if resid else 1)
        return f"gemm_kernel<f16,{CFG_NAMES[cfg]},plain,{'resid_scale' if resid else 'store'}>", bytes_
    if op in SHAPES8:
        n, k = SHAPES8[op]
        resid = op == "fc2_8"
        out = PROBE_M * n * (8 if resid else (2 if op == "qkv8" else 1 + 1 / 32))
        bytes_ = PROBE_M * k * (1 + 1 / 32) + n * k * (1 + 1 / 32) + out + n * 4 * (2 if resid else 1)
        # (fc2 at one image runs on the 352-row tile, gemm_fp8.hip fp8_tall_wins; the probe launches the plain residual form)
        return f"gemm_kernel<fp8,{'352' if resid else '256'}x256x128/8w-pp,plain,{'resid_scale' if resid else 'store'}>", int(bytes_)
    if op == "conv768":   # 16-bit bordered input, weights, f32 residual in, f32 + 16-bit out
        px = 768 * 768
        return f"gemm_kernel<f16,{CFG_NAMES[cfg]},conv,store>", 770 * 770 * 256 * 2 + 256 * 2304 * 2 + px * 256 * (4 + 4 + 2)
    return "attention2_kernel", 35 * 577 * (3072 + 1024) * 2
KERNEL_KEYS = ("gemm_kernel", "gemm_pp_kernel", "gemm_pp8_kernel", "gemm_pp8t_kernel", "gemm_ring_kernel", "gemm_8ph_kernel", "conv_halo_kernel",
               "attention_kernel", "attention2_kernel", "attention3_kernel")


def run_pass(op, cfg, counters, work):
    out = os.path.join(work, f"{op}_{'_'.join(counters)}"[:80])
    probe, rows = (SHAPES[op][0], SHAPES[op][3]) if op in SHAPES else (op, PROBE_M)
    # the interpreter itself after `--`, never a launcher script: the profiler's preloaded library has initialised
    # the GPU by then and this pool forbids an exec hop from such a process.  sys.executable as it is: a symlink is not
    # an exec hop, and resolving it would lose a venv's pyvenv.cfg and with it torch / numpy
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p",
           "--", sys.executable, os.path.join(ROOT, "tools", "gemm_probe.py"), probe, str(cfg), "6"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", PROBE_M=str(rows)),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    if r.returncode != 0:
        low = (r.stdout or "").lower()
        if any(c.lower() in low for c in counters) and ("not found" in low or "unknown" in low or "invalid" in low
                                                        or "unsupported" in low or "not supported" in low):
            print(f"pmc_collect: rocprofv3 does not know {counters}: pass skipped", flush=True)
            return {}, None
        raise RuntimeError(f"rocprofv3 pass {counters} on {op}:{cfg} failed with code {r.returncode}:\n{(r.stdout or '')[-2000:]}")
    vals, durs = {}, []
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNEL_KEYS):
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in KERNEL_KEYS):
                durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # rocprofv3 reports one row per (dispatch, counter); drop the first (cold) dispatch
    res = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in vals.items()}
    dur = sum(durs[1:]) / max(1, len(durs) - 1) if durs else None
    return res, dur


def main():
    out_json = sys.argv[1]
    ops = sys.argv[2:] or DEFAULT
    work = os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(work, exist_ok=True)
    result = {}
    for spec in ops:
        op, cfg = spec.split(":")
        entry, durs = {}, []
        for counters in PASSES:
            vals, dur = run_pass(op, int(cfg), counters, work)
            entry.update(vals)
            if dur:
                durs.append(dur)
            print(spec, counters, vals, dur, flush=True)
        entry["duration_us_under_pmc"] = sum(durs) / max(1, len(durs))
        if "FETCH_SIZE" in entry and "WRITE_SIZE" in entry:
            entry["hbm_bytes"] = (2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0
        if entry.get("TCC_HIT_sum") is not None and entry.get("TCC_MISS_sum") is not None:
            entry["l2_hit_rate"] = entry["TCC_HIT_sum"] / max(1.0, entry["TCC_HIT_sum"] + entry["TCC_MISS_sum"])
        if entry.get("GRBM_GUI_ACTIVE"):
            gui = entry["GRBM_GUI_ACTIVE"] / 8.0
            entry["mfma_util"] = entry.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0)
            entry["clock_ghz"] = gui / entry["duration_us_under_pmc"] / 1e3
        entry["tile_config"] = int(cfg)
        entry["bench_kernel"], entry["algorithmic_bytes"] = describe(op, int(cfg))
        result[op] = entry
    sys.path.insert(0, ROOT)
    import bench
    result["_meta"] = {"source_sha": bench.kernel_source_sha(),
                       "note": "sha256[:16] over matrix-eyes_amd/csrc/*.h + *.hip of the build these passes ran on; "
                               "bench.py quotes roofline.traffic from this file only when its own build has the same sha"}
    json.dump(result, open(out_json, "w"), indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
