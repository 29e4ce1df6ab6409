#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of bench.py and prints one step's timeline: per stream the
kernels in start order with duration, grid and the gap to the previous kernel, plus totals by kernel.

    python tools/step_timeline.py gpurun_out/trace/t_kernel_trace.csv [step_index_from_end]
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"_ZN2me(?:\d+_GLOBAL__N_1)?(\d+)", name)
    if m:   # mangled (rocprofv3 without demangling): kernel name + integer template arguments
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        ints = re.findall(r"Li(\d+)E", name)
        ty = "f16" if "DF16_" in name else ("bf16" if "DF16b" in name else "")
        return f"{base}<{ty}{',' if ints and ty else ''}{','.join(ints)}>"
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("me::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("__half", "f16").replace("__hip_bfloat16", "bf16")
    return name[:70]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r.get("Stream_Id", "0")))
    rows.sort()
    # a step starts at each preprocess kernel
    starts = [i for i, r in enumerate(rows) if "preprocess" in r[2]]
    lo = starts[-back - 1] if len(starts) > back else starts[0]
    hi = starts[-back] if back > 0 and len(starts) > back else len(rows)
    step = rows[lo:hi]
    t0 = step[0][0]
    print(f"step of {len(step)} kernels, {(max(r[1] for r in step) - t0) / 1e6:.3f} ms wall")
    last_end = {}
    tot = defaultdict(lambda: [0, 0.0])
    busy = defaultdict(float)
    for s, e, n, g, q in step:
        gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        tot[(n, g)][0] += 1
        tot[(n, g)][1] += (e - s) / 1e3
        busy[q] += (e - s) / 1e3
        if "-v" in sys.argv:
            print(f"q{q:>3} {(s - t0) / 1e3:9.1f} us  +{gap:6.1f}  {(e - s) / 1e3:8.1f} us  g={g:<6} {n}")
    print("busy per queue (us):", dict(busy))
    for (n, g), (c, us) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{us:9.1f} us  x{c:<4} avg {us / c:8.1f}  g={g:<6} {n}")


if __name__ == "__main__":
    main()
