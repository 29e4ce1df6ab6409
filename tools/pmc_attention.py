#!/usr/bin/env python3
"""Issue-side PMC counters of the attention kernel (development tool; same mechanics as pmc_collect.py).
    python3 tools/pmc_attention.py [op cfg]      default: attn 0"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_collect import ROOT, run_pass

PASSES = [["SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"],
          ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_WAIT_INST_ANY"],
          ["SQ_INSTS_VALU_TRANS_F32", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVES"]]


def main():
    op = sys.argv[1] if len(sys.argv) > 1 else "attn"
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    work = os.path.join(ROOT, "gpurun_out", "pmc_issue")
    os.makedirs(work, exist_ok=True)
    entry = {}
    for counters in PASSES:
        vals, dur = run_pass(op, cfg, counters, work)
        entry.update(vals)
        entry.setdefault("durations_us", []).append(dur)
        print(counters, vals, dur, flush=True)
    json.dump(entry, open(os.path.join(ROOT, "gpurun_out", f"pmc_issue_{op}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
