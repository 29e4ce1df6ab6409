"""Builds libmatrixeyes_hip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(jobs: int = 8, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmatrixeyes_hip.so failed")
    return os.path.join(os.path.dirname(CSRC), "libmatrixeyes_hip.so")
