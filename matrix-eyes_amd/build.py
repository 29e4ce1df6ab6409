"""Builds libmatrixeyes_hip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU) and the C++
host layer above it (host/: the `matrix-eyes-hip` CLI and its GPU-free self-test driver)."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HOST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host")


def build(jobs: int = 8, verbose: bool = False) -> str:
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libmatrixeyes_hip.so failed")
    res = subprocess.run(["make", "-C", HOST, f"-j{jobs}"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building the C++ host layer failed")
    return os.path.join(os.path.dirname(CSRC), "libmatrixeyes_hip.so")
