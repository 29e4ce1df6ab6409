#!/usr/bin/env python3
"""Times the output back end (output.rs kernels) on a 1536x1536 depth map resident in HBM and prices each
entry point against its algorithmic bytes: GB/s against the 8 TB/s HBM peak.

    python3 tools/bench_output.py [out.json]
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import matrix_eyes_amd as m


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(ctx, f, iters=20):
    f()
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ctx = m.Context(0, "f16", m.ModelConfig.tiny())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lib, h = ctx.lib, ctx.handle
    S = 1536
    g = torch.Generator().manual_seed(7)
    yy, xx = torch.meshgrid(torch.arange(S), torch.arange(S), indexing="ij")
    depth = (0.5 + 0.4 * torch.sin(xx / 97.0) * torch.cos(yy / 131.0) + 0.05 * torch.rand(S, S, generator=g)).float()
    depth[200:400, 300:900] = 5.0          # a near object: depth discontinuities drop mesh faces
    depth = depth.cuda().contiguous()
    res = []

    def add(name, ms, nbytes, note=""):
        res.append(dict(op=name, ms=round(ms, 4), algorithmic_MB=round(nbytes / 1e6, 2),
                        GBps=round(nbytes / ms / 1e6, 1), frac_of_8TBps=round(nbytes / ms / 1e6 / 8000.0, 4), note=note))
        print(res[-1], flush=True)

    mn, mx = C.c_float(), C.c_float()
    d2 = depth.clone()
    ms = timeit(ctx, lambda: lib.me_depth_clamp_minmax(h, ptr(d2), S * S, C.byref(mn), C.byref(mx)))
    add("depth_clamp_minmax", ms, S * S * 8, "read + write f32; includes the 8-byte D2H of the range")
    mm = torch.empty(2, dtype=torch.float32, device="cuda")
    ms = timeit(ctx, lambda: lib.me_depth_clamp_minmax_async(h, ptr(d2), S * S, ptr(mm)))
    add("depth_clamp_minmax_async", ms, S * S * 8, "read + write f32; the range stays on the device (no host round trip)")
    rgb = torch.empty(S, S, 3, dtype=torch.uint8, device="cuda")
    ms = timeit(ctx, lambda: lib.me_depthmap_rgb(h, ptr(d2), S * S, mn.value, mx.value, ptr(rgb)))
    add("depthmap_rgb", ms, S * S * 7, "f32 in, rgb8 out")
    noise = torch.randint(0, 256, (S, S, 3), dtype=torch.uint8, generator=g).cuda()
    out = torch.empty_like(noise)
    ms = timeit(ctx, lambda: lib.me_stereogram(h, ptr(d2), S, S, mn.value, mx.value, S, S, C.c_float(1.0 / 16.0), ptr(noise), ptr(out)))
    add("stereogram", ms, S * S * (4 + 3 + 3), "depth f32 + noise rgb8 in, rgb8 out")
    ms = timeit(ctx, lambda: (lib.me_depth_clamp_minmax_async(h, ptr(d2), S * S, ptr(mm)),
                              lib.me_stereogram_dev_range(h, ptr(d2), S, S, ptr(mm), S, S, C.c_float(1.0 / 16.0), ptr(noise), ptr(out))))
    add("clamp + stereogram chained on the device", ms, S * S * (8 + 4 + 3 + 3), "DepthMap::new -> output_stereogram without a host round trip")
    vidx = torch.empty(S * S, dtype=torch.int32, device="cuda")
    nv, nf = C.c_int64(), C.c_int64()
    faces = torch.empty((S - 1) * (S - 1) * 2 * 3, dtype=torch.int32, device="cuda")
    ms = timeit(ctx, lambda: lib.me_mesh_index(h, ptr(d2), S, S, ptr(vidx), C.byref(nv), C.byref(nf), ptr(faces)), iters=10)
    add("mesh_index", ms, S * S * 8 + nf.value * 12, f"{nv.value} vertices, {nf.value} faces; depth in, vertex ids + faces out")
    uv = torch.empty(nv.value * 2, dtype=torch.float32, device="cuda")
    xyz = torch.empty(nv.value * 3, dtype=torch.float32, device="cuda")
    ms = timeit(ctx, lambda: lib.me_mesh_vertices(h, ptr(d2), S, S, ptr(vidx), nv.value, S, S, ptr(uv), ptr(xyz)))
    add("mesh_vertices", ms, S * S * 8 + nv.value * 20, "depth + ids in, uv + xyz out")
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r02_output_kernels.json")
    json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
