#!/usr/bin/env python3
"""bench.py — 1536x1536 depth-maps/sec of the HIP Depth Pro path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  A step = one pass of the hot path (me_extract_depth_u8: preprocess ->
encoder -> decoder -> FOV head -> depth head) over one batch of synthetic u8 images that already sit
in HBM.  At N = 1 the workload is BASELINE.json configs[1] (a single 1536x1536 image, depth map +
FOV head).  With N > 1 every rank runs the same per-GPU batch on its own images (weak scaling, image
parallel, no collective in the timed loop); rank 0 builds the synthetic checkpoint and ships the
packed weight arena to the others with ONE RCCL broadcast before the timed region.

Rank 0 prints ONE JSON line.  `roofline` is for the kernel with the largest share of the step
(measured live with HIP events on the launch stream); `cpu_baseline` is the CPU oracle timed on this
box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per 1536x1536 image (SURVEY.md §8d / App. B, 2*M*N*K convention)
TFLOP_PER_IMAGE_FOV = 19.247
TFLOP_PER_IMAGE_NOFOV = 18.865
VIT_WINDOW_GFLOP = 382.13
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16/f16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_PEAK_TFLOPS_FP8 = 5000.0  # dense MX-scaled fp8, MI355X_MICROARCH.md "Peak FP8 MFMA"
HBM_PEAK_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_kernels.json")


def kernel_source_sha():
    """sha256 over the kernel sources: a PMC file is only quoted for the build it was collected on"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "matrix-eyes_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "matrix-eyes_amd", "csrc", "*.hip"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_name):
    """(HBM-side bytes per launch, algorithmic bytes per launch, MFMA utilisation, note) of the dominant kernel
    from the committed rocprofv3 PMC passes (profiles/r02_pmc_kernels.json, written by tools/pmc_collect.py:
    separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of tools/gemm_probe.py on the shapes this kernel
    alternates between in the step).  Both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
    of a wide coalesced stream (MI355X_MICROARCH.md, HBM), so it is doubled; Infinity-Cache hits are included in
    it.  The file records the sha of the kernel sources it was collected on: a file from another build is
    refused (traffic null), as is a kernel without a pass."""
    try:
        pmc = json.load(open(PMC_FILE))
        meta = pmc.get("_meta", {})
        if meta.get("source_sha") != kernel_source_sha():
            return None, None, None, f"PMC file is from another build (source sha {meta.get('source_sha')})"
        ops = [v for k, v in pmc.items() if k != "_meta" and v.get("bench_kernel") == kernel_name]
        if not ops:
            return None, None, None, "no PMC pass for this kernel"
        n = float(len(ops))
        return (sum(v["hbm_bytes"] for v in ops) / n, sum(v["algorithmic_bytes"] for v in ops) / n,
                sum(v["mfma_util"] for v in ops) / n, f"profiles/r02_pmc_kernels.json, source sha {meta['source_sha']}")
    except Exception as e:   # no file: traffic stays null
        return None, None, None, f"no PMC file ({type(e).__name__})"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "fp8"],
                    help="MFMA operand type: f16 (default; BASELINE configs[1], the fp16 checkpoint bit for bit), bf16, "
                         "or fp8 (BASELINE configs[3]: the ViT linears on MX block-scaled fp8, the rest f16)")
    ap.add_argument("--no-fov", action="store_true", help="pass f_norm = 1 instead of the FOV head")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured hipGraph (me_ctx_set_graph); default: eager launches, "
                         "measured equal (f16) to 1 %% faster (fp8): the GPU is never starved by the host")
    ap.add_argument("--cpu-windows", type=int, default=1,
                    help="ViT windows of the oracle sample timed for cpu_baseline")
    return ap.parse_args()


def cpu_baseline(cfg, weights, windows):
    """The oracle's patch-encoder ViT-L over some of the image's 35 windows on the host cores, scaled to images/s
    by its share of the image's algorithmic FLOPs.  Two samples: all host cores -- the figure in
    `value` -- and ONE thread, which is what the reference's lockfile implies for its Burn-ndarray build
    (matrixmultiply without the threading feature, Cargo.lock:643,3452); about 10 s of CPU work each."""
    import torch
    from oracle import depth_pro_oracle as O
    ocfg = O.OracleConfig(grid=cfg.grid, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads,
                          depth=cfg.depth, tap_blocks=tuple(cfg.tap_blocks), enc_dims=tuple(cfg.enc_dims),
                          dec_dim=cfg.dec_dim, head_dims=tuple(cfg.head_dims))
    g = torch.Generator().manual_seed(1234)

    def run(n):
        xs = torch.rand(n, 3, cfg.window, cfg.window, generator=g) * 2 - 1
        with torch.no_grad():
            t0 = time.perf_counter()
            O.vit_forward_features(xs, weights, "encoder.patch_encoder.", ocfg, list(cfg.tap_blocks))
            return time.perf_counter() - t0

    def sample(n, threads):
        torch.set_num_threads(threads)
        dt = run(n)
        share = n * VIT_WINDOW_GFLOP / 1e3 / TFLOP_PER_IMAGE_FOV
        return {"value": share / dt, "cores": threads,
                "sample": (f"patch-encoder ViT-L over {n} of 35 windows = {n * VIT_WINDOW_GFLOP:.0f} GFLOP = "
                           f"{share * 100:.2f}% of one image's {TFLOP_PER_IMAGE_FOV} TFLOP, {dt:.1f} s on {threads} "
                           f"thread(s); scaled by that share")}, dt

    all_threads = torch.get_num_threads()
    multi, dt = sample(windows, all_threads)
    if dt < 6.0:      # aim at about 10 s of CPU work per sample
        multi, dt = sample(int(min(35, max(windows + 1, round(windows * 10.0 / dt)))), all_threads)
    single, _ = sample(3, 1)          # three windows = 1146 GFLOP: ~10 s at one core's sgemm rate
    torch.set_num_threads(all_threads)
    return {
        "value": multi["value"],
        "unit": "depth-maps/s",
        "cores": multi["cores"],
        "kind": "port",
        "sample": ("CPU oracle (PyTorch fp32 restatement of the reference; Burn-ndarray cannot be built here): " +
                   multi["sample"]),
        "single_thread": {"value": single["value"], "unit": "depth-maps/s", "cores": 1, "sample": single["sample"]},
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import matrix_eyes_amd as m
    from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes "
                         f"(WORLD_SIZE={world})")
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # ME_DIST_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (RCCL refuses
    # two ranks on one device); the driver's runs use the default, nccl = RCCL, one GPU per rank
    backend = os.environ.get("ME_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    cfg = m.ModelConfig()
    ctx = m.Context(local_rank, args.dtype, cfg)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_graph(args.graph)

    # ---- weights: rank 0 parses/packs, one RCCL broadcast of the arena
    weights = {}
    t_load = time.perf_counter()

    def make_weights():
        weights.update(synthetic_checkpoint(cfg))   # seed 2024; the real depth_pro.pt is not available offline
        return weights

    from matrix_eyes_amd import distributed as D
    D.distribute_weights(ctx, make_weights, rank, world)
    t_load = time.perf_counter() - t_load

    # ---- inputs resident in HBM before the timed region
    S, B = cfg.img_size, args.batch
    rgb = torch.from_numpy(synthetic_images(B, S, "structured", seed=4321 + rank)).cuda()
    depth = torch.empty(B, S, S, dtype=torch.float32, device="cuda")
    f_norm = torch.ones(B, device="cuda") if args.no_fov else None   # on the device: no host pointer in the call

    def step():
        ctx.extract_depth(rgb, f_norm, out=depth)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    graph_replays = ctx.graph_launch_count
    # roofline leg: the same steps once more with every GEMM / attention / LayerNorm launch bracketed
    # by HIP events on the launch stream (an event costs ~5 us of queue time, so it is kept out of the
    # timed region above)
    ctx.profile_enable(True)
    for _ in range(args.steps):
        step()
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    # end to end as the boundary hands buffers over when the caller keeps them on the host: u8 image in pageable
    # host memory in, f32 depth in host memory out, H2D + D2H and their synchronisation included (never `value`)
    e2e_ms = None
    if rank == 0:
        rgb_host = rgb.cpu().numpy()
        ctx.extract_depth(rgb_host, f_norm)
        t1 = time.perf_counter()
        n_e2e = max(3, min(10, args.steps))
        for _ in range(n_e2e):
            ctx.extract_depth(rgb_host, f_norm)
        e2e_ms = (time.perf_counter() - t1) / n_e2e * 1e3
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(depth).all()), "non-finite depth"

    if rank == 0:
        images = world * B * args.steps
        value = images / elapsed
        tflop_img = TFLOP_PER_IMAGE_NOFOV if args.no_fov else TFLOP_PER_IMAGE_FOV
        # dominant kernel: largest total time over the timed region
        dom = max(prof, key=lambda k: k["total_ms"])
        prof_ms = sum(k["total_ms"] for k in prof)
        dom_ms = dom["total_ms"] / dom["launches"]
        traffic, alg_bytes, mfma_util, pmc_note = pmc_traffic(dom["kernel"])
        achieved = dom["flops"] / dom["launches"] / (dom_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS_FP8 if "<fp8," in dom["kernel"] else MFMA_PEAK_TFLOPS
        step_ms = elapsed / args.steps * 1e3
        kernels = sorted(prof, key=lambda k: -k["total_ms"])
        out = {
            "metric": "1536x1536 depth-maps/sec",
            "value": round(value, 3),
            "unit": "depth-maps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3),
            "end_to_end_ms_per_step": None if e2e_ms is None else round(e2e_ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": (("BASELINE.json configs[3] at one image: single 1536x1536 image, ViT linears on MX fp8 MFMA, "
                              "the rest f16, depth map + FOV head" if args.dtype == "fp8" else
                              "BASELINE.json configs[1]: single 1536x1536 image, 16-bit MFMA HIP path on "
                              "1xMI355X, depth map + FOV head") if (B == 1 and not args.no_fov) else
                             f"{B} x 1536x1536 images per GPU per step, " +
                             ("f_norm given" if args.no_fov else "FOV head")),
                "batch_per_gpu": B,
                "images": "u8 [B,1536,1536,3] synthetic 'structured' (seed 4321 + rank), resident in HBM",
                "checkpoint": "synthetic, seed 2024, exact key set, 951.99 M parameters rounded to fp16",
                "f_norm": "1.0" if args.no_fov else "FOV head (mod.rs:343-358)",
                "parallelism": f"image-parallel, {world} process(es), one per GPU; packed weights by one RCCL "
                               f"broadcast at start-up ({t_load:.1f} s incl. synthetic init)",
                "end_to_end": "end_to_end_ms_per_step: u8 image in pageable host memory in, f32 depth in host memory out "
                              "(H2D 7.1 MB + D2H 9.4 MB per image and their synchronisation included), rank 0",
                "split_operands": cfg.split_operands,
                "launch": (f"one hipGraphLaunch per step ({graph_replays} replays up to the end of the timed region)"
                           if graph_replays else "eager: one launch per kernel"),
                "tflop_per_image": tflop_img,
                "model_tflops": round(value * tflop_img, 1),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": dom["kernel"],
                "launches_per_step": dom["launches"] / args.steps,
                "avg_launch_ms": round(dom_ms, 5),
                "achieved": round(achieved, 1),
                "peak": peak,
                "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "traffic_source": pmc_note,
                "algorithmic_bytes": alg_bytes,
                "mfma_util_pmc": None if mfma_util is None else round(mfma_util, 3),
                "share_of_profiled_kernel_time": round(dom["total_ms"] / prof_ms, 3),
                "whole_step_frac": round(value / world * tflop_img / MFMA_PEAK_TFLOPS, 4),
            },
            "kernels": [{"kernel": k["kernel"], "launches_per_step": k["launches"] / args.steps,
                         "ms_per_step": round(k["total_ms"] / args.steps, 4),
                         "tflops": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12, 1) if k["flops"] else None}
                        for k in kernels[:8]],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, args.cpu_windows)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
