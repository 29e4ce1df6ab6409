#!/usr/bin/env python3
"""bench.py — 1536x1536 depth-maps/sec of the HIP Depth Pro path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the script starts its N
ranks itself, as child processes created BEFORE anything touches the GPU, each with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, waits for them and exits with their worst code; under
torch.distributed.run it is one of the ranks.  A step = one pass of the hot path (me_extract_depth_u8: preprocess ->
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
PMC_FILE = os.path.join(ROOT, "profiles", "r05_pmc_kernels.json")
# Box calibration.  ctx.calibrate() runs two FIXED loops of the library (csrc/calibrate.hip, never
# to be edited): an MFMA-only loop and a 512 MiB device copy.  The reference below is the first box measured in round 5
# (profiles/r05_calibration_boxes.json lists every box that produced a profiles/r05_* file); `value_normalised` is what
# this run's rate would read on THAT box under a two-term model of the step -- CAL_MFMA_SHARE of its time scales with
# the MFMA loop's rate (the matrix pipe at the clock the part holds), the rest with the copy's -- so that runs of
# different rounds on different boxes can be put on one scale.  The raw calibration is printed beside it.
CAL_REFERENCE = {"mfma_tflops": 1954.0, "copy_gbs": 4882.0}   # gpurun_out/r05a: the two bench runs of the first round-5 box
CAL_MFMA_SHARE = 0.8


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
    from the committed rocprofv3 PMC passes (profiles/r05_pmc_kernels.json, written by tools/pmc_collect.py:
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
                sum(v["mfma_util"] for v in ops) / n, f"profiles/r05_pmc_kernels.json, source sha {meta['source_sha']}")
    except Exception as e:   # no file: traffic stays null
        return None, None, None, f"no PMC file ({type(e).__name__})"


# Node-level ceiling of the configs[4] chain (profiles/r04_node_write_ceiling.txt, tools/node_write_ceiling.py): 8 processes x 2
# writer threads put 178 files/s of 110 MB onto one tmpfs -- the host kernel's page-cache copy, no GPU involved
NODE_WRITE_CEILING_FILES_PER_S = 178.0


def reduce_over_ranks(elapsed, my_step_ms, world, device):
    """The contract's MAX over ranks of the timed region + every rank's own step time (torch.distributed must be
    initialised; `device` is where the backend wants its tensors: "cuda" under RCCL, "cpu" under gloo)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    per_rank_ms = [None] * world
    dist.all_gather_object(per_rank_ms, my_step_ms)
    return float(t.item()), per_rank_ms


def whole_job_rate(world, batch_per_gpu, steps, elapsed_max):
    """`value`: the images ALL ranks processed in the timed region / the slowest rank's time"""
    return world * batch_per_gpu * steps / elapsed_max


def chain_ceiling_note(world, value):
    """configs[4]: does this run ask the host for more OBJ files per second than one node's page cache takes?"""
    over = value > NODE_WRITE_CEILING_FILES_PER_S
    return {"files_per_s_this_run": round(value, 1), "node_ceiling_files_per_s": NODE_WRITE_CEILING_FILES_PER_S,
            "source": "profiles/r04_node_write_ceiling.txt (8 processes x 2 pwrite threads, 110 MB files, one tmpfs, no GPU)",
            "at_ceiling": bool(value > 0.85 * NODE_WRITE_CEILING_FILES_PER_S),
            "note": (f"{world} rank(s) wrote {value:.0f} files/s; the host's page-cache copy takes about "
                     f"{NODE_WRITE_CEILING_FILES_PER_S:.0f} files/s of this size per node" +
                     (": the chain is bound by the host here, not by the GPUs -- a plateau from this rank count on is the "
                      "file system's" if over or value > 0.85 * NODE_WRITE_CEILING_FILES_PER_S else
                      f" ({NODE_WRITE_CEILING_FILES_PER_S / max(value / world, 1e-9):.1f} ranks at this per-GPU rate would reach it)"))}


def kernel_row(k, steps):
    """One row of the bench line's kernels[]: time, rate on the algorithmic FLOPs, rate on the algorithmic BYTES (every
    operand of the launch once, gemm.hip), both as fractions of the quoted peaks, and which of the two bounds the launch
    sits closer to -- the MFMA fraction alone misstates the memory-bound launches (proj's residual read-modify-write, the
    ConvTransposes' pixel-shuffle stores, the f32-residual convolutions)."""
    sec = k["total_ms"] * 1e-3
    tf = k["flops"] / sec / 1e12 if k["flops"] else None
    gbs = k["bytes"] / sec / 1e9 if k.get("bytes") else None
    peak = MFMA_PEAK_TFLOPS_FP8 if "<fp8," in k["kernel"] else MFMA_PEAK_TFLOPS
    mf = tf / peak if tf else None
    hf = gbs / HBM_PEAK_GBS if gbs else None
    return {"kernel": k["kernel"], "launches_per_step": k["launches"] / steps, "ms_per_step": round(k["total_ms"] / steps, 4),
            "tflops": None if tf is None else round(tf, 1), "mfma_frac": None if mf is None else round(mf, 3),
            "algorithmic_gbs": None if gbs is None else round(gbs, 0), "hbm_frac": None if hf is None else round(hf, 3),
            # nearer which roof: "hbm" where the byte fraction is the larger one and at least 0.3 (a launch at a quarter of
            # both roofs -- attention, whose K and V are re-read from L2, not from HBM -- is short of neither by its bytes)
            "bound": None if (mf is None and hf is None) else ("hbm" if (hf or 0) > (mf or 0) and (hf or 0) >= 0.3 else "mfma")}


def calibration_object(cal, value):
    """bench line's `calibration`: this box's two loop rates, the reference box's, and `value` restated for the
    reference box (rank 0's device; at N > 1 every rank has its own box-to-box factor, rank 0's is quoted)."""
    ref = CAL_REFERENCE
    out = {"mfma_loop_tflops": round(cal["mfma_tflops"], 1), "mfma_loop_clock_ghz": round(cal["mfma_clock_ghz"], 4),
           "copy_gbs": round(cal["copy_gbs"], 1), "mfma_loop_ms": round(cal["mfma_loop_ms"], 3),
           "copy_ms": round(cal["copy_ms"], 4), "cus": cal["cus"],
           "mfma_loop_tflops_passes": cal.get("mfma_runs"), "copy_gbs_passes": cal.get("copy_runs"),
           "reference": dict(ref, box="the first box measured in round 5 (profiles/r05_calibration_boxes.json)"),
           "model": f"step time = {CAL_MFMA_SHARE} x (MFMA-loop-bound) + {round(1 - CAL_MFMA_SHARE, 2)} x (copy-bound)",
           "loops": "csrc/calibrate.hip (fixed: 256 x 512 threads x 40000 x 32 v_mfma_f32_16x16x32_f16, operands in "
                    "registers; 10 x 512 MiB device copy), five passes after the timed region, the medians quoted; "
                    "box-to-box the step follows these loops to +-2 % only (DESIGN.md 5.3): a sanity line against a slow "
                    "box, not a ruler -- rounds are compared on same-box alternating runs"}
    if ref["mfma_tflops"] > 0 and ref["copy_gbs"] > 0 and cal["mfma_tflops"] > 0 and cal["copy_gbs"] > 0:
        speed = CAL_MFMA_SHARE * cal["mfma_tflops"] / ref["mfma_tflops"] + (1 - CAL_MFMA_SHARE) * cal["copy_gbs"] / ref["copy_gbs"]
        out["box_speed_vs_reference"] = round(speed, 4)
        out["value_normalised"] = round(value / speed, 3)
    else:
        out["box_speed_vs_reference"] = None
        out["value_normalised"] = None
    return out


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
    ap.add_argument("--cpu-windows", type=int, default=0,
                    help="cpu_baseline all-cores leg: 0 (default) = the oracle's WHOLE path on one 1536x1536 image, timed "
                         "once (about 75 s on a 128-thread host); n > 0 = a bounded sample of n ViT windows scaled by "
                         "FLOP share (quick runs)")
    ap.add_argument("--chain", action="store_true",
                    help="BASELINE configs[4]: time depth -> DepthMap::new (clamp + range) -> stereogram -> textured OBJ per "
                         "image instead of the depth step alone; reports images/s with the OBJ leg split out")
    return ap.parse_args()


def launch_ranks(args, child_cmd=None, timeout_s=None):
    """`python bench.py --gpus N` without a launcher: N child processes, one per GPU, over RCCL.  This process has
    not imported torch or touched the GPU (a process that has must never be replaced or forked from); the children
    are plain `python bench.py ...` commands with the rank environment torch.distributed.run would give them.  Rank
    0 prints the JSON line on this process's stdout, the other ranks' output goes to stderr; a rank that fails ends
    the others, and so do an interrupt of this process and `timeout_s` (ME_BENCH_RANK_TIMEOUT, default 1800 s: a rank
    hung in a collective would otherwise hold its GPU for ever).  Returns the worst exit code.
    child_cmd: the command of a rank (tests pass a stub); default this script with this process's arguments."""
    import signal
    import socket
    import subprocess
    if timeout_s is None:
        timeout_s = float(os.environ.get("ME_BENCH_RANK_TIMEOUT", "1800"))
    cmd = child_cmd or [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    # the rendezvous port stays bound (SO_REUSEADDR) until every child exists, so that nothing else takes it in between
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    procs = []
    worst = 0

    def on_term(signum, frame):
        raise KeyboardInterrupt

    old_term = signal.signal(signal.SIGTERM, on_term)
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
            procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
        sock.close()
        sock = None
        deadline = time.time() + timeout_s
        pending = set(range(args.gpus))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = worst or rc
                    for o in pending:              # the others would wait for this rank in a collective for ever
                        procs[o].terminate()
            if pending and time.time() > deadline:
                print(f"bench.py: ranks {sorted(pending)} still running after {timeout_s:.0f} s: ending them", file=sys.stderr)
                worst = worst or 124
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        worst = worst or 130
    finally:
        signal.signal(signal.SIGTERM, old_term)
        if sock is not None:
            sock.close()
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        for p in alive:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return worst


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
    if windows <= 0:
        # the whole path, once: preprocess -> encoder (35 + 1 windows) -> decoder -> FOV head -> depth head on one
        # 1536x1536 image -- nothing scaled.  (A few-window sample scaled by FLOP share read 2x too slow: 128 threads
        # barely beat one on 577-row problems, the batched 35-window pass is what the oracle really runs.)
        from matrix_eyes_amd.synthetic import synthetic_images
        img = O.preprocess_u8(synthetic_images(1, cfg.img_size, "structured", seed=4321))
        torch.set_num_threads(all_threads)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.extract_depth(img, None, weights, ocfg)
            dt = time.perf_counter() - t0
        multi = {"value": 1.0 / dt, "cores": all_threads,
                 "sample": (f"the whole path on ONE 1536x1536 image (FOV head on, {TFLOP_PER_IMAGE_FOV} TFLOP), timed once: "
                            f"{dt:.1f} s on {all_threads} thread(s); nothing scaled")}
    else:
        multi, dt = sample(windows, all_threads)
        if dt < 6.0:      # aim at about 10 s of CPU work per sample
            multi, dt = sample(int(min(35, max(windows + 1, round(windows * 10.0 / dt)))), all_threads)
    single, _ = sample(3, 1)          # three windows = 1146 GFLOP: ~10 s at one core's sgemm rate (scaled by share)
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
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist
    import matrix_eyes_amd as m
    from matrix_eyes_amd.synthetic import synthetic_checkpoint, synthetic_images

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` plainly (it starts "
                         f"its ranks itself) or under torch.distributed.run with --nproc-per-node N")
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

    chain_ms = {"depth": 0.0, "raster": 0.0, "obj": 0.0, "obj_bytes": 0, "obj_device": 0.0, "obj_d2h": 0.0, "obj_file": 0.0}
    chain_pipelined = False
    if args.chain:
        # BASELINE configs[4] per image: depth -> DepthMap::new (clamp + range, output.rs:44-75) -> stereogram
        # (output.rs:141-193) -> textured OBJ + MTL (output.rs:195-261) written to a file on tmpfs
        import shutil
        import tempfile
        out_dir = tempfile.mkdtemp(prefix=f"me_chain_r{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        noise = torch.from_numpy(np.random.default_rng(99).integers(0, 256, size=(S, S, 3), dtype=np.uint8)).cuda()
        stereo = torch.empty(B, S, S, 3, dtype=torch.uint8, device="cuda")
        # me_ctx_set_write_behind: image i's file is written by a host thread while the GPU works on image i + 1; every
        # file is complete (me_output_flush) before the timed region ends.  ME_CHAIN_SYNC_WRITES=1: the reference's form.
        write_behind = os.environ.get("ME_CHAIN_SYNC_WRITES") is None
        ctx.set_write_behind(max(2, B) if write_behind else 0)
        # me_ctx_set_output_overlap: step k + 1's depth is queued BEFORE step k's output calls, which run
        # on the context's output stream behind the step that wrote their buffer -- the GPU works on the next depth maps
        # while the host waits for this step's mesh counts, OBJ text and its D2H copy.  Two depth buffers in turn.
        # ME_CHAIN_SERIAL=1: round 4's form (depth, synchronise, raster, synchronise, OBJ).
        chain_pipelined = os.environ.get("ME_CHAIN_SERIAL") is None
        ctx.set_output_overlap(chain_pipelined)
        depth_bufs = [depth, torch.empty_like(depth)]
    chain_step = [0]

    def chain_outputs(k, timed_legs):
        """DepthMap::new -> stereogram -> textured OBJ for the B images of step k (depth buffer k & 1)"""
        buf = depth_bufs[k & 1]
        t1 = time.perf_counter()
        maps = [m.DeviceDepthMap(ctx, buf[b], (S, S)) for b in range(B)]
        for b in range(B):
            maps[b].stereogram(1.0 / 16.0, noise, out=stereo[b])
        if timed_legs:
            ctx.synchronize()
        t2 = time.perf_counter()
        chain_step[0] += 1
        for b in range(B):
            # two names per image slot: a file still being written behind the caller is never the next call's target
            path = os.path.join(out_dir, f"mesh{b}_{chain_step[0] & 1}.obj")
            maps[b].output_mesh(path, "photo.jpg", m.VertexMode.Texture)
            if timed_legs:
                legs = ctx.last_mesh_timing()
                chain_ms["obj_bytes"] = legs["bytes"]
                chain_ms["obj_device"] += legs["mesh_ms"] + legs["format_ms"]
                chain_ms["obj_d2h"] += legs["d2h_ms"]
                chain_ms["obj_file"] += legs["file_ms"]
        t3 = time.perf_counter()
        if timed_legs:
            chain_ms["raster"] += (t2 - t1) * 1e3
            chain_ms["obj"] += (t3 - t2) * 1e3

    def step():
        ctx.extract_depth(rgb, f_norm, out=depth)

    def chain_serial(n, timed_legs):
        """every leg behind the one before it, synchronised: the per-leg diagnostic (and ME_CHAIN_SERIAL=1's timed form)"""
        for k in range(n):
            t0 = time.perf_counter()
            ctx.extract_depth(rgb, f_norm, out=depth_bufs[k & 1])
            ctx.synchronize()
            if timed_legs:
                chain_ms["depth"] += (time.perf_counter() - t0) * 1e3
            chain_outputs(k, timed_legs)

    chain_host = {"enqueue_depth": 0.0, "output_calls": 0.0, "steps": 0}

    def chain_pipeline(n):
        """step k + 1's depth queued before step k's output calls; nothing synchronised between legs"""
        ctx.extract_depth(rgb, f_norm, out=depth_bufs[0])
        for k in range(n):
            h0 = time.perf_counter()
            if k + 1 < n:
                ctx.extract_depth(rgb, f_norm, out=depth_bufs[(k + 1) & 1])
            h1 = time.perf_counter()
            chain_outputs(k, False)
            chain_host["enqueue_depth"] += (h1 - h0) * 1e3
            chain_host["output_calls"] += (time.perf_counter() - h1) * 1e3
            chain_host["steps"] += 1

    def run_steps(n):
        if not args.chain:
            for _ in range(n):
                step()
        elif chain_pipelined:
            chain_pipeline(n)
        else:
            chain_serial(n, False)

    run_steps(args.warmup)
    chain_host.update(enqueue_depth=0.0, output_calls=0.0, steps=0)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    if args.chain:
        ctx.output_flush()
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    if args.chain:
        ctx.output_flush()          # every OBJ file of the timed steps is on the file system
    fence()
    elapsed = time.perf_counter() - t0
    my_step_ms = elapsed / args.steps * 1e3
    if args.chain:
        # per-leg diagnostic: a separate pass, synchronised between legs (never the timed region of the pipelined form)
        n_diag = max(2, min(5, args.steps))
        chain_serial(n_diag, True)
        ctx.output_flush()
        chain_report = {k: (v / (n_diag * B) if k != "obj_bytes" else v) for k, v in chain_ms.items()}
        shutil.rmtree(out_dir, ignore_errors=True)
        ctx.set_output_overlap(False)
        args_chain, args.chain = True, False      # the roofline / end-to-end legs below time the depth step itself
    else:
        chain_report = {}
        args_chain = False
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
    # calibration leg: outside the timed region, on a chip as warm as the steps left it.  FIVE passes of the two loops (about
    # 0.25 s): one 25 ms pass reads +-3 % on the same box depending on the clock state it starts in (tools/calib_probe.py:
    # 1885 / 1997 / 1911 TFLOP/s within one process), the median of five is what the normalisation uses
    cal_runs = [ctx.calibrate() for _ in range(5)]
    cal = dict(sorted(cal_runs, key=lambda c: c["mfma_tflops"])[2])
    cal["copy_gbs"] = sorted(c["copy_gbs"] for c in cal_runs)[2]
    cal["mfma_runs"] = [round(c["mfma_tflops"], 1) for c in cal_runs]
    cal["copy_runs"] = [round(c["copy_gbs"], 1) for c in cal_runs]
    per_rank_ms = [my_step_ms]
    if distributed:
        elapsed, per_rank_ms = reduce_over_ranks(elapsed, my_step_ms, world, "cuda")
    assert bool(torch.isfinite(depth).all()), "non-finite depth"
    # device-result calls leave the flag alone (matrix_eyes_hip.h): this covers every warm-up and timed step
    assert ctx.status_flags() == 0, "an f16 operand overflowed during the benchmark (me_status_flags)"

    if rank == 0:
        value = whole_job_rate(world, B, args.steps, elapsed)
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
                              "1xMI355X, depth map + FOV head" +
                              ("" if args.dtype != "f16" else " (f16 operands: the fp16 checkpoint bit for bit, 7.1e-4 relative L2 "
                               "from the fp32 CPU path; bf16 operands measure 1.06e-2, outside north_star's 1e-3)")) if (B == 1 and not args.no_fov) else
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
                "per_rank_ms_per_step": [round(v, 3) for v in per_rank_ms],
                "weights_broadcast": ("none (one rank)" if world == 1 else
                                      ("me_bcast_weights: the library's own RCCL communicator" if os.environ.get("ME_NATIVE_RCCL") == "1"
                                       else f"torch.distributed broadcast of the arena tensor (backend {backend})")),
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
            "kernels": [kernel_row(k, args.steps) for k in kernels[:10]],
        }
        out["calibration"] = calibration_object(cal, value)
        out["value_normalised"] = out["calibration"]["value_normalised"]
        if args_chain:
            # configs[4]: `value` is whole-chain images/s; the legs are per image, host wall clock around synchronised
            # sections (depth = the model step, raster = DepthMap::new + stereogram kernels, obj = mesh index +
            # vertex kernels + text formatting + file write on tmpfs)
            out["config"]["workload"] = (f"BASELINE.json configs[4] per GPU: {B} image(s) per step, depth -> DepthMap::new -> "
                                         "stereogram -> textured OBJ + MTL on tmpfs")
            out["chain"] = {"depth_ms_per_image": round(chain_report["depth"], 3),
                            "raster_ms_per_image": round(chain_report["raster"], 3),
                            "obj_ms_per_image": round(chain_report["obj"], 3),
                            "obj_legs_ms_per_image": {
                                "mesh_index_and_text_kernels": round(chain_report["obj_device"], 3),
                                "d2h_of_the_text": round(chain_report["obj_d2h"], 3),
                                "file_write_tmpfs": round(chain_report["obj_file"], 3)},
                            "obj_bytes": chain_report["obj_bytes"],
                            "write_behind": bool(write_behind),
                            "pipelined": bool(chain_pipelined),
                            "host_ms_per_step": ({k: round(v / max(1, chain_host["steps"]), 3) for k, v in chain_host.items() if k != "steps"}
                                                 if chain_pipelined else None),
                            "form": ("step k + 1's me_extract_depth is queued before step k's output calls, which run on the "
                                     "context's output stream behind the step that wrote their depth buffer "
                                     "(me_ctx_set_output_overlap); nothing is synchronised between legs inside the timed region; "
                                     "the *_ms_per_image legs are a separate, synchronised diagnostic pass"
                                     if chain_pipelined else "serial: every leg synchronised (ME_CHAIN_SERIAL=1)"),
                            "node_write_ceiling": chain_ceiling_note(world, value),
                            "note": "the file write is the host kernel's page-cache copy (about 2.8 GB/s on tmpfs whatever the "
                                    "thread count); everything before it runs on the GPU.  With write_behind "
                                    "(me_ctx_set_write_behind) a host thread writes image i's file while the GPU works on "
                                    "image i + 1; file_write_tmpfs is then what the CALLER waited (hand-over, or an earlier "
                                    "file still being written), and every file is flushed inside the timed region"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, args.cpu_windows)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
