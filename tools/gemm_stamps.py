"""diagnostic: per-workgroup phase times of the persistent GEMM (needs a -DME_GEMM_STAMPS build:
make -C matrix-eyes_amd/csrc clean && make CXXFLAGS+=-DME_GEMM_STAMPS ...)"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matrix_eyes_amd as m
from tools.bench_kernels import ptr
ctx = m.Context(0, "f16", m.ModelConfig.tiny())
lib, h = ctx.lib, ctx.handle
lib.me_debug_set_stamps.argtypes = [C.c_void_p]
M = int(os.environ.get("STAMPS_M", 35 * 577))
PP = int(os.environ.get("STAMPS_PP_CFG", 0))   # 0 = the two-group 256x256 kernel, 5 = its 192x256 form
for (N, K, name, cfg) in [(3072, 1024, "qkv", PP), (1024, 4096, "fc2", 3), (4096, 1024, "fc1", PP), (1024, 1024, "proj-resid", PP), (1024, 4096, "fc2-resid", PP)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).half()
    bias = torch.randn(N, device="cuda"); out16 = torch.empty(M, N, dtype=torch.float16, device="cuda")
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    for it in range(3):
        stamps.zero_()
        lib.me_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
        if name.endswith("resid"):
            if it == 0:
                x32 = torch.randn(M, N, device="cuda"); gamma = torch.rand(N, device="cuda")
            rc = lib.me_op_linear_residual(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(gamma), ptr(x32), cfg)
        else:
            rc = lib.me_op_linear(h, M, N, K, ptr(a), ptr(w), ptr(bias), ptr(out16), None, 1 if name == "fc1" else 0, cfg)
        if rc: print("rc", rc, lib.me_last_error(h))
        ctx.synchronize()
    lib.me_debug_set_stamps(None)
    raw = stamps.cpu().numpy()
    s = raw.reshape(-1, 16).astype(np.float64)
    nwg = int(np.sum(s[:256, 0] > 0)) if cfg in (0, 5, 7, 10) else int(np.sum(s[:512, 0] > 0))
    nw = 8 if cfg in (0, 5, 7, 10) else 4
    npz = 8 if cfg in (0, 5, 7, 10) else 4
    ph = raw[nwg * 16: nwg * 16 + nwg * nw * npz].reshape(nwg, nw, npz).astype(np.float64)
    names = ["dma top", "k-substep 0", "dma mid", "k-substep 1", "vmcnt", "barrier"] if cfg in (0, 5, 7, 10) else ["dma issue", "reads+mfma", "vmcnt(0)", "barrier"]
    s = s[:nwg]
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0   # 100 MHz
    us[s == 0] = np.nan
    print(name, "cfg", cfg, "workgroups", len(s), "kernel span us", np.nanmax(us))
    # stamps: 0 start, then per tile (main_end, epi_end)
    for t in range(5):
        me, ee = 1 + 2 * t, 2 + 2 * t
        if np.all(np.isnan(us[:, me])): break
        prev = us[:, 0] if t == 0 else us[:, ee - 2]
        print(f"  tile {t}: n={np.sum(~np.isnan(us[:, me]))} main {np.nanmean(us[:, me] - prev):.2f} us (min {np.nanmin(us[:, me] - prev):.2f} max {np.nanmax(us[:, me] - prev):.2f})"
              f"  epilogue {np.nanmean(us[:, ee] - us[:, me]):.2f} us (min {np.nanmin(us[:, ee] - us[:, me]):.2f} max {np.nanmax(us[:, ee] - us[:, me]):.2f})  start spread {np.nanstd(prev):.2f}")
    tot = ph.sum(axis=2)
    print("  phase cycles per wave (mean over workgroups), waves 0..%d:" % (nw - 1))
    for i, nm in enumerate(names):
        print(f"    {nm:11s}", " ".join(f"{ph[:, w, i].mean():9.0f}" for w in range(nw)), f"  share {ph[:, :, i].sum() / tot.sum():.3f}")
    print(f"    total      ", " ".join(f"{tot[:, w].mean():9.0f}" for w in range(nw)))
