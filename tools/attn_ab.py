"""attention kernel variants at the step's shape (37 windows x 577 tokens x 16 heads), random data, interleaved rounds
in ONE process (cdna_hip_programming.md rule 24): the round-1 kernel (ME_ATT_V=1) against attention2_kernel with its
deferred and exact running maximum.  Prints median / min us and TFLOP/s on the real
FLOPs (4 * windows * heads * tokens^2 * 64)."""
import ctypes as C, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import matrix_eyes_amd as m

ctx = m.Context(0, "f16", m.ModelConfig.tiny())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 37
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
M = W * 577
torch.manual_seed(1)
qkv = (torch.randn(M, 3072, device="cuda") * 1.2).half()
qkv_pre = qkv.clone()
qkv_pre[:, :1024] = (qkv[:, :1024].float() * (0.125 * 1.4426950408889634)).half()
out = torch.empty(M, 1024, dtype=torch.float16, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())

VARIANTS = [
    ("v1 (round 1 kernel)", {}, False),
    ("v2 4 waves x 32 q (round 4)", {}, True),
    ("v2p next tile's S ahead, 3 waves/SIMD", {"ME_ATT_V": "4"}, True),
    ("v2p next tile's S ahead, 2 waves/SIMD", {"ME_ATT_V": "4", "ME_ATT_MINW": "2"}, True),
    ("v3 4 waves x 48 q, persistent", {"ME_ATT_V": "3"}, True),
]
KEYS = ("ME_ATT_V", "ME_ATT_THR", "ME_ATT_HALVES", "ME_ATT_MINW")


def run(env, pre):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    fn = ctx.lib.me_op_attention_prescaled if pre else ctx.lib.me_op_attention
    rc = fn(ctx.handle, p(qkv_pre if pre else qkv), p(out), W, 577, 16)
    assert rc == 0, rc


outs = {}
for name, env, pre in VARIANTS:
    for _ in range(3):
        run(env, pre)
    torch.cuda.synchronize()
    outs[name] = out.float().clone()
ref = outs[VARIANTS[0][0]]
print("v2p == v2 bit for bit:", bool(torch.equal(outs[VARIANTS[1][0]], outs[VARIANTS[2][0]])), bool(torch.equal(outs[VARIANTS[1][0]], outs[VARIANTS[3][0]])))
for name, o in outs.items():
    print(f"{name:42s} finite {bool(torch.isfinite(o).all())}  rel-L2 vs v1 {float((o - ref).norm() / ref.norm()):.2e}", flush=True)

times = {name: [] for name, _, _ in VARIANTS}
for r in range(ROUNDS):
    for name, env, pre in VARIANTS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run(env, pre)
        e1.record()
        torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) * 1e3 / 4)
flop = 4.0 * W * 16 * 577 * 577 * 64
for name, ts in times.items():
    med = statistics.median(ts)
    print(f"{name:42s} median {med:7.1f} us  min {min(ts):7.1f} us  {flop / med / 1e6:6.0f} TFLOP/s", flush=True)
