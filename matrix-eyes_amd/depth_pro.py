"""Host-side mirror of the reference's `depth_pro` module interface over the C ABI.

reference src/depth_pro/mod.rs:
  pub const IMG_SIZE                                   (mod.rs:33)
  DepthProModelLoader::new(checkpoint_path, convert_checkpoints)   (mod.rs:167-172)
  DepthProModelLoader::extract_depth(img, f_norm, device, pl)      (mod.rs:251-363)
and the pub(super) module forwards the C ABI also exposes (vit.rs:328, encoder.rs:218,
decoder.rs:153, fov.rs:40).  Arrays are numpy (host) or torch CUDA tensors (device, zero copy).
"""
import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib as L
from .config import ModelConfig, expected_weights

IMG_SIZE = 384 * 4  # mod.rs:33


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _in_ptr(x, dtype):
    """(void* , keepalive) of a contiguous array of `dtype`; numpy -> host, torch -> its device."""
    if _is_torch(x):
        import torch
        want = {np.float32: torch.float32, np.uint8: torch.uint8, np.int32: torch.int32}[dtype]
        t = x if (x.dtype == want and x.is_contiguous()) else x.to(want).contiguous()
        return C.c_void_p(t.data_ptr()), t
    a = np.ascontiguousarray(x, dtype=dtype)
    return C.c_void_p(a.ctypes.data), a


def _out(out, shape, dtype=np.float32):
    """Output buffer: caller's torch/numpy array or a fresh numpy array."""
    if out is None:
        out = np.empty(shape, dtype=dtype)
    if _is_torch(out):
        assert out.is_contiguous() and tuple(out.shape) == tuple(shape), (out.shape, shape)
        return C.c_void_p(out.data_ptr()), out
    assert out.flags["C_CONTIGUOUS"] and out.shape == tuple(shape) and out.dtype == dtype
    return C.c_void_p(out.ctypes.data), out


class Context:
    """One GPU: stream, packed weights, workspaces (`me_ctx`)."""

    def __init__(self, device_id: int = 0, dtype: str = "f16", cfg: Optional[ModelConfig] = None):
        self.lib = L.load_library()
        self.cfg = cfg or ModelConfig()
        self.dtype = dtype
        self._h = C.c_void_p()
        ccfg = self.cfg.to_c()
        code = {"f16": L.ME_DTYPE_F16, "bf16": L.ME_DTYPE_BF16, "fp8": L.ME_DTYPE_FP8}[dtype]
        rc = self.lib.me_ctx_create(device_id, code, C.byref(ccfg), C.byref(self._h))
        if rc != L.ME_OK:
            raise L.MatrixEyesError(rc, self.lib.me_last_error(None).decode())
        self._progress_ref = None

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.me_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        L.check(self.lib, self._h, rc)

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        self._check(self.lib.me_ctx_synchronize(self._h))

    def status_flags(self) -> int:
        """Status bits the kernels raised since the last call, read and cleared (matrix_eyes_hip.h ME_STATUS_*:
        1 = an f16 operand store met a magnitude beyond 65504)."""
        flags = C.c_uint32(0)
        self._check(self.lib.me_status_flags(self._h, C.byref(flags)))
        return int(flags.value)

    def calibrate(self):
        """The two fixed calibration loops (csrc/calibrate.hip): what THIS device gives on an MFMA-only loop and on a
        device copy -- bench.py prints it beside the step time so that runs on different boxes can be compared."""
        out = (C.c_double * 6)()
        self._check(self.lib.me_calibrate(self._h, out))
        return {"mfma_tflops": out[0], "mfma_clock_ghz": out[1], "copy_gbs": out[2], "mfma_loop_ms": out[3],
                "copy_ms": out[4], "cus": int(out[5])}

    def ln_fusion_state(self):
        """(fused, fallbacks): whether the ViT's residual launches still carry the LayerNorm behind them, and how many
        steps were run again on stand-alone LayerNorm launches after an ME_STATUS_SYNC_TIMEOUT
        (matrix_eyes_hip.h me_ln_fusion_state)."""
        fused, fallbacks = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.me_ln_fusion_state(self._h, C.byref(fused), C.byref(fallbacks)))
        return bool(fused.value), int(fallbacks.value)

    def last_mesh_timing(self):
        """legs of the last output_mesh(".obj") call in ms: {mesh, format, d2h, file} and the text size"""
        ms = (C.c_double * 4)()
        n = C.c_int64()
        self._check(self.lib.me_last_mesh_timing(self._h, ms, C.byref(n)))
        return {"mesh_ms": ms[0], "format_ms": ms[1], "d2h_ms": ms[2], "file_ms": ms[3], "bytes": int(n.value)}

    def set_output_overlap(self, on=True):
        """The output back end on a second stream, ordered behind the extract_depth call that wrote the depth buffer it
        reads (matrix_eyes_hip.h me_ctx_set_output_overlap): queue image i + 1's extract_depth, then make image i's
        DeviceDepthMap calls."""
        self._check(self.lib.me_ctx_set_output_overlap(self._h, 1 if on else 0))

    def set_write_behind(self, files_in_flight=2):
        """OBJ files written by host threads behind the caller, up to `files_in_flight` at a time; 0 / False: the
        synchronous form (matrix_eyes_hip.h me_ctx_set_write_behind)."""
        self._check(self.lib.me_ctx_set_write_behind(self._h, int(files_in_flight)))

    def output_flush(self):
        """Waits for every pending write-behind file; raises for a failed write (me_output_flush)."""
        self._check(self.lib.me_output_flush(self._h))

    def weight_arena_layout(self) -> int:
        """Hash of the weight arena's layout; contexts that exchange arenas must agree on it."""
        return int(self.lib.me_weight_arena_layout(self._h))

    def set_graph(self, on: bool = True):
        """Run device-pointer extract_depth calls as one captured hipGraph (matrix_eyes_hip.h me_ctx_set_graph)."""
        self._check(self.lib.me_ctx_set_graph(self._h, 1 if on else 0))

    @property
    def graph_launch_count(self) -> int:
        """extract_depth calls replayed from the captured hipGraph so far (matrix_eyes_hip.h)."""
        return int(self.lib.me_graph_launch_count(self._h))

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self.lib.me_ctx_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def set_progress(self, fn):
        """fn(pos: float, message: Optional[str]) — ProgressListener (mod.rs:366-372)."""
        if fn is None:
            self._progress_ref = None
            self._check(self.lib.me_ctx_set_progress(self._h, L.PROGRESS_FN(0), None))
            return
        cb = L.PROGRESS_FN(lambda user, pos, msg: fn(pos, msg.decode() if msg else None))
        self._progress_ref = cb
        self._check(self.lib.me_ctx_set_progress(self._h, cb, None))

    def profile_enable(self, on: bool = True):
        self._check(self.lib.me_profile_enable(self._h, 1 if on else 0))

    def profile_report(self):
        import json
        buf = C.create_string_buffer(1 << 16)
        self._check(self.lib.me_profile_report(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    # -- weights (mod.rs:174-249 load_record)
    def expected_weights(self):
        n = self.lib.me_expected_weight_count(self._h)
        out = []
        name, dims, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int32()
        for i in range(n):
            self._check(self.lib.me_expected_weight(self._h, i, C.byref(name), dims, C.byref(nd)))
            out.append((name.value.decode(), tuple(dims[j] for j in range(nd.value))))
        return out

    def load_weight(self, name: str, tensor):
        """tensor: torch (f16/f32, CPU or CUDA) or numpy (float16/float32), PyTorch layout."""
        if _is_torch(tensor):
            import torch
            t = tensor.detach().contiguous()
            if t.dtype == torch.float16:
                wd = L.ME_WEIGHT_F16
            else:   # bf16, f64, integer buffers: through f32
                t, wd = t.to(torch.float32), L.ME_WEIGHT_F32
            ptr, shape, keep = C.c_void_p(t.data_ptr()), tuple(t.shape), t
        else:
            a = np.ascontiguousarray(tensor)
            if a.dtype == np.float16:
                wd = L.ME_WEIGHT_F16
            else:
                a, wd = a.astype(np.float32), L.ME_WEIGHT_F32
            ptr, shape, keep = C.c_void_p(a.ctypes.data), a.shape, a
        dims = (C.c_int64 * len(shape))(*shape)
        self._check(self.lib.me_load_weight(self._h, name.encode(), ptr, wd, dims, len(shape)))
        del keep

    def load_state_dict(self, state: Dict[str, object]):
        """Every tensor of `state` the model has a slot for, then me_weights_finalize (missing keys are errors:
        mod.rs:241-243).  Keys the model does not use are skipped and kept in `unused_keys`, as in the
        reference: each of its four parts is applied from the full snapshot list and only `result.errors` and
        `result.missing` are checked (mod.rs:236-243).  A tensor of the wrong shape stays an error."""
        expected = {n for n, _ in self.expected_weights()}
        self.unused_keys = [n for n in state if n not in expected]
        for name, t in state.items():
            if name in expected:
                self.load_weight(name, t)
        self._check(self.lib.me_weights_finalize(self._h))

    def load_checkpoint_pt(self, path: str):
        """me_load_checkpoint_pt: the library's own reader of `torch.save` archives (mod.rs:229-249)."""
        self._check(self.lib.me_load_checkpoint_pt(self._h, os.fsencode(path)))
        n = self.lib.me_unused_weight_count(self._h)
        self.unused_keys = [self.lib.me_unused_weight_name(self._h, i).decode() for i in range(n)]

    def weight_arena_bytes(self) -> int:
        return int(self.lib.me_weight_arena_bytes(self._h))

    def weight_arena_tensor(self):
        """The packed weight arena as a zero-copy uint8 torch tensor on this context's GPU (for a
        caller-side collective)."""
        import torch

        class _DevMem:
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1",
                                                 "data": (ptr, False), "version": 2}
        ptr = self.lib.me_weight_arena_ptr(self._h)
        return torch.as_tensor(_DevMem(int(ptr), self.weight_arena_bytes()), device="cuda")

    def adopt_weights(self):
        self._check(self.lib.me_weights_adopt(self._h))

    def bcast_weights(self, unique_id: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self.lib.me_bcast_weights(self._h, buf, rank, nranks))

    def rccl_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        rc = self.lib.me_rccl_unique_id(buf)
        if rc != L.ME_OK:
            raise L.MatrixEyesError(rc, "ncclGetUniqueId failed")
        return buf.raw

    # -- forwards
    def preprocess_u8(self, rgb, out=None):
        B = rgb.shape[0]
        S = self.cfg.img_size
        p, keep = _in_ptr(rgb, np.uint8)
        po, out = _out(out, (B, 3, S, S))
        self._check(self.lib.me_preprocess_u8(self._h, p, B, po))
        return out

    def vit_forward_features(self, which: int, xs, intermediate_blocks: Sequence[int] = ()):
        """vit.rs:328-346 -> (final [W,T,C], [intermediate ...])"""
        W = xs.shape[0]
        T, Cd = self.cfg.tokens, self.cfg.embed_dim
        p, keep = _in_ptr(xs, np.float32)
        n = len(intermediate_blocks)
        blocks = (C.c_int32 * max(n, 1))(*intermediate_blocks)
        final = np.empty((W, T, Cd), np.float32)
        inter = [np.empty((W, T, Cd), np.float32) for _ in range(n)]
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in inter])
        self._check(self.lib.me_vit_forward_features(self._h, which, p, W, blocks, n,
                                                     C.c_void_p(final.ctypes.data), ptrs))
        return final, inter

    def _enc_shapes(self, B):
        c, S = self.cfg, self.cfg.img_size
        return [(B, c.dec_dim, S // 2, S // 2), (B, c.enc_dims[0], S // 4, S // 4),
                (B, c.enc_dims[1], S // 8, S // 8), (B, c.enc_dims[2], S // 16, S // 16),
                (B, c.enc_dims[3], S // 32, S // 32)]

    def encoder_forward_encodings(self, x):
        """encoder.rs:218-335 -> 5 NCHW f32 arrays"""
        B = x.shape[0]
        p, keep = _in_ptr(x, np.float32)
        outs = [np.empty(s, np.float32) for s in self._enc_shapes(B)]
        ptrs = (C.c_void_p * 5)(*[a.ctypes.data for a in outs])
        self._check(self.lib.me_encoder_forward_encodings(self._h, p, B, ptrs))
        return outs

    def decoder_forward(self, encodings):
        """decoder.rs:153-208 -> (features, lowres_features)"""
        if len(encodings) != 5:   # decoder.rs:161-165
            raise L.MatrixEyesError(2, f"got encoder output levels {len(encodings)}, expected levels 5")
        B = encodings[0].shape[0]
        keep = [_in_ptr(e, np.float32) for e in encodings]
        ptrs = (C.c_void_p * 5)(*[k[0].value for k in keep])
        shapes = self._enc_shapes(B)
        feat = np.empty(shapes[0], np.float32)
        low = np.empty((B, self.cfg.dec_dim) + shapes[4][2:], np.float32)
        self._check(self.lib.me_decoder_forward(self._h, ptrs, B, C.c_void_p(feat.ctypes.data),
                                                C.c_void_p(low.ctypes.data)))
        return feat, low

    def head_forward(self, features):
        """mod.rs:323-338 -> canonical inverse depth [B,S,S]"""
        B, S = features.shape[0], self.cfg.img_size
        p, keep = _in_ptr(features, np.float32)
        out = np.empty((B, S, S), np.float32)
        self._check(self.lib.me_head_forward(self._h, p, B, C.c_void_p(out.ctypes.data)))
        return out

    def fov_forward(self, x, lowres_feature):
        """fov.rs:40-88 -> fov_deg [B]"""
        B = x.shape[0]
        p, k1 = _in_ptr(x, np.float32)
        pl, k2 = _in_ptr(lowres_feature, np.float32)
        out = np.empty((B,), np.float32)
        self._check(self.lib.me_fov_forward(self._h, p, pl, B, C.c_void_p(out.ctypes.data)))
        return out

    def extract_depth(self, img, f_norm=None, out=None, want_fov=False):
        """mod.rs:251-363.  img: f32 [B,3,S,S] or u8 [B,S,S,3]; f_norm: None (FOV head), a float,
        or [B] floats.  Returns inverse depth [B,S,S] (and fov_deg [B] when want_fov)."""
        S = self.cfg.img_size
        is_u8 = (img.dtype == np.uint8) if not _is_torch(img) else (str(img.dtype) == "torch.uint8")
        B = img.shape[0]
        p, keep = _in_ptr(img, np.uint8 if is_u8 else np.float32)
        fn_ptr, fn_keep = C.c_void_p(0), None
        if f_norm is not None and _is_torch(f_norm) and f_norm.is_cuda:
            # [B] f32 on the device: nothing in the call touches the host (eligible for the captured graph)
            fn_keep = f_norm.float().contiguous().reshape(B)
            fn_ptr = C.c_void_p(fn_keep.data_ptr())
        elif f_norm is not None:
            fn_keep = np.ascontiguousarray(np.broadcast_to(np.asarray(f_norm, np.float32), (B,)))
            fn_ptr = C.c_void_p(fn_keep.ctypes.data)
        po, out = _out(out, (B, S, S))
        fov = np.empty((B,), np.float32) if (want_fov and f_norm is None) else None
        fov_ptr = C.c_void_p(fov.ctypes.data) if fov is not None else C.c_void_p(0)
        fn = self.lib.me_extract_depth_u8 if is_u8 else self.lib.me_extract_depth
        self._check(fn(self._h, p, B, fn_ptr, po, fov_ptr))
        return (out, fov) if want_fov else out


class DepthProModelLoader:
    """reference `depth_pro::DepthProModelLoader` (mod.rs:120-123,166-172).

    The reference re-opens the checkpoint and builds/drops each stage's modules per call
    (mod.rs:276-351); here the weights are loaded once into the context and stay resident.
    `checkpoint_path` may be a PyTorch `.pt` state dict (torch.load) — or None / "synthetic" for
    the seeded synthetic checkpoint used when depth_pro.pt is not available."""

    def __init__(self, checkpoint_path: Optional[str] = "./checkpoints/depth_pro.pt",
                 convert_checkpoints: bool = False, cfg: Optional[ModelConfig] = None):
        self.checkpoint_path = checkpoint_path
        self.convert_checkpoints = convert_checkpoints  # .mpk cache: Burn-private, not produced
        self.cfg = cfg or ModelConfig()
        self._ctx = None

    def _state_dict(self):
        if self.checkpoint_path in (None, "synthetic"):
            from .synthetic import synthetic_checkpoint
            return synthetic_checkpoint(self.cfg)
        if not os.path.exists(self.checkpoint_path):
            # LoaderError::Pytorch (mod.rs:426)
            raise L.MatrixEyesError(7, f"Failed to load depth model: {self.checkpoint_path}: no such file")
        import torch
        state = torch.load(self.checkpoint_path, map_location="cpu", weights_only=True)
        return state.get("state_dict", state) if isinstance(state, dict) else state

    def context(self, device: int = 0, dtype: str = "f16") -> Context:
        if self._ctx is None:
            ctx = Context(device, dtype, self.cfg)
            if self.checkpoint_path in (None, "synthetic") or os.environ.get("ME_TORCH_LOAD"):
                ctx.load_state_dict(self._state_dict())
            else:
                # the library maps and parses the .pt archive itself (me_load_checkpoint_pt); ME_TORCH_LOAD=1
                # goes through torch.load instead
                ctx.load_checkpoint_pt(self.checkpoint_path)
            self._ctx = ctx
        return self._ctx

    def extract_depth(self, img, f_norm: Optional[float], device: int = 0, pl=None, dtype="f16"):
        """img [1,3,1536,1536] f32 (reconstruction.rs:116-124 layout) -> inverse depth [1536,1536]
        (mod.rs:336-338 squeezes the batch: batch 1 only, like the reference)."""
        if img.shape[0] != 1:
            raise L.MatrixEyesError(2, "extract_depth: batch must be 1 (mod.rs:336-338)")
        ctx = self.context(device, dtype)
        ctx.set_progress(pl)
        try:
            return ctx.extract_depth(img, f_norm)[0]
        finally:
            ctx.set_progress(None)
