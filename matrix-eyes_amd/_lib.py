"""ctypes binding of include/matrix_eyes_hip.h and include/matrix_eyes_hip_ops.h."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

ME_OK = 0
ERROR_NAMES = {
    1: "BAD_ARG", 2: "BAD_SHAPE", 3: "MISSING_WEIGHT", 4: "BAD_WEIGHT", 5: "HIP", 6: "RCCL",
    7: "IO", 8: "NOT_READY", 9: "OOM", 10: "OVERFLOW",
}
ME_STATUS_OVERFLOW_16BIT = 1
ME_STATUS_SYNC_TIMEOUT = 2
ME_DTYPE_F16, ME_DTYPE_BF16, ME_DTYPE_FP8 = 0, 1, 2
ME_WEIGHT_F32, ME_WEIGHT_F16, ME_WEIGHT_BF16, ME_WEIGHT_F64 = 0, 1, 2, 3
ME_VIT_PATCH_ENCODER, ME_VIT_IMAGE_ENCODER, ME_VIT_FOV_ENCODER = 0, 1, 2


class MatrixEyesError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ME_ERR_{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class CModelConfig(C.Structure):
    _fields_ = [
        ("grid", C.c_int32), ("embed_dim", C.c_int32), ("num_heads", C.c_int32),
        ("depth", C.c_int32), ("tap_blocks", C.c_int32 * 2), ("enc_dims", C.c_int32 * 4),
        ("dec_dim", C.c_int32), ("head_dims", C.c_int32 * 2), ("ln_eps", C.c_float),
        ("align_corners", C.c_int32), ("split_operands", C.c_int32), ("fp8_linears", C.c_int32),
    ]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_float, C.c_char_p)

_vp, _i32, _i64, _f32, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32

# name -> (restype, argtypes); every symbol include/*.h declares
SIGNATURES = {
    "me_abi_version": (_i32, []),
    "me_default_config": (_i32, [C.POINTER(CModelConfig)]),
    "me_ctx_create": (_i32, [_i32, _i32, C.POINTER(CModelConfig), C.POINTER(_vp)]),
    "me_ctx_destroy": (None, [_vp]),
    "me_last_error": (C.c_char_p, [_vp]),
    "me_ctx_set_progress": (_i32, [_vp, PROGRESS_FN, _vp]),
    "me_ctx_set_stream": (_i32, [_vp, _vp]),
    "me_ctx_synchronize": (_i32, [_vp]),
    "me_status_flags": (_i32, [_vp, C.POINTER(_u32)]),
    "me_ln_fusion_state": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "me_ctx_set_output_overlap": (_i32, [_vp, _i32]),
    "me_load_weight": (_i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    "me_expected_weight_count": (_i32, [_vp]),
    "me_expected_weight": (_i32, [_vp, _i32, C.POINTER(C.c_char_p), C.POINTER(_i64), C.POINTER(_i32)]),
    "me_weights_finalize": (_i32, [_vp]),
    "me_load_checkpoint_pt": (_i32, [_vp, C.c_char_p]),
    "me_unused_weight_count": (_i32, [_vp]),
    "me_unused_weight_name": (C.c_char_p, [_vp, _i32]),
    "me_weight_arena_bytes": (_i64, [_vp]),
    "me_weight_arena_ptr": (_vp, [_vp]),
    "me_weights_adopt": (_i32, [_vp]),
    "me_weight_arena_layout": (C.c_uint64, [_vp]),
    "me_rccl_unique_id": (_i32, [_vp]),
    "me_bcast_weights": (_i32, [_vp, _vp, _i32, _i32]),
    "me_preprocess_u8": (_i32, [_vp, _vp, _i32, _vp]),
    "me_vit_forward_features": (_i32, [_vp, _i32, _vp, _i32, C.POINTER(_i32), _i32, _vp, C.POINTER(_vp)]),
    "me_encoder_forward_encodings": (_i32, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "me_decoder_forward": (_i32, [_vp, C.POINTER(_vp), _i32, _vp, _vp]),
    "me_head_forward": (_i32, [_vp, _vp, _i32, _vp]),
    "me_fov_forward": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "me_extract_depth": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "me_extract_depth_u8": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "me_ctx_set_graph": (_i32, [_vp, _i32]),
    "me_graph_launch_count": (_i64, [_vp]),
    "me_depth_clamp_minmax": (_i32, [_vp, _vp, _i64, C.POINTER(_f32), C.POINTER(_f32)]),
    "me_depth_clamp_minmax_async": (_i32, [_vp, _vp, _i64, _vp]),
    "me_stereogram": (_i32, [_vp, _vp, _i32, _i32, _f32, _f32, _i32, _i32, _f32, _vp, _vp]),
    "me_stereogram_dev_range": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _i32, _f32, _vp, _vp]),
    "me_depthmap_rgb_dev_range": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "me_depthmap_rgb": (_i32, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "me_mesh_index": (_i32, [_vp, _vp, _i32, _i32, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "me_mesh_vertices": (_i32, [_vp, _vp, _i32, _i32, _vp, _i64, _u32, _u32, _vp, _vp]),
    "me_output_mesh": (_i32, [_vp, _vp, _i32, _i32, _u32, _u32, C.c_char_p, C.c_char_p, _i32, _vp]),
    "me_mesh_obj_text": (_i32, [_vp, _vp, _i32, _i32, _u32, _u32, C.c_char_p, _i32, _vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "me_last_mesh_timing": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "me_ctx_set_write_behind": (_i32, [_vp, _i32]),
    "me_output_flush": (_i32, [_vp]),
    # matrix_eyes_hip_ops.h
    "me_op_linear": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32]),
    "me_op_linear_residual": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32]),
    "me_op_attention": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "me_op_attention_prescaled": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32]),
    "me_op_linear_scaled_cols": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _i32]),
    "me_op_layernorm": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32]),
    "me_op_conv2d": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                            _vp, _vp, _i32, _i32, _i32, _i32]),
    "me_op_head_final": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _i32]),
    "me_op_conv_transpose2x2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp,
                                       _i32, _i32]),
    "me_op_quantize_fp8": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "me_op_attention_fp8": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    "me_op_scale_index": (_i64, [_i64, _i32, _i64, _i32]),
    "me_op_layernorm_fp8": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32]),
    "me_op_linear_fp8": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "me_op_linear_fp8_segments": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp),
                                         C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp, _vp]),
    "me_op_linear_segments": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                     _vp, _vp, _i32, _i32]),
    "me_op_linear_residual_layernorm": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                               C.POINTER(_vp), C.POINTER(_vp), _f32, _vp, _vp]),
    "me_op_linear_residual_layernorm_fp8": (_i32, [_vp, _i32, _i32, _i32, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                                   C.POINTER(_vp), C.POINTER(_vp), _f32, _vp, _vp, _vp]),
    "me_op_linear_fp8_residual_layernorm": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, C.POINTER(_vp), C.POINTER(_vp),
                                                   C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _f32, _vp, _vp, _vp]),
    "me_op_format_f64": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp]),
    "me_calibrate": (_i32, [_vp, C.POINTER(C.c_double)]),
    "me_op_cast_to16": (_i32, [_vp, _vp, _vp, _i64]),
    "me_op_cast_to32": (_i32, [_vp, _vp, _vp, _i64]),
    "me_profile_enable": (_i32, [_vp, _i32]),
    "me_profile_report": (_i32, [_vp, C.c_char_p, _i64]),
    "me_op_gemm_config_count": (_i32, []),
    "me_op_gemm_config_name": (C.c_char_p, [_i32]),
}

_lib = None


def library_path() -> str:
    return os.environ.get("MATRIX_EYES_HIP_LIB", os.path.join(_HERE, "libmatrixeyes_hip.so"))


def load_library():
    """Loads libmatrixeyes_hip.so and declares every prototype.  Raises if the library has not
    been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise MatrixEyesError(5, f"{path} not found: build it first (__graft_entry__.build()); "
                                 "there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, ctx, rc):
    if rc != ME_OK:
        msg = lib.me_last_error(ctx)
        raise MatrixEyesError(rc, msg.decode("utf-8", "replace") if msg else "")
