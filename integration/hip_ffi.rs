//! src/hip_ffi.rs for zlogic/matrix-eyes: the `extern "C"` surface of include/matrix_eyes_hip.h, one item per
//! header declaration, in header order.  NOT COMPILED in the build image (no Rust toolchain there):
//! tests/test_integration_files.py holds this file to the header (same symbol set, same argument counts); the
//! compiled and tested callers of the same entry points are matrix-eyes_amd/_lib.py (ctypes) and
//! matrix-eyes_amd/host/ (C++).
#![cfg(feature = "hip")]
#![allow(dead_code)]
use std::ffi::{c_char, c_void};

pub const ME_ABI_VERSION: i32 = 4;

// status codes
pub const ME_OK: i32 = 0;
pub const ME_ERR_BAD_ARG: i32 = 1;
pub const ME_ERR_BAD_SHAPE: i32 = 2;
pub const ME_ERR_MISSING_WEIGHT: i32 = 3; // LoaderError::RecorderMissing (mod.rs:241-243)
pub const ME_ERR_BAD_WEIGHT: i32 = 4; // LoaderError::RecorderErrors (mod.rs:238-240)
pub const ME_ERR_HIP: i32 = 5;
pub const ME_ERR_RCCL: i32 = 6;
pub const ME_ERR_IO: i32 = 7; // OutputError::Io / LoaderError::Pytorch
pub const ME_ERR_NOT_READY: i32 = 8;
pub const ME_ERR_OOM: i32 = 9;
pub const ME_ERR_OVERFLOW: i32 = 10; // an activation left the f16 operand range (me_status_flags)
// me_status_flags bits
pub const ME_STATUS_OVERFLOW_16BIT: i32 = 1;
/// me_status_flags: a workgroup gave up waiting for its neighbours' LayerNorm statistics (never on a healthy device)
pub const ME_STATUS_SYNC_TIMEOUT: i32 = 2;

// MFMA operand type
pub const ME_DTYPE_F16: i32 = 0;
pub const ME_DTYPE_BF16: i32 = 1;
pub const ME_DTYPE_FP8: i32 = 2;
// element type of a tensor handed to me_load_weight
pub const ME_WEIGHT_F32: i32 = 0;
pub const ME_WEIGHT_F16: i32 = 1;
pub const ME_WEIGHT_BF16: i32 = 2;
pub const ME_WEIGHT_F64: i32 = 3;
// which DINOv2 ViT-L of the three (encoder.rs:23-24, fov.rs:25)
pub const ME_VIT_PATCH_ENCODER: i32 = 0;
pub const ME_VIT_IMAGE_ENCODER: i32 = 1;
pub const ME_VIT_FOV_ENCODER: i32 = 2;
// output.rs:34-38 VertexMode
pub const ME_VERTEX_PLAIN: i32 = 0;
pub const ME_VERTEX_COLOR: i32 = 1;
pub const ME_VERTEX_TEXTURE: i32 = 2;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct MeModelConfig {
    // me_model_config
    pub grid: i32,
    pub embed_dim: i32,
    pub num_heads: i32,
    pub depth: i32,
    pub tap_blocks: [i32; 2],
    pub enc_dims: [i32; 4],
    pub dec_dim: i32,
    pub head_dims: [i32; 2],
    pub ln_eps: f32,
    pub align_corners: i32,
    pub split_operands: i32,
    pub fp8_linears: i32,
}
#[repr(C)]
pub struct MeCtx {
    _private: [u8; 0],
}
pub type MeProgressFn = Option<unsafe extern "C" fn(user: *mut c_void, pos: f32, msg: *const c_char)>;

extern "C" {
    pub fn me_abi_version() -> i32;
    pub fn me_default_config(cfg: *mut MeModelConfig) -> i32;
    pub fn me_ctx_create(device_id: i32, dtype: i32, cfg: *const MeModelConfig, out: *mut *mut MeCtx) -> i32;
    pub fn me_ctx_destroy(ctx: *mut MeCtx);
    pub fn me_last_error(ctx: *const MeCtx) -> *const c_char;
    pub fn me_ctx_set_progress(ctx: *mut MeCtx, f: MeProgressFn, user: *mut c_void) -> i32;
    pub fn me_ctx_set_stream(ctx: *mut MeCtx, hip_stream: *mut c_void) -> i32;
    pub fn me_ctx_synchronize(ctx: *mut MeCtx) -> i32;
    pub fn me_status_flags(ctx: *mut MeCtx, flags: *mut u32) -> i32;
    pub fn me_ln_fusion_state(ctx: *mut MeCtx, fused: *mut i32, fallbacks: *mut i32) -> i32;
    pub fn me_ctx_set_output_overlap(ctx: *mut MeCtx, on: i32) -> i32;
    pub fn me_load_weight(ctx: *mut MeCtx, name: *const c_char, data: *const c_void, weight_dtype: i32, dims: *const i64, ndim: i32) -> i32;
    pub fn me_expected_weight_count(ctx: *const MeCtx) -> i32;
    pub fn me_expected_weight(ctx: *const MeCtx, index: i32, name: *mut *const c_char, dims: *mut i64, ndim: *mut i32) -> i32;
    pub fn me_weights_finalize(ctx: *mut MeCtx) -> i32;
    pub fn me_load_checkpoint_pt(ctx: *mut MeCtx, path: *const c_char) -> i32;
    pub fn me_unused_weight_count(ctx: *const MeCtx) -> i32;
    pub fn me_unused_weight_name(ctx: *const MeCtx, index: i32) -> *const c_char;
    pub fn me_weight_arena_bytes(ctx: *const MeCtx) -> i64;
    pub fn me_weight_arena_ptr(ctx: *const MeCtx) -> *mut c_void;
    pub fn me_weights_adopt(ctx: *mut MeCtx) -> i32;
    pub fn me_weight_arena_layout(ctx: *const MeCtx) -> u64;
    pub fn me_rccl_unique_id(id128: *mut c_void) -> i32;
    pub fn me_bcast_weights(ctx: *mut MeCtx, id128: *const c_void, rank: i32, nranks: i32) -> i32;
    pub fn me_preprocess_u8(ctx: *mut MeCtx, rgb: *const u8, batch: i32, img: *mut f32) -> i32;
    pub fn me_vit_forward_features(ctx: *mut MeCtx, which_vit: i32, xs: *const f32, windows: i32, intermediate_blocks: *const i32, n_intermediate: i32, final_out: *mut f32, intermediate_out: *const *mut f32) -> i32;
    pub fn me_encoder_forward_encodings(ctx: *mut MeCtx, x: *const f32, batch: i32, encodings: *const *mut f32) -> i32;
    pub fn me_decoder_forward(ctx: *mut MeCtx, encodings: *const *const f32, batch: i32, features: *mut f32, lowres_features: *mut f32) -> i32;
    pub fn me_head_forward(ctx: *mut MeCtx, features: *const f32, batch: i32, canonical_inverse_depth: *mut f32) -> i32;
    pub fn me_fov_forward(ctx: *mut MeCtx, x: *const f32, lowres_feature: *const f32, batch: i32, fov_deg: *mut f32) -> i32;
    pub fn me_extract_depth(ctx: *mut MeCtx, img: *const f32, batch: i32, f_norm: *const f32, inverse_depth: *mut f32, fov_deg_out: *mut f32) -> i32;
    pub fn me_extract_depth_u8(ctx: *mut MeCtx, rgb: *const u8, batch: i32, f_norm: *const f32, inverse_depth: *mut f32, fov_deg_out: *mut f32) -> i32;
    pub fn me_ctx_set_graph(ctx: *mut MeCtx, on: i32) -> i32;
    pub fn me_graph_launch_count(ctx: *const MeCtx) -> i64;
    pub fn me_depth_clamp_minmax(ctx: *mut MeCtx, depth: *mut f32, count: i64, min_out: *mut f32, max_out: *mut f32) -> i32;
    pub fn me_depth_clamp_minmax_async(ctx: *mut MeCtx, depth: *mut f32, count: i64, minmax_dev: *mut f32) -> i32;
    pub fn me_stereogram(ctx: *mut MeCtx, depth: *const f32, rows: i32, cols: i32, min_depth: f32, max_depth: f32, out_w: i32, out_h: i32, amplitude: f32, noise: *const u8, out: *mut u8) -> i32;
    pub fn me_stereogram_dev_range(ctx: *mut MeCtx, depth: *const f32, rows: i32, cols: i32, minmax_dev: *const f32, out_w: i32, out_h: i32, amplitude: f32, noise: *const u8, out: *mut u8) -> i32;
    pub fn me_depthmap_rgb(ctx: *mut MeCtx, depth: *const f32, count: i64, min_depth: f32, max_depth: f32, rgb: *mut u8) -> i32;
    pub fn me_depthmap_rgb_dev_range(ctx: *mut MeCtx, depth: *const f32, count: i64, minmax_dev: *const f32, rgb: *mut u8) -> i32;
    pub fn me_mesh_index(ctx: *mut MeCtx, depth: *const f32, width: i32, height: i32, vertex_index: *mut i32, nvertices: *mut i64, nfaces: *mut i64, faces: *mut i32) -> i32;
    pub fn me_mesh_vertices(ctx: *mut MeCtx, depth: *const f32, width: i32, height: i32, vertex_index: *const i32, nvertices: i64, original_width: u32, original_height: u32, uv: *mut f32, xyz: *mut f32) -> i32;
    pub fn me_mesh_obj_text(ctx: *mut MeCtx, depth: *const f32, width: i32, height: i32, original_width: u32, original_height: u32, stem: *const c_char, vertex_mode: i32, vertex_colors: *const u8, text_dev: *mut *const u8, nbytes: *mut i64) -> i32;
    pub fn me_last_mesh_timing(ctx: *const MeCtx, ms_out: *mut f64, text_bytes: *mut i64) -> i32;
    pub fn me_ctx_set_write_behind(ctx: *mut MeCtx, files_in_flight: i32) -> i32;
    pub fn me_output_flush(ctx: *mut MeCtx) -> i32;
    pub fn me_output_mesh(ctx: *mut MeCtx, depth: *const f32, width: i32, height: i32, original_width: u32, original_height: u32, destination_path: *const c_char, source_path: *const c_char, vertex_mode: i32, vertex_colors: *const u8) -> i32;
}
