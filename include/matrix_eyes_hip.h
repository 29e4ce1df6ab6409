/*
 * matrix_eyes_hip.h — C ABI of libmatrixeyes_hip.so, the MI355X (gfx950) back end for the
 * Depth Pro hot path of zlogic/matrix-eyes.
 *
 * The reference has no FFI seam; its only seam is the compile-time generic `B: Backend`
 * (reference src/reconstruction.rs:155-165, src/depth_pro/mod.rs:251-260).  Every entry point
 * below replaces one Burn-module `forward()` (or one `DepthMap` method) of the reference and
 * cites it.  A Rust caller swaps each `Tensor<B, D>` argument for a `(pointer, dims)` pair;
 * see INTEGRATION.md for the `extern "C"` block and the safe wrapper a maintainer would add.
 *
 * Conventions
 *   - Every function returns an `int32_t` status (ME_OK == 0).  Nothing aborts the process:
 *     shape-invariant violations that `panic!` in the reference (vit.rs:213-218,282,318-324;
 *     decoder.rs:161-165; mod.rs:43-46) come back as ME_ERR_BAD_SHAPE, a missing checkpoint
 *     key (mod.rs:241-243) as ME_ERR_MISSING_WEIGHT.  `me_last_error` gives the text.
 *   - Tensor pointers may be HOST or DEVICE (hipMalloc) addresses; the library asks the HIP
 *     runtime which (hipPointerGetAttributes) and stages host buffers itself.  Layouts are
 *     the reference's: row-major, NCHW for images/feature maps, [B, tokens, C] for token
 *     tensors, f32 elements unless stated.
 *   - A context owns one GPU, one stream, the packed weights and all workspaces; it is used by
 *     one host thread at a time (the reference is single-threaded: reconstruction.rs:61-64).
 *   - No compute entry point has a CPU fallback: without a GPU every one of them fails with
 *     ME_ERR_HIP.
 */
#ifndef MATRIX_EYES_HIP_H
#define MATRIX_EYES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ME_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------------------- */
enum {
    ME_OK = 0,
    ME_ERR_BAD_ARG = 1,        /* null pointer, unknown enum value, bad size            */
    ME_ERR_BAD_SHAPE = 2,      /* the reference's shape panics                          */
    ME_ERR_MISSING_WEIGHT = 3, /* LoaderError::RecorderMissing (mod.rs:241-243)         */
    ME_ERR_BAD_WEIGHT = 4,     /* LoaderError::RecorderErrors: unknown key, wrong shape */
    ME_ERR_HIP = 5,            /* a HIP runtime call or kernel launch failed            */
    ME_ERR_RCCL = 6,           /* an RCCL call failed                                   */
    ME_ERR_IO = 7,             /* OutputError::Io / LoaderError::Pytorch                */
    ME_ERR_NOT_READY = 8,      /* forward called before the weights were finalized      */
    ME_ERR_OOM = 9,            /* device allocation failed                              */
    ME_ERR_OVERFLOW = 10       /* an activation left the f16 operand range (me_status_flags) */
};

/* ---- me_status_flags bits ---------------------------------------------------------------- */
enum {
    ME_STATUS_OVERFLOW_16BIT = 1, /* a kernel rounded a magnitude beyond 65504 to an f16 operand (stored as +-inf) */
    ME_STATUS_SYNC_TIMEOUT = 2    /* a workgroup gave up waiting for its neighbours' LayerNorm statistics (the fused
                                     residual epilogue; needs its sibling workgroups co-resident: a second tenant on
                                     the device's CUs or a CU mask can break that): the step's result is not valid.
                                     The context then falls back to stand-alone LayerNorm launches (me_ln_fusion_state) */
};

/* ---- arithmetic type of the MFMA operands (accumulation is always f32) --------------- */
enum {
    ME_DTYPE_F16 = 0,  /* default: the checkpoint is fp16, so weights are exact          */
    ME_DTYPE_BF16 = 1,
    ME_DTYPE_FP8 = 2   /* BASELINE configs[3]: the ViT linears on MX block-scaled fp8 (e4m3 values, one e8m0
                          scale per 32 K-elements, v_mfma_scale_f32_16x16x128_f8f6f4); everything else f16 */
};

/* ---- element type of a weight tensor handed to me_load_weight ------------------------ */
enum { ME_WEIGHT_F32 = 0, ME_WEIGHT_F16 = 1, ME_WEIGHT_BF16 = 2, ME_WEIGHT_F64 = 3 };

/* ---- which DINOv2 ViT-L of the three in the model (encoder.rs:23-24, fov.rs:25) ------- */
enum { ME_VIT_PATCH_ENCODER = 0, ME_VIT_IMAGE_ENCODER = 1, ME_VIT_FOV_ENCODER = 2 };

/* ---- output.rs:34-38 ------------------------------------------------------------------ */
enum { ME_VERTEX_PLAIN = 0, ME_VERTEX_COLOR = 1, ME_VERTEX_TEXTURE = 2 };

/*
 * Model geometry.  me_default_config() fills in the reference's constants:
 *   vit.rs:17-19,349-358  grid 24 (IMG_SIZE 384 / PATCH_SIZE 16), embed 1024, depth 24, heads 16
 *   encoder.rs:227        taps after blocks 5 and 11
 *   mod.rs:262-263        ENCODER_FEATURE_DIMS [256,512,1024,1024], DECODER_FEATURES 256
 *   mod.rs:308-311        head last_dims [32, 1]
 * Smaller values (grid a multiple of 8, embed a multiple of 64 ...) exist so that parity tests
 * can run the same code on a model the CPU oracle finishes in seconds.  The image side is
 * always 4 * 16 * grid (mod.rs:33) and the merge paddings are grid/8 and grid/4
 * (encoder.rs:267-293 hard-codes 3 and 6 for grid 24).
 */
typedef struct me_model_config {
    int32_t grid;          /* tokens per window side; 24 */
    int32_t embed_dim;     /* 1024 */
    int32_t num_heads;     /* 16; embed_dim / num_heads must be 64 */
    int32_t depth;         /* 24 */
    int32_t tap_blocks[2]; /* {5, 11} */
    int32_t enc_dims[4];   /* {256, 512, 1024, 1024} */
    int32_t dec_dim;       /* 256 */
    int32_t head_dims[2];  /* {32, 1} */
    float ln_eps;          /* Burn LayerNormConfig default 1e-5 (vit.rs:141) */
    int32_t align_corners; /* bilinear pyramid (encoder.rs:128-137): 1 = Burn's historical
                              align_corners=true, 0 = half-pixel centres                  */
    int32_t split_operands; /* bit mask of the stages whose 16-bit activation operands are carried as hi + lo
                               pairs against duplicated weights (2x the MFMA work of that stage, operand
                               rounding 2^-22 instead of 2^-11): 1 = encoder upsample / fuse convs
                               (encoder.rs:307-325), 2 = fusion deconv + out_conv (decoder.rs:95-101), 4 = head
                               (mod.rs:323-333), 8 = decoder.convs (decoder.rs:189-195).  me_default_config: 3
                               (full-size depth error 7.5e-4 relative L2 against the fp32 reference; 0: 1.0e-3,
                               7: 5.8e-4, 15: 5.2e-4 -- DESIGN.md section 5) */
    int32_t fp8_linears;    /* ME_DTYPE_FP8 contexts only: bit mask of the ViT linears that run on MX fp8 -- 1 = qkv,
                               2 = proj, 4 = fc1, 8 = fc2 (vit.rs:60-62,74,120,122); the others stay on the 16-bit kernels.
                               0 = the default ME_FP8_LINEARS_DEFAULT.  profiles/r05_fp8_mask_budget.txt has depth error and
                               time per step for each mask */
} me_model_config;

#define ME_FP8_LINEARS_DEFAULT 13 /* qkv + fc1 + fc2: proj on fp8 buys 0.3 ms for 12 % more depth error */

typedef struct me_ctx me_ctx;

/* progress(user, fraction in [0,1], message-or-NULL): ProgressListener (mod.rs:366-372). */
typedef void (*me_progress_fn)(void* user, float pos, const char* message);

int32_t me_abi_version(void);
int32_t me_default_config(me_model_config* cfg);

/* reconstruction.rs:42-72 init_device + mod.rs:167 DepthProModelLoader::new.
   cfg == NULL means me_default_config. */
int32_t me_ctx_create(int32_t device_id, int32_t dtype, const me_model_config* cfg, me_ctx** out);
void me_ctx_destroy(me_ctx* ctx);
/* Text of the last failure on ctx (ctx == NULL: of the last failed me_ctx_create on this
   thread).  Never NULL. */
const char* me_last_error(const me_ctx* ctx);
int32_t me_ctx_set_progress(me_ctx* ctx, me_progress_fn fn, void* user);
/* Run on a caller-owned hipStream_t (e.g. the framework's current stream) instead of the
   context's own; NULL restores the own stream. */
int32_t me_ctx_set_stream(me_ctx* ctx, void* hip_stream);
int32_t me_ctx_synchronize(me_ctx* ctx);
/* Status bits raised by the kernels since the last call (ME_STATUS_*), read and cleared; synchronises the stream.
   The reference computes in f32 (decoder.rs:35-44: relu, conv, adds in f32); this back end rounds MFMA operands to
   16 bit.  An f16 operand holds magnitudes up to 65504: past it the operand is +-inf, and behind a conv + ReLU the
   branch drops out silently.  Every kernel that writes 16-bit operands therefore raises ME_STATUS_OVERFLOW_16BIT
   when that happens.  The word is sticky: no call clears it when it starts.  me_extract_depth[_u8] with a result in
   HOST memory checks it when it has finished, fails with ME_ERR_OVERFLOW itself (also for an overflow that an earlier
   asynchronous call left unread) and clears the bit it reports; with a DEVICE result the call is asynchronous and
   never touches the flag: it stays raised over any number of calls (and graph replays) until the caller asks here.
   bf16 operands (ME_DTYPE_BF16) have f32's range and never raise it.
   ME_STATUS_SYNC_TIMEOUT: a host-result call that meets it runs its step once more on stand-alone LayerNorm launches
   (bit for bit the ME_LN_FUSE=0 result) and succeeds; after device-result calls the NEXT me_extract_depth[_u8] on the
   context fails with ME_ERR_HIP (the earlier depth maps are invalid) and the one after that runs unfused. */
int32_t me_status_flags(me_ctx* ctx, uint32_t* flags);
/* Whether the residual launches of the ViT still carry the LayerNorm behind them (1) or the context has fallen back
   to stand-alone LayerNorm launches (0: after an ME_STATUS_SYNC_TIMEOUT, see above; me_last_error holds the note
   logged then), and how many steps were run again because of it.  Either pointer may be NULL. */
int32_t me_ln_fusion_state(me_ctx* ctx, int32_t* fused, int32_t* fallbacks);

/* ---- weights: mod.rs:174-249 load_record ---------------------------------------------
   Tensors are handed over under their PyTorch checkpoint names and layouts (SURVEY App. C:
   Linear [out,in], Conv2d [out,in,kh,kw], ConvTranspose2d [in,out,kh,kw]); the library
   repacks them for its kernels (the reference's PyTorchToBurnAdapter transposes instead).
   Unknown names and wrong shapes fail like the Applier does (mod.rs:238-240). */
int32_t me_load_weight(me_ctx* ctx, const char* name, const void* data, int32_t weight_dtype,
                       const int64_t* dims, int32_t ndim);
/* Number of tensors the model expects, and the i-th expected name/shape (for loaders). */
int32_t me_expected_weight_count(const me_ctx* ctx);
int32_t me_expected_weight(const me_ctx* ctx, int32_t index, const char** name, int64_t dims[4],
                           int32_t* ndim);
/* Fails with ME_ERR_MISSING_WEIGHT while any expected tensor is absent (mod.rs:241-243). */
int32_t me_weights_finalize(me_ctx* ctx);
/* mod.rs:229-249 load_record for a caller without a PyTorch-checkpoint reader of its own: maps the
   `torch.save` zip archive at `path` (the reference reads it through burn-store's PytorchStore, mod.rs:231),
   loads every tensor the model expects (f16 / f32 / bf16 / f64 storage) and finalizes.  Keys the model does
   not use are skipped, as in the reference, where each part is applied from the full snapshot list and only
   `result.errors` / `result.missing` are checked (mod.rs:236-243); they are listed by
   me_unused_weight_count / me_unused_weight_name.  An unreadable or malformed file is ME_ERR_IO
   (LoaderError::Pytorch), a tensor of the wrong shape ME_ERR_BAD_WEIGHT, an absent one
   ME_ERR_MISSING_WEIGHT. */
int32_t me_load_checkpoint_pt(me_ctx* ctx, const char* path);
int32_t me_unused_weight_count(const me_ctx* ctx);
const char* me_unused_weight_name(const me_ctx* ctx, int32_t index);
/* Size of the packed device weight arena, bytes. */
int64_t me_weight_arena_bytes(const me_ctx* ctx);
/* Device address of the arena, for a caller that moves the packed weights with its own collective
   (e.g. torch.distributed's RCCL communicator) instead of me_bcast_weights; after the bytes have
   arrived on a rank, me_weights_adopt marks every tensor loaded and finalizes. */
void* me_weight_arena_ptr(const me_ctx* ctx);
int32_t me_weights_adopt(me_ctx* ctx);
/* Hash of the arena's layout (dtype, me_model_config, split_operands, every tensor's offset and size).  Contexts
   that exchange arenas must agree on it: me_bcast_weights compares it with rank 0's before the payload moves and
   fails with ME_ERR_BAD_ARG on every rank when one differs; a caller that moves the bytes itself compares it before
   me_weights_adopt. */
uint64_t me_weight_arena_layout(const me_ctx* ctx);

/* ---- multi-GPU start-up: one RCCL broadcast of the packed arena, no collective later ---
   rank 0 finalizes its weights, every rank calls me_bcast_weights with the same 128-byte id
   (from me_rccl_unique_id on rank 0, shipped by the launcher's own store). */
int32_t me_rccl_unique_id(void* id128);
int32_t me_bcast_weights(me_ctx* ctx, const void* id128, int32_t rank, int32_t nranks);

/* ---- forward passes ------------------------------------------------------------------- */

/* reconstruction.rs:114-124: u8 HWC [B,S,S,3] -> f32 NCHW [B,3,S,S], ((x/255)-0.5)/0.5. */
int32_t me_preprocess_u8(me_ctx* ctx, const uint8_t* rgb, int32_t batch, float* img);

/* vit.rs:328-346 DinoVisionTransformer::forward_features.
   xs [W,3,16g,16g]; final_out [W,g*g+1,C] (layer-normed); intermediate_out[i] [W,g*g+1,C]
   (output of block intermediate_blocks[i], not normed).  A requested block that does not
   exist is ME_ERR_BAD_SHAPE (vit.rs:318-324). */
int32_t me_vit_forward_features(me_ctx* ctx, int32_t which_vit, const float* xs, int32_t windows,
                                const int32_t* intermediate_blocks, int32_t n_intermediate,
                                float* final_out, float* const* intermediate_out);

/* encoder.rs:218-335 DepthProEncoder::forward_encodings.  x [B,3,S,S], S = 64*grid;
   encodings[0..4] = [B,dec,S/2,S/2], [B,enc0,S/4,S/4], [B,enc1,S/8,S/8], [B,enc2,S/16,S/16],
   [B,enc3,S/32,S/32]. */
int32_t me_encoder_forward_encodings(me_ctx* ctx, const float* x, int32_t batch,
                                     float* const encodings[5]);

/* decoder.rs:153-208 MultiresConvDecoder::forward: features [B,dec,S/2,S/2],
   lowres_features [B,dec,S/32,S/32]. */
int32_t me_decoder_forward(me_ctx* ctx, const float* const encodings[5], int32_t batch,
                           float* features, float* lowres_features);

/* mod.rs:323-338 head[0..3] + ReLUs: features [B,dec,S/2,S/2] -> canonical inverse depth
   [B,S,S]. */
int32_t me_head_forward(me_ctx* ctx, const float* features, int32_t batch,
                        float* canonical_inverse_depth);

/* fov.rs:40-88 FOVNetwork::forward: x [B,3,S,S], lowres_feature [B,dec,S/32,S/32] ->
   fov_deg [B]. */
int32_t me_fov_forward(me_ctx* ctx, const float* x, const float* lowres_feature, int32_t batch,
                       float* fov_deg);

/* mod.rs:251-363 DepthProModelLoader::extract_depth, for a batch of independent images (the
   reference is batch 1; a batch here equals a loop of batch-1 calls).
   img [B,3,S,S]; f_norm NULL = estimate with the FOV head (mod.rs:343-358), else [B] values;
   inverse_depth [B,S,S] = clamp(canonical / f_norm, 1e-4, 1e4); fov_deg_out NULL or [B]. */
int32_t me_extract_depth(me_ctx* ctx, const float* img, int32_t batch, const float* f_norm,
                         float* inverse_depth, float* fov_deg_out);
/* The same from u8 HWC images, with reconstruction.rs:114-124 fused into the first kernel. */
int32_t me_extract_depth_u8(me_ctx* ctx, const uint8_t* rgb, int32_t batch, const float* f_norm,
                            float* inverse_depth, float* fov_deg_out);
/* The step as one hipGraph (off by default; ME_GRAPH=1 in the environment turns it on at me_ctx_create).  When
   on, a me_extract_depth / me_extract_depth_u8 call whose pointers all name device memory, on a context without a
   progress callback, is enqueued eagerly the first time, captured the second time and replayed with one
   hipGraphLaunch from its third identical invocation on (the shapes are static per batch size).  New weights,
   another stream, batch or pointer start over.  Results are bit-identical to eager launches; the GPU is never
   starved by the host either way, so this buys host CPU time, not depth-maps/s.
   me_graph_launch_count returns the number of replays so far. */
int32_t me_ctx_set_graph(me_ctx* ctx, int32_t on);
int64_t me_graph_launch_count(const me_ctx* ctx);

/* ---- output back end (src/output.rs) -------------------------------------------------- */

/* output.rs:44-75 DepthMap::new clamp to [1/250, 1/0.1] (in place) + inverse_depth_range. */
int32_t me_depth_clamp_minmax(me_ctx* ctx, float* depth, int64_t count, float* min_out,
                              float* max_out);

/* The same without the host round trip: depth and minmax_dev are DEVICE memory, the range {min, max} stays in
   minmax_dev[0..1] for me_stereogram_dev_range / me_depthmap_rgb_dev_range queued behind it on the context's
   stream (DepthMap::new -> output_image chained on the GPU, output.rs:44-75 + :100-121). */
int32_t me_depth_clamp_minmax_async(me_ctx* ctx, float* depth, int64_t count, float* minmax_dev);

/* output.rs:141-193 output_stereogram.  depth [rows,cols] already clamped (DepthMap.data);
   noise [out_h,out_w,3] is the reference's per-row rand stream made an input;
   out [out_h,out_w,3].  Bit-exact with the reference's f32 arithmetic. */
int32_t me_stereogram(me_ctx* ctx, const float* depth, int32_t rows, int32_t cols, float min_depth,
                      float max_depth, int32_t out_w, int32_t out_h, float amplitude,
                      const uint8_t* noise, uint8_t* out);

/* me_stereogram with the depth range read from device memory (me_depth_clamp_minmax_async). */
int32_t me_stereogram_dev_range(me_ctx* ctx, const float* depth, int32_t rows, int32_t cols,
                                const float* minmax_dev, int32_t out_w, int32_t out_h, float amplitude,
                                const uint8_t* noise, uint8_t* out);

/* output.rs:123-131 + 633-714 map_depth: depth [count] -> rgb [count,3] (before the Lanczos
   resize, which is the identity at the native size). */
int32_t me_depthmap_rgb(me_ctx* ctx, const float* depth, int64_t count, float min_depth,
                        float max_depth, uint8_t* rgb);

/* me_depthmap_rgb with the depth range read from device memory (me_depth_clamp_minmax_async). */
int32_t me_depthmap_rgb_dev_range(me_ctx* ctx, const float* depth, int64_t count, const float* minmax_dev,
                                  uint8_t* rgb);

/* output.rs:264-363 IndexedMesh::new + for_each_face + remap_face.
   depth [height,width] (DepthMap.data, stride `width`).  vertex_index [height*width]: the
   first-use vertex id or -1.  faces [nfaces,3] remapped vertex ids in the reference's
   emission order; faces may be NULL to only count, else it needs room for
   2*(width-1)*(height-1) triangles. */
int32_t me_mesh_index(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                      int32_t* vertex_index, int64_t* nvertices, int64_t* nfaces, int32_t* faces);

/* output.rs:228-249: per vertex id (sorted_vertices order) uv [nvertices,2] = (x/w, y/h) and
   xyz [nvertices,3] = (xm*(xn-0.5)*z, ym*(yn-0.5)*z, z), z = 1/depth, before the writer's
   sign flips.  vertex_index comes from me_mesh_index. */
int32_t me_mesh_vertices(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                         const int32_t* vertex_index, int64_t nvertices, uint32_t original_width,
                         uint32_t original_height, float* uv, float* xyz);

/* output.rs:100-112 + 195-261 output_mesh with ObjWriter (:484-630) / PlyWriter (:385-482):
   indexes the mesh on the GPU and writes `destination_path` (".obj" or ".ply", case-insensitive,
   anything else is ME_ERR_BAD_ARG; for ".obj" in texture mode also "<stem>.mtl" next to it).
   depth [height,width] is DepthMap.data (already clamped).  vertex_colors: NULL, or u8
   [height*width,3] = the source image resized to the depth map (only read in ME_VERTEX_COLOR
   mode; the Lanczos resize itself is the caller's). Text output is byte-identical to
   the reference's: numbers are printed like Rust's `{}` for f64 (shortest round-trip, never
   scientific). */
int32_t me_output_mesh(me_ctx* ctx, const float* depth, int32_t width, int32_t height,
                       uint32_t original_width, uint32_t original_height,
                       const char* destination_path, const char* source_path, int32_t vertex_mode,
                       const uint8_t* vertex_colors);

/* Write-behind for me_output_mesh(".obj") (BASELINE configs[4]: a batch of images, each ending in a file of 70 - 450 MB).
   files_in_flight >= 2: the call returns once the text sits in pinned host memory; a host thread writes the file (and
   the .mtl) while the caller goes on to the next image.  That many pinned buffers are used in turn, so that many files
   can be in flight; the next call waits for the oldest (1 is taken as 2) -- and for any pending write to its own
   destination path -- before it starts any work of its own.  A failed write is reported (ME_ERR_IO, me_last_error) by
   the call that next waits for it: a later me_output_mesh (which then has written nothing and can be repeated),
   me_ctx_set_write_behind, or me_output_flush, which waits for every pending file.  me_ctx_destroy flushes.  0 (the default): the reference's
   form, output_mesh returns with the file written. */
int32_t me_ctx_set_write_behind(me_ctx* ctx, int32_t files_in_flight);
/* Output overlap (BASELINE configs[4]; reconstruction.rs:155-205 runs extract_depth -> DepthMap::new -> output_image per
   image, one after the other): with on = 1 every output back-end call of this section that reads a DEVICE depth buffer
   runs on a second stream of the context, ordered behind the me_extract_depth[_u8] call that WROTE that buffer (known by
   its address) instead of behind everything queued since.  A caller that alternates two device depth buffers can
   therefore queue image i + 1's me_extract_depth first and then make image i's output calls: the GPU works on the next
   depth map while the host waits for this image's mesh counts, OBJ text and its copy to the host.  The next
   me_extract_depth that writes a buffer waits (on the GPU) for the output calls still reading it.  me_ctx_synchronize
   and me_output_flush wait for both streams.  Results are unchanged.  0 (the default): one stream, the reference's order. */
int32_t me_ctx_set_output_overlap(me_ctx* ctx, int32_t on);
int32_t me_output_flush(me_ctx* ctx);

/* The OBJ text of me_output_mesh without the file: the mesh is indexed and every "vt" / "v" / "f" line formatted on
   the GPU (output.rs:484-630 ObjWriter; numbers as Rust's `{}` prints an f64), the lines packed in the reference's
   order behind the "mtllib <stem>.mtl" / "usemtl Textured" header of texture mode.  *text_dev: DEVICE address of the
   bytes, owned by the context and valid until its next mesh call; *nbytes: their count.  me_output_mesh(".obj") is
   this text copied to the host once and written to the file. */
int32_t me_mesh_obj_text(me_ctx* ctx, const float* depth, int32_t width, int32_t height, uint32_t original_width,
                         uint32_t original_height, const char* stem, int32_t vertex_mode,
                         const uint8_t* vertex_colors, const uint8_t** text_dev, int64_t* nbytes);

/* Where the last me_output_mesh(".obj") call on this context spent its time, host wall clock in milliseconds:
   ms_out[0] mesh indexing + vertex kernels (incl. their read-back of the counts), [1] the text formatting kernels,
   [2] the D2H copy of the text into pinned memory, [3] the file write (the host kernel's page-cache copy);
   *text_bytes (optional): the size of the file. */
int32_t me_last_mesh_timing(const me_ctx* ctx, double ms_out[4], int64_t* text_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MATRIX_EYES_HIP_H */
