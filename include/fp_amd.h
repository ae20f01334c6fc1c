/*
 * fp_amd.h -- C ABI of the MI355X-native render-and-compare hot path (libfp_amd.so).
 *
 * The reference (NVlabs/FoundationPose) has no FFI boundary for this path: it is plain
 * Python calling nvdiffrast / kornia / warp / torch.  This header is
 * therefore the boundary *underneath* the preserved Python API; every entry point cites the
 * reference lines it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / STL types.
 *   - return 0 on success, a negative fp_status otherwise; fp_last_error() gives the message
 *     (thread-local).
 *   - pointers marked [dev] are device (HBM) pointers owned by the caller; [host] are host
 *     pointers read before the call returns.  No entry point allocates, frees or synchronises;
 *     all work is enqueued on `stream` (a hipStream_t), so sequences are hipGraph-capturable.
 *   - images are row-major; poses are row-major 4x4 float32 `ob_in_cam` (OpenCV camera).
 */
#ifndef FP_AMD_H
#define FP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  FP_OK = 0,
  FP_ERR_INVALID_ARG = -1,
  FP_ERR_WORKSPACE = -2,
  FP_ERR_LAUNCH = -3,
  FP_ERR_UNSUPPORTED = -4
} fp_status;

/* flags for fp_render_crops / fp_warp_crops */
#define FP_FLAG_NORMALIZE_XYZ 1 /* cfg['normalize_xyz'] (h5_dataset.py:95-101,151-156) */
#define FP_FLAG_OUT_F16 2       /* write the network tensor (A / B) as fp16 instead of fp32 */

/* fp_warp_crops mode */
#define FP_MODE_REFINE 0 /* predict_pose_refine.py:63,72 + PairH5Dataset.transform_batch */
#define FP_MODE_SCORE 1  /* predict_score.py:89-90 + TripletH5Dataset.transform_depth_to_xyzmap */

/* fp_pose_update rot_rep / trans_rep */
#define FP_ROT_AXIS_ANGLE 0
#define FP_ROT_6D 1
#define FP_TRANS_TRACKNET 0 /* cfg['trans_rep'] = 'tracknet' (the released configuration), predict_pose_refine.py:195-199 */
#define FP_TRANS_DEEPIM 1   /* 'deepim': crop-space shift of the projected centre + depth ratio, predict_pose_refine.py:201-215 */
#define FP_TRANS_RAW 2      /* any other trans_rep: the plain `else` branch (:217-218), the raw output (x diameter/2 if normalize_xyz) */

/* integer z-buffer definition; shared with oracle/fp_oracle.c */
#define FP_SUBPIXEL_BITS 4
#define FP_ZBUF_STEPS_PER_METRE 1048576 /* 2^20 */
#define FP_ZBUF_EMPTY 0xFFFFFFFFu

typedef struct fp_mesh fp_mesh; /* opaque: device pointers + sizes of one object's mesh tensors */

const char* fp_last_error(void);
/* ABI version of the library = FP_AMD_ABI_VERSION of the header it was built from; a binding compares the two at load time
 * (foundationpose_amd/_lib.py does and refuses a mismatch).  History of breaks that kept a symbol's name:
 *   200 -> 210 (round 4 / 5): fp_linear_layernorm_fwd takes the FRAGMENT-PACKED weight (fp_pack_linear512_f16) and requires K = 512;
 *                             a caller that still passes the nn.Linear-layout weight gets FP_OK and garbage -- check the version.
 *   210 -> 211 (round 5): + fp_igemm_f16_splitk_fwd / fp_igemm_splitk_workspace_bytes (additions only).
 *   211 -> 212 (round 5): fp_igemm_epilogue grew by one member at its end (w_tiles); + fp_pack_conv3x3_tiles_f16.
 *   212 -> 213 (round 6): w_tiles is read only when flags has FP_IGEMM_HAS_W_TILES (a 212 caller that sets w_tiles without the bit
 *                         gets the plain weight path: correct, slower); fp_igemm_f16_splitk_fwd refuses w_tiles instead of ignoring it;
 *                         fp_linear_layernorm_fwd takes the row stride of x16 (new argument before the stream);
 *                         + fp_encoder_tail_mean_fwd / fp_encoder_tail_workspace_bytes. */
#define FP_AMD_ABI_VERSION 213
int fp_version(void);

/* Utils.py:104-130 make_mesh_tensors: records caller-owned device tensors.
 * pos/nrm (V,3) f32, faces (T,3) i32; either {tex (Ht,Wt,3) f32 in [0,1], uv (V',2) f32 with v already
 * flipped (Utils.py:117), uv_idx (T,3) i32 or NULL => faces} or {vcol (V,3) f32 in [0,1]}. */
int fp_mesh_create(const float* pos /*dev*/, const float* nrm /*dev*/, const int32_t* faces /*dev*/,
                   const float* uv /*dev|NULL*/, const int32_t* uv_idx /*dev|NULL*/,
                   const float* tex /*dev|NULL*/, const float* vcol /*dev|NULL*/, int V, int T, int Ht,
                   int Wt, fp_mesh** out);
void fp_mesh_destroy(fp_mesh* mesh);

/* Utils.py:359-395 erode_depth (+ kernel) */
int fp_depth_erode(const float* depth /*dev H,W*/, float* out /*dev*/, int H, int W, int radius,
                   float depth_diff_thres, float ratio_thres, float zfar, void* stream);
/* Utils.py:304-356 bilateral_filter_depth (+ kernel) */
int fp_depth_bilateral(const float* depth /*dev*/, float* out /*dev*/, int H, int W, int radius,
                       float zfar, float sigmaD, float sigmaR, void* stream);
/* Utils.py:399-417 depth2xyzmap (f64_internal=1, numpy promotion) / :420-438 depth2xyzmap_batch (0) */
int fp_depth_to_xyz(const float* depth /*dev H,W*/, const double* K /*host 9*/, float zfar,
                    int f64_internal, float* xyz /*dev H,W,3*/, int H, int W, void* stream);

/* Utils.py:577-621 compute_crop_window_tf_batch(method='box_3d') and the bbox of
 * predict_pose_refine.py:44-45 / predict_score.py:74-75 (closed-form inverse). */
int fp_crop_windows(const float* poses /*dev N,16*/, const double* K /*host 9*/, double mesh_diameter,
                    double crop_ratio, int out_w, int out_h, int N, float* tf_to_crops /*dev N,9*/,
                    float* bbox2d /*dev N,4*/, void* stream);

/* bytes of scratch fp_render_crops needs for (N hypotheses, V vertices, T triangles, oh x ow crops): per-hypothesis
 * vertex records (32 B/vertex) and per-strip triangle lists; caller-owned device memory, no alignment beyond 256 B */
size_t fp_workspace_bytes(int N, int V, int T, int oh, int ow);

/* Utils.py:133-219 nvdiffrast_render (dr.rasterize + interpolate x5 + texture + Lambert shading + flips)
 * fused with predict_pose_refine.py:54-56 (*255), h5_dataset.py:79-114 (/255, xyz - t, 1/radius, masks)
 * and the channel concat of predict_pose_refine.py:187 (A = [rgb, xyz]).
 * bbox2d NULL => full frame (needs oh=H, ow=W).  Any output may be NULL.
 *   A      (N,6,oh,ow) f32|f16 ; color (N,oh,ow,3) ; depth (N,oh,ow) ; xyz (N,oh,ow,3) ; normal (N,oh,ow,3)
 *   zbuf   (N,oh,ow) u32 fixed-point camera depth of the winner, FP_ZBUF_EMPTY if none ; tri_id i32, -1 if none */
int fp_render_crops(const fp_mesh* mesh, const float* poses /*dev N,16*/, const float* bbox2d /*dev N,4|NULL*/,
                    const float* K9 /*host 9 f32*/, int H, int W, int N, int oh, int ow, float w_ambient,
                    float w_diffuse, float mesh_diameter, float xyz_thr, int flags, void* A /*dev*/,
                    float* color /*dev*/, float* depth /*dev*/, float* xyz /*dev*/, float* normal /*dev*/,
                    uint32_t* zbuf /*dev*/, int32_t* tri_id /*dev*/, void* workspace /*dev*/,
                    size_t workspace_bytes, void* stream);

/* kornia warp_perspective call sites predict_pose_refine.py:63,72 / predict_score.py:89,90 fused with
 * h5_dataset.py:79-114 (refine) or :137-170 (score: depth crop -> frame -> back-projection -> crop) and
 * the concat of predict_pose_refine.py:188 (B = [rgb, xyz]).
 * rgb (H,W,3) f32 in 0..255; xyz_map (H,W,3) f32 (REFINE); depth (H,W) f32 (SCORE). */
int fp_warp_crops(const float* rgb /*dev*/, const float* xyz_map /*dev|NULL*/, const float* depth /*dev|NULL*/,
                  const float* tf_to_crops /*dev N,9*/, const float* K9 /*host 9 f32*/,
                  const float* poses /*dev N,16*/, float mesh_diameter, int flags, int mode, int H, int W,
                  int N, int oh, int ow, void* B /*dev N,6,oh,ow*/, void* stream);

/* predict_pose_refine.py:195-234 + Utils.py:848-855 + pytorch3d so3_exp_map / rotation_6d_to_matrix.
 * trans_delta_out / rot_delta_out (optional): the metric translation delta and the applied rotation matrix
 * (so3_exp_map(.)^T), i.e. what the reference keeps in last_trans_update / last_rot_update (:238-239). */
int fp_pose_update(const float* trans /*dev N,3*/, const float* rot /*dev N,3|6*/,
                   const float* poses_in /*dev N,16*/, int rot_rep, int normalize_xyz,
                   const float* trans_normalizer /*host 3*/, float rot_normalizer, float mesh_diameter, int N,
                   float* poses_out /*dev N,16*/, float* trans_delta_out /*dev N,3|NULL*/,
                   float* rot_delta_out /*dev N,9|NULL*/, int trans_rep, const float* K9 /*host 9 f32|NULL (deepim)*/,
                   const float* tf_to_crops /*dev N,9|NULL (deepim)*/, float input_w /*crop width (deepim)*/, void* stream);

/* ---- network stage.  Arithmetic policy of every entry point below = the op sequence torch.cuda.amp.autocast(fp16)
 * produces for the reference's modules (predict_pose_refine.py:190-191, predict_score.py:193-194): fp16 operands,
 * fp32 accumulation, and a rounding to fp16 wherever the reference holds an fp16 tensor. ---- */

/* refine_network.py:38 / score_network.py:37 first ConvBNReLU (7x7, stride 2, pad 3, 6 -> 64): the "patch-embed conv"
 * as an MFMA implicit GEMM.  x (B,6,Hin,Win) f16 NCHW; w (64, 6*7*7) f16 row-major (PyTorch conv weight flattened);
 * bias (64) f32 holding fp16-representable values | NULL; bn_scale/bn_shift (64) f32 | NULL: eval BatchNorm2d as
 * x*scale + shift.  y = relu(f16(f16(f16(conv) + bias) * scale + shift)), NHWC inside a (B, Hin/2 + 2 pad, Win/2 + 2 pad, 64)
 * f16 buffer whose `pad`-pixel border (pad = 0 | 1) the kernel does not touch (pad 1 = the input layout of fp_igemm_f16_fwd). */
int fp_conv7x7s2_bn_relu_fwd(const void* x /*dev*/, const void* w /*dev*/, const float* bias /*dev|NULL*/,
                             const float* bn_scale /*dev|NULL*/, const float* bn_shift /*dev|NULL*/, void* y /*dev*/,
                             int B, int Hin, int Win, int pad, void* stream);

/* Addressing of one operand of fp_igemm_f16_fwd: GEMM row m = (image b, oy, ox) with b = m / pixels_per_image,
 * oy = (m % pixels_per_image) / width, ox = ... % width, lives at element offset
 *   (((b' * padded_h + oy*stride + offset) * padded_w + ox*stride + offset) * cstride + coff + cg * cgroup
 * of an NHWC fp16 buffer with a zero border, where (b', cg) = (b % bsplit, b / bsplit) if bsplit > 0 else (b, 0)
 * (bsplit writes the A- and B-image features of a pair side by side along C: the channel concat of
 * refine_network.py:82-85 / score_network.py:66-69 without a copy).  A plain matrix is {1,1,1,1,1,0,ld,0,0,0}. */
typedef struct {
  int pixels_per_image, width, padded_h, padded_w, stride, offset, cstride, coff, bsplit, cgroup;
} fp_igemm_geom;

#define FP_IGEMM_RELU 1      /* ReLU after the residual add */
#define FP_IGEMM_HAS_W_TILES 4 /* the struct has the trailing member w_tiles and the library may read it (ABI 213: a caller compiled
                                 against the ABI-211 header passes a shorter struct, never sets the bit, and is never read past its end) */
#define FP_IGEMM_ROUND_ACC 2 /* nn.Conv2d semantics: the accumulator is rounded to fp16 BEFORE the bias is added (ATen adds the
                                bias to the fp16 convolution output) and BatchNorm, if given, rounds once more; without the
                                flag: nn.Linear semantics, one rounding of accumulator + bias */

/* What fp_igemm_f16_fwd does with the fp32 accumulators (all members optional; NULL struct = plain fp16 store) */
typedef struct {
  const float* bias;           /* dev (N) f32 | NULL; for conv semantics fp16-representable values */
  const float* bn_scale;       /* dev (N) f32 | NULL: eval BatchNorm2d as x*scale + shift (needs FP_IGEMM_ROUND_ACC) */
  const float* bn_shift;
  const void* residual;        /* dev fp16 | NULL: `out += identity` (network_modules.py:107), rounded to fp16 */
  const fp_igemm_geom* r_geom; /* host: addressing of the residual */
  int flags;                   /* FP_IGEMM_RELU | FP_IGEMM_ROUND_ACC | FP_IGEMM_HAS_W_TILES */
  const float* pe;             /* dev (pe_period, N) f32 | NULL: second output y_pe[m, n] = f16(f32(y[m, n]) + pe[m % pe_period, n]), */
  int pe_period;               /*   the PositionalEmbedding add of network_modules.py:133-137 fused into the last conv of the */
  void* y_pe;                  /*   encoder; y_pe is a plain (M, N) fp16 matrix */
  const void* w_tiles;         /* dev | NULL (since ABI 212; read only with FP_IGEMM_HAS_W_TILES): the SAME weights once more, in the tile-packed layout of fp_pack_conv3x3_tiles_f16; */
                               /*   the shifted-window 3x3 kernel then fetches a k-step's 128 x 32 weight tile as one contiguous 8 KiB run */
} fp_igemm_epilogue;

/* network_modules.py:37-50 ConvBNReLU / :73-111 ResnetBasicBlock (3x3, pad 1, stride 1|2, eval BatchNorm) and the
 * 512-wide Linear layers of refine_network.py:56-70 / score_network.py:52-53, as ONE MFMA implicit GEMM:
 *   acc[m, n] = sum_{tap, ci} x[row(m) + tap][ci] * w[n][tap*Cin + ci]                       (fp32 accumulation)
 *   y = act( f16(epilogue(acc)) (+ residual[m, n], rounded to f16) ),  epilogue per fp_igemm_epilogue.
 * x / y / residual: NHWC fp16 addressed by their fp_igemm_geom (the input's border must be zero; its geometry
 * addresses tap (0,0), i.e. offset = 0 for pad 1); w (N, taps*Cin) fp16 with k ordered (ky, kx, ci).
 * taps = 9 (3x3) or 1 (GEMM); N % 128 == 0; Cin % 64 == 0.  Stride-1 3x3 convolutions over one padded grid run the
 * shifted-window kernel (csrc/conv_sw.hip), everything else the generic implicit GEMM. */
int fp_igemm_f16_fwd(const void* x /*dev*/, const fp_igemm_geom* x_geom /*host*/, const void* w /*dev*/, void* y /*dev*/,
                     const fp_igemm_geom* y_geom /*host*/, int M, int N, int Cin, int taps,
                     const fp_igemm_epilogue* epilogue /*host|NULL*/, void* stream);

/* One-time repack of a 3x3 convolution weight (N, 9*Cin) fp16, k ordered (ky, kx, ci), for fp_igemm_epilogue.w_tiles: per block of 128
 * output channels and per k-step s = (32-channel chunk cc, tap) in the order the shifted-window kernel consumes them (s = 9 cc + tap), the
 * 128 x 32 weight tile as the 8 KiB LDS image of that kernel (rows of 64 B, 16-byte chunk c of row r at position c ^ ((r >> 2) & 3)):
 *   w_tiles[((bn * 9 * Cin / 32 + s) * 128 + r) * 32 + 8 * pc + e] = w[128 bn + r][tap * Cin + 32 cc + 8 * (pc ^ ((r >> 2) & 3)) + e].
 * N % 128 == 0, Cin % 32 == 0; w_tiles has the size of w and must not alias it.  No reference counterpart (a layout, not an operation). */
int fp_pack_conv3x3_tiles_f16(const void* w /*dev*/, void* w_tiles /*dev*/, int N, int Cin, void* stream);

/* fp_igemm_f16_fwd for launches of a few dozen tiles -- the reference's tracking call (estimater.py:250-268: ONE hypothesis, so the
 * 512 -> 512 convolutions are 400 x 512 x 4608 products = 16 tiles on 256 CUs, each running its whole k loop): the k range is cut into
 * `splits` contiguous pieces of whole 64-wide k-steps, workgroup (tile, piece) leaves fp32 partial accumulators in `workspace`, and a
 * second launch adds the pieces IN ORDER and applies the epilogue (same operations per element as fp_igemm_f16_fwd).  Deterministic;
 * another fp32 summation order than fp_igemm_f16_fwd's, so equal to it up to summation order, not bit for bit -- a caller that
 * needs the bits of a larger batch (shards, sub-batches) must not mix the two for one layer.  Arguments as fp_igemm_f16_fwd;
 * 1 <= splits <= taps * Cin / 64; workspace: fp_igemm_splitk_workspace_bytes(M, N, splits) bytes of device scratch, 16-byte aligned. */
size_t fp_igemm_splitk_workspace_bytes(int M, int N, int splits);
int fp_igemm_f16_splitk_fwd(const void* x /*dev*/, const fp_igemm_geom* x_geom /*host*/, const void* w /*dev*/, void* y /*dev*/,
                            const fp_igemm_geom* y_geom /*host*/, int M, int N, int Cin, int taps,
                            const fp_igemm_epilogue* epilogue /*host|NULL*/, int splits, void* workspace /*dev*/,
                            size_t workspace_bytes, void* stream);

/* network_modules.py:133-137 PositionalEmbedding as the in_proj operand: out = f16(f32(tok) + pe[row % S]).
 * tok / out (M, D) fp16, pe (S, D) f32; D must be 512.  (The fp32 sum itself is never stored: fp_layernorm_res_fwd
 * recomputes it.) */
int fp_add_pe_f16_fwd(const void* tok /*dev*/, const float* pe /*dev*/, void* out /*dev*/, int M, int S, int D, void* stream);

/* refine_network.py:82-85 `ab = torch.cat((a, b), 1)` when every b is ONE image -- the first refine iteration of
 * estimater.py:214-215 register(), whose hypotheses share a translation (estimater.py:132-133) and therefore the observed
 * crop: dst[c][r][0..channels) = src[r][0..channels) for c < copies.  fp16 rows at row strides (in fp16 values; channels and
 * all strides multiples of 8, pointers 16-byte aligned), copy c at dst + c * dst_copy_stride.  src and dst may be
 * different images of one buffer as long as the source rows are not among the destination rows. */
int fp_replicate_rows_f16(const void* src /*dev*/, void* dst /*dev*/, int copies, int rows, int channels, int src_row_stride,
                          int dst_row_stride, long long dst_copy_stride, void* stream);

/* The LayerNorms of nn.TransformerEncoderLayer (refine_network.py:56-70; post-norm, eps 1e-5) on the fp32 residual
 * stream autocast keeps:  z = resid + f32(branch16);  y = LN(z)*gamma + beta  -> y32 (M,D) f32 and/or y16 (M,D) f16,
 * with resid = x32 (M,D) f32, or f32(tok16) + pe[row % S] when x32 is NULL.  D must be 512. */
int fp_layernorm_res_fwd(const float* x32 /*dev|NULL*/, const void* tok16 /*dev|NULL*/, const float* pe /*dev|NULL*/, int S,
                         const void* branch16 /*dev*/, const float* gamma /*dev D*/, const float* beta /*dev D*/, float eps,
                         float* y32 /*dev|NULL*/, void* y16 /*dev|NULL*/, int M, int D, void* stream);

/* Fragment-packed copy of a (512, 512) fp16 weight matrix W[out][in] (nn.Linear layout) for fp_linear_layernorm_fwd and
 * fp_ffn_layernorm_mean_fwd, whose waves read their weight rows straight from L2 into MFMA operand registers: for channel group
 * w = out / 64, k16-step q = in / 16, channel tile i = (out / 32) % 2 the 64 lanes' operands stand back to back,
 *   packed[((w * 32 + q) * 2 + i) * 64 + lane][0..7] = W[64 w + 32 i + (lane & 31)][16 q + 8 (lane >> 5) + 0..7],
 * so a wave load is one contiguous KiB (the row-per-lane form of the same load is served at one lane per clock by the vector
 * cache).  Same size as the matrix; done once per weight matrix; `packed` must not alias `w16`. */
int fp_pack_linear512_f16(const void* w16 /*dev*/, void* packed /*dev*/, void* stream);

/* y16 (M, N) = f16(x16 (M, 512) @ W^T + bias) [ReLU if relu != 0]: nn.Linear under autocast (fp16 operands, fp32 accumulation +
 * bias, one rounding) for in_features = 512 and N = 512 n <= 3072 output features -- the in_proj of nn.MultiheadAttention (N = 1536;
 * refine_network.py:56-70 through nn.TransformerEncoderLayer, score_network.py:52-53,73,86).  The bits of fp_igemm_f16_fwd with
 * taps = 1 (same k order per accumulator); a workgroup fetches its 128 x 512 input tile once and keeps it in LDS for all N / 512
 * column blocks, weights from L2 into registers.  w_packed: the N / 512 blocks of 512 output channels of W (N, 512), each through
 * fp_pack_linear512_f16, back to back; bias (N) f32 | NULL. */
int fp_linear512_f16_fwd(const void* x16 /*dev*/, const void* w_packed /*dev*/, const float* bias /*dev|NULL*/, void* y16 /*dev*/,
                         int M, int N, int relu, void* stream);

/* A 512 -> 512 nn.Linear of the encoder layer (refine_network.py:56-70: self_attn.out_proj or linear2, under autocast: fp16
 * operands, fp32 accumulation + bias, one rounding to fp16) fused with the residual add and the LayerNorm that consume it:
 * fp_igemm_f16_fwd (taps = 1, N = 512) followed by fp_layernorm_res_fwd with branch16 = that product, in one launch and
 * without the (M, 512) product reaching HBM; per element the same instruction sequence, i.e. the same bits.
 * x16 (M, 512) fp16, w16_packed = fp_pack_linear512_f16 of the (512, 512) weight, bias (512) f32 | NULL; K and D must be 512 (the
 * 128 x K A tile of a workgroup lives in LDS whole); the other arguments as fp_layernorm_res_fwd.  ldx (ABI 213): x16 may be a column
 * block of a wider matrix -- one head's (M, 512) half of the (M, 1024) output of a two-head fp_attention_f16_fwd call (round 6: the
 * self-attention of RefineNet's trans_head and rot_head, refine_network.py:56-70, as one 8-head launch). */
int fp_linear_layernorm_fwd(const void* x16 /*dev*/, const void* w16_packed /*dev*/, const float* bias /*dev|NULL*/,
                            const float* x32 /*dev|NULL*/, const void* tok16 /*dev|NULL*/, const float* pe /*dev|NULL*/, int S,
                            const float* gamma /*dev D*/, const float* beta /*dev D*/, float eps, float* y32 /*dev|NULL*/,
                            void* y16 /*dev|NULL*/, int M, int K, int D, int ldx /* row stride of x16 in fp16 values; 0 = K */,
                            void* stream);

/* Everything of nn.TransformerEncoderLayer behind the attention context (refine_network.py:56-70: self_attn.out_proj, `x + sa`, norm1,
 * linear1 -> ReLU -> linear2, `x + ff`, norm2) + the token mean of RefineNet.forward (refine_network.py:90-91), in ONE launch (+ the
 * finish kernel) = fp_linear_layernorm_fwd followed by fp_ffn_layernorm_mean_fwd, with norm1's fp16 output staying in LDS between them
 * (round 6; the same bits as those two calls: per element the same instruction sequences).  ctx16: (M, 512) fp16 with row stride ldx
 * (0 = 512), M = groups * rows_per_group; tok16 (M, 512) fp16 + pe (rows_per_group, 512) f32 = the layer input f32(tok16) + pe[row %
 * rows_per_group]; *_packed: fp_pack_linear512_f16 of the three (512, 512) weights; out (groups, 512) f32.  workspace:
 * fp_encoder_tail_workspace_bytes(groups, rows_per_group) bytes of device scratch (norm1's fp32 output -- the residual of norm2 -- and
 * the token-mean chunk sums), 16-byte aligned. */
size_t fp_encoder_tail_workspace_bytes(int groups, int rows_per_group);
int fp_encoder_tail_mean_fwd(const void* ctx16 /*dev*/, int ldx, const void* wo_packed /*dev*/, const float* bo /*dev|NULL*/,
                             const void* tok16 /*dev*/, const float* pe /*dev*/, const float* gamma1 /*dev*/, const float* beta1 /*dev*/,
                             const void* w1_packed /*dev*/, const float* b1 /*dev|NULL*/, const void* w2_packed /*dev*/,
                             const float* b2 /*dev|NULL*/, const float* gamma2 /*dev*/, const float* beta2 /*dev*/, float eps,
                             float* out /*dev*/, void* workspace /*dev*/, size_t workspace_bytes, int groups, int rows_per_group,
                             void* stream);

/* The feed-forward half of nn.TransformerEncoderLayer (refine_network.py:56-70: linear1 -> ReLU -> linear2, `x + ff`, norm2; under
 * autocast: fp16 Linears with fp32 accumulation + bias and one rounding each, fp32 residual stream and LayerNorm) fused with the
 * token mean that follows it in RefineNet.forward (refine_network.py:90-91, taken before the 512 -> 3|6 head: the mean and that
 * Linear commute):  out[g, :] = mean_{r < rows_per_group} ( LN(x32[row] + f32(linear2(relu(linear1(y16[row]))))) * gamma + beta ),
 * row = g * rows_per_group + r.  = fp_igemm_f16_fwd x 2 + fp_colmean_f16_fwd with neither (M, 512) intermediate reaching HBM:
 * a workgroup owns 128 complete rows through both Linears and the LayerNorm; the token mean is summed in chunks of 16
 * consecutive rows of a group, then over a group's chunks in order -- deterministic, and independent of where in the batch a
 * group sits (a sub-batch or shard returns the bits of the full batch); another fp32 summation order than fp_colmean_f16_fwd's, so
 * equal to it up to fp32 rounding, not bit for bit.  y16 (M, 512) fp16 = norm1's output, x32 (M, 512) f32 = the residual stream,
 * w1_packed / w2_packed = fp_pack_linear512_f16 of the (512, 512) fp16 weights, b1 / b2 (512) f32 | NULL, out (groups, 512) f32,
 * M = groups * rows_per_group, rows_per_group a multiple of 16.
 * workspace: M / 16 * 512 floats of device scratch owned by the caller. */
int fp_ffn_layernorm_mean_fwd(const void* y16 /*dev*/, const void* w1_packed /*dev*/, const float* b1 /*dev|NULL*/,
                              const void* w2_packed /*dev*/, const float* b2 /*dev|NULL*/, const float* x32 /*dev*/,
                              const float* gamma /*dev 512*/, const float* beta /*dev 512*/, float eps, float* out /*dev*/,
                              float* workspace /*dev*/, size_t workspace_bytes, int groups, int rows_per_group, void* stream);

/* `.mean(dim=1)` over the tokens of each hypothesis (refine_network.py:90-91, score_network.py:74), optionally fused
 * with the residual add + LayerNorm that precede it: out[g, :] = mean_{r < rows_per_group} f(row g*rows_per_group + r),
 * f = LN(resid32 + f32(x))*gamma + beta if gamma != NULL (resid32 may be NULL) else f32(x).  x (groups*rows_per_group, D)
 * fp16, resid32 same shape f32, out (groups, D) f32. */
int fp_colmean_f16_fwd(const void* x /*dev*/, const float* resid32 /*dev|NULL*/, const float* gamma /*dev D|NULL*/,
                       const float* beta /*dev D|NULL*/, float eps, float* out /*dev*/, int groups, int rows_per_group,
                       int D, void* stream);

#define FP_ROWS_ROUND_F16 1 /* y (f32) holds fp16-representable values: the reference keeps these outputs in fp16 */
#define FP_ROWS_X_F16 2     /* x is fp16 instead of f32 */
#define FP_ROWS_Y_F16 4     /* y is stored as fp16 instead of f32 */

/* The Linear layers that follow a token mean (refine_network.py:58,69 heads; score_network.py:53 out_proj after the
 * mean has been commuted in front of it; score_network.py:57 final Linear): y[M,N] = x[M,K] @ w[N,K]^T + bias for a
 * few hundred rows, fp32 accumulation.  x f32|f16, w f16, bias f32|NULL, y f32|f16.  K % 8 == 0, K <= 2048. */
int fp_rows_linear_fwd(const void* x /*dev*/, const void* w /*dev*/, const float* bias /*dev|NULL*/, void* y /*dev*/,
                       int M, int K, int N, int flags, void* stream);

#define FP_ATT_FP16_SCORES 1 /* the need_weights=True branch of F.multi_head_attention_forward under autocast
                                (score_network.py:73,86): q * sqrt(1/d) rounded to fp16 and the q.k scores rounded to fp16
                                before the fp32 softmax; default: F.scaled_dot_product_attention (fp32 scores) */

/* The attention core of nn.MultiheadAttention(512, 4) as the networks call it (query = key = value, no mask, eval):
 * refine_network.py:56-70 (nn.TransformerEncoderLayer self-attention over the 400 tokens of a hypothesis),
 * score_network.py:52-53 (token attention) and :84-88 (attention across the hypotheses), i.e. what
 * torch.nn.functional.multi_head_attention_forward does between in_proj and out_proj:
 *   out[b, t, h*hd:(h+1)*hd] = softmax_t'(q[b,t,h,:] . k[b,t',h,:] / sqrt(hd)) v[b,t',h,:]
 * qkv (B*S, 3*H*hd) fp16 = the in_proj output [q | k | v]; out (B*S, H*hd) fp16 = the out_proj input.  hd must be 128.
 * fp32 softmax statistics and accumulation, probabilities rounded to fp16 for the second product, normalisation
 * after it (flash-attention order); the (B*H, S, S) probability tensor is never formed. */
int fp_attention_f16_fwd(const void* qkv /*dev*/, void* out /*dev*/, int B, int S, int H, int head_dim, int flags,
                         void* stream);

/* mycpp/src/app/pybind_api.cpp:24-68 cluster_poses (host, init-time). Returns #kept, indices in keep_idx. */
int fp_cluster_poses(float angle_diff_deg, float dist_diff, const float* poses /*host N,16*/, int N,
                     const float* symmetry_tfs /*host S,16*/, int S, int* keep_idx /*host N*/);

#ifdef __cplusplus
}
#endif
#endif /* FP_AMD_H */
