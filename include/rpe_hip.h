/*
 * rpe_hip.h -- C ABI of librpe_hip.so: the MI355X (gfx950) implementation of the
 * RGB + proprioception pose-regression TRAIN STEP of
 * cremebrule/rgb-proprioceptive-pose-estimator.
 *
 * The reference has no FFI / plugin layer: the path sits behind Python classes
 * (models/naive.py, models/time_sensitive.py, models/losses.py, util/learn_utils.py).
 * Every entry point below therefore replaces the torch op(s) a reference line
 * dispatches; the citation after "replaces:" is the reference file:line.  The Python
 * host side (rgb-proprioceptive-pose-estimator_amd/) binds these with ctypes and
 * keeps the reference's class / constructor / state_dict surface.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless its name ends in _host.  Buffers are
 *    borrowed: nothing is freed or retained past the call except by the engine
 *    object, which retains exactly what rpe_resnet50_bind() hands it.
 *  - `stream` is a hipStream_t passed as void*.  Calls only enqueue work; no entry
 *    point synchronises, allocates or frees device memory (hipGraph-capturable).
 *  - Activations of the conv trunk are NHWC ([B][H][W][C], C contiguous) in the
 *    compute dtype (RPE_F32, RPE_BF16 or RPE_F16; accumulation is always fp32).  Head tensors
 *    (features, MLP / LSTM, loss) are fp32 with a leading dimension `ld`.
 *  - Operands must be 16-byte aligned and channel counts / leading dimensions
 *    multiples of the 16-byte chunk (4 fp32 / 8 bf16) unless stated otherwise.
 *  - Return value: 0 on success, an RPE_ERR_* code otherwise; rpe_last_error()
 *    returns a thread-local message.  No exceptions cross the boundary.
 */
#ifndef RPE_HIP_H
#define RPE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define RPE_ABI_VERSION 3   /* 2: BN partial-sum row counts depend on the element type (halo form of the 3x3 convs); 3: the staged image x4 is zero-bordered */

enum { RPE_F32 = 0, RPE_BF16 = 1, RPE_F16 = 2 };   /* RPE_F16: IEEE half activations / weight copies (BASELINE config C5), fp32 accumulate */
enum {
    RPE_OK = 0,
    RPE_ERR_SHAPE = 1,     /* bad dimension / unsupported configuration */
    RPE_ERR_DTYPE = 2,
    RPE_ERR_ALIGN = 3,
    RPE_ERR_HIP = 4,       /* a HIP runtime call failed */
    RPE_ERR_WORKSPACE = 5, /* caller-provided workspace too small */
    RPE_ERR_STATE = 6      /* engine used before bind / forward */
};

int rpe_abi_version(void);
/* the first 16 hex digits of the SHA-256 of the sources this library was built from (fixed at build time; csrc/Makefile): ties
 * measurements kept beside the repository (profiles/hbm_traffic_pmc.json, "build_id") to the library a process actually loaded */
const char* rpe_build_id(void);
const char* rpe_last_error(void);

/* ------------------------------------------------------------------ convolution */
/* replaces: nn.Conv2d(bias=False) forward/backward inside torchvision resnet50
 * (constructed at util/model_utils.py:136, called at models/naive.py:84,316 and
 * models/time_sensitive.py:185,472,733). */
typedef struct {
    int batch, in_h, in_w, in_c; /* NHWC input */
    int out_c, kh, kw;
    int stride, pad;             /* stride 1 or 2 */
} rpe_conv_desc;

int rpe_conv_out_hw(const rpe_conv_desc* d, int* ho, int* wo);
/* number of 128-row tiles = rows of the BN partial-sum buffer [tiles][2][out_c] of the 1x1 / stem / strided launches */
long rpe_conv_stats_tiles(long rows);
/* rows of stats_part [tiles][2][out_c] rpe_conv2d_fwd writes for this conv and element type: 128-row tiles, except the 3x3 / stride 1 /
 * pad 1 convs of the 16-bit types, whose tiles are whole output rows (the halo form) */
long rpe_conv2d_fwd_stats_tiles(const rpe_conv_desc* d, int dtype);

/* y[B][Ho][Wo][out_c] = conv(x, w);  w_krsc = [out_c][kh][kw][in_c].
 * stats_part (nullable): per-128-row-tile column sums and sums of squares of the fp32
 * accumulators, consumed by rpe_bn_finalize. */
int rpe_conv2d_fwd(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* y, float* stats_part, void* stream);
/* inference form: out = [relu](conv(x, w) + bias[out_c] (+ addend[rows][out_c])) in one launch.  With w = weight * BN scale
 * (rpe_pack_desc.scale) and bias = BN shift this is conv -> BatchNorm2d(eval) -> (+identity) -> ReLU of torchvision's
 * Bottleneck (called at models/naive.py:316 in eval mode / rollout). */
int rpe_conv2d_fwd_affine(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* out, const float* bias, const void* addend,
                          int relu, void* stream);
/* The same with a caller-provided workspace: launches with few output tiles and a long reduction (one rollout frame through
 * layer2-4: 4..26 tiles of 64 x 64 walking K = 1152..4608 alone) are split along K over 4..32 workgroups per tile; the
 * partial fp32 tiles go through the workspace and are added in a fixed order by a second small launch.  The query returns 0
 * for shapes that are not split (the call then equals rpe_conv2d_fwd_affine); workspace 16-byte aligned, no state kept in it. */
long rpe_conv2d_fwd_affine_workspace_bytes(const rpe_conv_desc* d, int dtype);
int rpe_conv2d_fwd_affine_ws(const rpe_conv_desc* d, int dtype, const void* x, const void* w_krsc, void* out, const float* bias, const void* addend,
                             int relu, void* workspace, long workspace_bytes, void* stream);
/* dx[B][H][W][in_c] = conv_transpose(dy, w) (+ addend);  w_crsk = [in_c][kh][kw][out_c]. */
int rpe_conv2d_dgrad(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dx, const void* addend, void* stream);
/* Data gradient with the BatchNorm-backward reduction of the PRODUCING layer fused into the epilogue:
 * dz[B][H][W][in_c] = (conv_transpose(dy, w) + addend) * [ReLU mask of that layer's output], and per-128-row-tile partial
 * sums (sum dz, sum dz*xhat) -> bn->stats_part [tiles][2][in_c], consumed by rpe_bn_backward_from_dz.
 * mask: a_mask != NULL: its bits;  else a_out != NULL: a_out > 0;  else scale/shift != NULL: y*scale+shift > 0 (BN+ReLU
 * without residual);  else none. */
typedef struct {
    const void* y;       /* raw conv output the BN normalised, same shape as dz.  NULL together with a_mask (1x1 / stride-1 convs of a 16-bit
                          * type): that output does not exist -- only sum dz is emitted (the second half of every partial row is 0),
                          * sum dz*xhat then follows from the weight gradient's first product (rpe_bn_backward_coeffs_t) */
    const void* a_out;   /* BN(+residual)+ReLU output, or NULL */
    const float *mean, *invstd, *scale, *shift;
    float* stats_part;
    const unsigned char* a_mask; /* or: the packed ReLU mask rpe_bn_apply_mask wrote, [rows][in_c/8] bytes (bit j = channel 8k+j > 0) */
} rpe_bn_bwd_epilogue;
/* rows of bn->stats_part the fused data gradient writes (stride-2 layers enumerate rows per parity class; 3x3 / stride 1 in 16-bit types: halo tiles) */
long rpe_conv2d_dgrad_stats_tiles(const rpe_conv_desc* d, int dtype);
int rpe_conv2d_dgrad_bn(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dz, const void* addend,
                        const rpe_bn_bwd_epilogue* bn, void* stream);
/* A bottleneck block WITHOUT its raw conv3 output (16-bit element types; replaces conv3 -> bn3 -> (+identity) -> ReLU of torchvision's
 * Bottleneck in training mode -- util/model_utils.py:136 builds it, models/naive.py:316 calls it -- and their backward).
 * BatchNorm statistics of a 1x1 conv follow from its INPUT: y = x W^T gives mean_c = w_c . colsum(x) / M, E[y_c^2] = w_c^T (x^T x) w_c / M.
 * rpe_gram: x [M][C] -> out fp32 [rpe_gram_ones_row(C) + 1][C]: x^T x in rows [0, C), colsum(x) in row rpe_gram_ones_row(C) (one
 * weight-gradient-style launch, fixed-order slab sum).  rpe_bn_stats_from_gram: what rpe_bn_finalize produces (scale, shift, saved
 * mean / invstd, running statistics), from the Gram matrix and the compute-dtype weight.  rpe_conv1x1_fwd_bn: the conv with
 * out = relu(acc * scale + shift + residual [* res_scale + res_shift]) and the packed ReLU mask in its epilogue -- the raw output
 * is written only when y_out is given.  Without y_out, with a residual and a mask, in_c 64 / 128, out_c a multiple of 256 and >= 512 rows
 * (the y3-free blocks of layers 1-2) the launch is the row-streaming kernel of csrc/stream1x1.hip (weights resident in registers, a per-wave
 * LDS-DMA ring), bitwise the tiled form; RPE_NO_STREAM1X1=1 keeps the tiled form. */
/* rpe_bn_apply_gram: out = relu(y * scale + shift) (bn2's apply pass) AND the Gram buffer of `out` -- same layout and values as rpe_gram(out)
 * -- in one pass over y (C = 64 or 128, 16-bit element types; workspace = per-workgroup fp32 partials, summed in a fixed order). */
long rpe_bn_apply_gram_workspace_bytes(int dtype, long rows, int C);
int rpe_bn_apply_gram(int dtype, const void* y, void* out, const float* scale, const float* shift, long rows, int C, float* gram_out,
                      void* workspace, long workspace_bytes, void* stream);
long rpe_gram_ones_row(int C);
long rpe_gram_workspace_bytes(int dtype, long M, int C);
int rpe_gram(int dtype, const void* x, long M, int C, float* out, void* workspace, long workspace_bytes, void* stream);
int rpe_bn_stats_from_gram(int dtype, const void* w, int Co, int Ci, const float* gram, int ones_row, long count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, long long* num_batches, float momentum, float eps, float* scale, float* shift,
                           float* save_mean, float* save_invstd, void* stream);
int rpe_conv1x1_fwd_bn(const rpe_conv_desc* d, int dtype, const void* x, const void* w, void* out, void* y_out, const float* scale, const float* shift,
                       const void* residual, const float* res_scale, const float* res_shift, unsigned char* relu_mask, void* stream);
/* BatchNorm backward folded into the data gradient of the 1x1 / stride-1 conv in front of it (y = a_in W^T, z = BN(y)):
 *   dx = dz (A o W) + a_in G + 1 b^T   with A = gamma invstd, G = W^T diag(C') W, b = W^T B'  (C', B' from the BN coefficients),
 * so the data gradient reads dz and the conv's INPUT instead of a materialised dy = BN'(dz, y).
 * rpe_bn_bwd_fold_conv1x1 builds w_kcat [in_c][out_c + in_c] (compute dtype) and bias [in_c] from the packed weights
 * (w_fwd [out_c][in_c], w_dgrad [in_c][out_c]) and c1c2 (rpe_bn_backward_coeffs); rpe_conv1x1_dgrad_kcat is the GEMM
 * (A operand = [dz | a_in], optional fused BN-backward epilogue of the layer BEHIND, as rpe_conv2d_dgrad_bn).
 * replaces: autograd's BatchNorm2d backward + conv3 data gradient of a torchvision Bottleneck (util/model_utils.py:136). */
long rpe_bn_bwd_fold_scratch_bytes(int dtype, int out_c, int in_c);
int rpe_bn_bwd_fold_conv1x1(int dtype, int out_c, int in_c, const void* w_fwd, const void* w_dgrad, const float* gamma, const float* invstd,
                            const float* mean, const float* c1c2, void* w_kcat, float* bias, void* scratch, long scratch_bytes, void* stream);
int rpe_conv1x1_dgrad_kcat(const rpe_conv_desc* d, int dtype, const void* dz, const void* a_in, const void* w_kcat, const float* bias, void* dx,
                           const rpe_bn_bwd_epilogue* bn, void* stream);
/* The same fold for the conv in FRONT of a narrow BatchNorm -- a Bottleneck's conv1 (x [M][in_c] -> y [M][out_c], z = bn1(y)):
 *   dx = [dz | y] [A o W ; C' o W] + 1 b^T   (y is the second K-concatenated operand itself: K = 2 out_c, no Gram matrix),
 *   dW[k][n] = A_k ((dz^T x)[k][n] - c1_k s1[n]) + C'_k ((y^T x)[k][n] - mean_k s1[n]),  s1 = colsum(x)   (one row-concatenated launch),
 * so neither gradient reads a materialised dy and bn1's streaming dz, y -> dy pass is gone (replaces the BatchNorm2d backward + conv1
 * backward of a torchvision Bottleneck as torch autograd runs them: util/model_utils.py:136, models/naive.py:316).
 * rpe_bn_bwd_fold_y_conv1x1: w_kcat [in_c][2 out_c] (compute dtype), bias [in_c] from the data-gradient copy of the weight [in_c][out_c]
 * and the BN coefficients (rpe_bn_backward_coeffs); rpe_conv1x1_dgrad_kcat_y: the data gradient (+ bias, + addend, + the fused BN-backward
 * epilogue of the layer behind, as rpe_conv2d_dgrad_bn); rpe_conv1x1_wgrad_folded_y: the weight gradient into dw [out_c][in_c] fp32. */
int rpe_bn_bwd_fold_y_conv1x1(int dtype, int out_c, int in_c, const void* w_dgrad, const float* gamma, const float* invstd, const float* mean,
                              const float* c1c2, void* w_kcat, float* bias, void* stream);
int rpe_conv1x1_dgrad_kcat_y(const rpe_conv_desc* d, int dtype, const void* dz, const void* y, const void* w_kcat, const float* bias, void* dx,
                             const void* addend, const rpe_bn_bwd_epilogue* bn, void* stream);
long rpe_conv1x1_wgrad_folded_y_scratch_bytes(const rpe_conv_desc* d, int dtype);
int rpe_conv1x1_wgrad_folded_y(const rpe_conv_desc* d, int dtype, const void* dz, const void* y, const void* x, const float* gamma, const float* invstd,
                               const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, void* stream);
/* The weight-gradient side of the same fold: dw[out_c][in_c] (fp32, OVERWRITTEN) = A o (dz^T a_in) + B' colsum(a_in)^T + C' o (W S),
 * S = a_in^T a_in -- reads dz and a_in only, no dy.  w_master: the fp32 weight [out_c][in_c].  Deterministic (slab sums). */
long rpe_conv1x1_wgrad_folded_scratch_bytes(const rpe_conv_desc* d, int dtype);
int rpe_conv1x1_wgrad_folded(const rpe_conv_desc* d, int dtype, const void* dz, const void* a_in, const float* w_master, const float* gamma,
                             const float* invstd, const float* mean, const float* c1c2, float* dw, void* scratch, long scratch_bytes, void* stream);
/* Backward of a block whose raw conv3 output was never written (rpe_conv1x1_fwd_bn with y_out = NULL).  The fused data gradient that
 * produces dz (rpe_conv2d_dgrad_bn with bn->y = NULL, bn->a_mask set) emits sum dz only; the weight gradient's first product
 * T = dz^T a_in ([out_c][in_c] fp32 = rpe_conv2d_wgrad_det(x = a_in, dy = dz)) then gives sum dz*y = rowdot(T_c, W_c):
 * rpe_bn_backward_coeffs_t = rpe_bn_backward_coeffs with that second sum (w: the compute-dtype weight [C][Ci] the forward used), and
 * rpe_conv1x1_wgrad_combine = the last step of rpe_conv1x1_wgrad_folded from T and the FORWARD's Gram buffer (rpe_gram of a_in):
 * dw = A o (T - c1 s1^T) + C' o (W S - mean s1^T).  Replaces autograd's BatchNorm2d + conv3 backward of a torchvision Bottleneck. */
/* rpe_conv1x1_dgrad_bn_t: the fused data gradient of the NEXT block's conv1 (rpe_conv2d_dgrad_bn with bn->y = NULL: dz of the producing
 * block, its ReLU mask applied, sum dz per tile) that ALSO leaves T = dz^T a_prev ([in_c][P] fp32, a_prev [rows][P] = the producing
 * block's conv3 input, P = 64 or 128) behind -- from the dz tile on its way out, so the separate dz^T a launch and its re-read of dz
 * are gone.  Persistent launch (per-workgroup partials in `workspace`, summed in workgroup order: deterministic). */
long rpe_conv1x1_dgrad_bn_t_workspace_bytes(const rpe_conv_desc* d, int P);
int rpe_conv1x1_dgrad_bn_t(const rpe_conv_desc* d, int dtype, const void* dy, const void* w_crsk, void* dz, const void* addend, const rpe_bn_bwd_epilogue* bn,
                           const void* a_prev, int P, float* t_out, void* workspace, long workspace_bytes, void* stream);
int rpe_bn_backward_coeffs_t(int dtype, const float* stats_part, int tiles, int C, long rows, const float* dzt_a, const void* w, int Ci, const float* mean,
                             const float* invstd, float* dgamma, float* dbeta, float* c1c2, double* dpart, void* stream);
int rpe_conv1x1_wgrad_combine(const rpe_conv_desc* d, const float* dzt_a, const float* gram, const float* w_master, const float* gamma, const float* invstd,
                              const float* mean, const float* c1c2, float* dw, void* stream);
/* the two halves of rpe_bn_backward_from_dz as separate calls (c1c2: [2][C] fp32 = mean(dz), mean(dz xhat)) */
int rpe_bn_backward_coeffs(const float* stats_part, int tiles, int C, long rows, float* dgamma, float* dbeta, float* c1c2, double* dpart, void* stream);
int rpe_bn_backward_apply_dz(int dtype, const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma, const float* c1c2,
                             void* dy, long rows, int C, void* stream);
/* dw_krsc[out_c][kh][kw][in_c] (fp32) += x (*) dy.  Atomic accumulation: zero it first. */
int rpe_conv2d_wgrad(const rpe_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_krsc, void* stream);
/* Deterministic form: dw_krsc = x (*) dy (OVERWRITTEN, no zeroing needed).  Every workgroup stores its fp32 tile into
 * `workspace` (>= rpe_conv2d_wgrad_workspace_bytes(), 16-byte aligned, caller-owned, free again when the call's work has run)
 * and a second launch sums the tiles in a fixed order: bitwise reproducible, no float atomics. */
long rpe_conv2d_wgrad_workspace_bytes(const rpe_conv_desc* d, int dtype);
int rpe_conv2d_wgrad_det(const rpe_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_krsc, void* workspace, long workspace_bytes,
                         void* stream);

/* ResNet stem conv1 (3->64, 7x7 / 2, pad 3) on the staged image x4; w_packed = [64][8][8][4] from rpe_pack_stem_weight.
 * x4 is ZERO-BORDERED NHWC4: [B][H + 2 RPE_STEM_PAD][W + 2 RPE_STEM_PAD][4] in the compute dtype, the image at rows / columns
 * [RPE_STEM_PAD, RPE_STEM_PAD + H) x [.., + W), channel 3 = 0, and ZEROS around it (conv1's padding made explicit: every 16-byte
 * piece of an im2col row is then an in-bounds aligned read, so the conv and its weight gradient stage through LDS-DMA -- round 4; the
 * unpadded layout of rounds 1-3 forced register staging at 1.8-2.1 TB/s).  rpe_stage_image_nhwc4 / rpe_stage_frames_u8[_resized] write
 * the interior only: zero the buffer once before its first use (rpe_x4_bytes() bytes).  H, W even. */
#define RPE_STEM_PAD 3
long rpe_x4_bytes(int dtype, int B, int H, int W);
int rpe_stem_conv_fwd(int dtype, const void* x4, const void* w_packed, void* y, float* stats_part, int B, int H, int W, void* stream);
int rpe_stem_conv_fwd_affine(int dtype, const void* x4, const void* w_packed, void* out, const float* bias, int relu, int B, int H, int W,
                             void* stream);
int rpe_stem_conv_wgrad(int dtype, const void* x4, const void* dy, float* dw_packed, int B, int H, int W, void* stream);
/* deterministic form (see rpe_conv2d_wgrad_det): dw_packed is overwritten */
long rpe_stem_conv_wgrad_workspace_bytes(int dtype, int B, int H, int W);
int rpe_stem_conv_wgrad_det(int dtype, const void* x4, const void* dy, float* dw_packed, int B, int H, int W, void* workspace, long workspace_bytes,
                            void* stream);

/* weight layouts.  w_krsc_f32 is the fp32 master in channels_last storage. */
int rpe_pack_conv_weight(int dtype, const float* w_krsc, void* w_fwd, void* w_dgrad, int Co, int R, int S, int Ci, void* stream);
/* the same for many layers in one launch; the descriptor table lives in device memory */
typedef struct {
    const float* src; /* [Co][RS][Ci] fp32 */
    void* wf;         /* forward copy or NULL */
    void* wd;         /* dgrad copy [Ci][RS][Co] or NULL */
    int Co, RS, Ci;
    int pad_;
    long start;       /* first flat element index of this layer */
    const float* scale; /* NULL, or [Co]: the forward copy is written as w * scale[co] (inference: BatchNorm folded into the conv) */
} rpe_pack_desc;
int rpe_pack_conv_weights_multi(int dtype, const rpe_pack_desc* table_dev, int nlayers, long total, void* stream);
/* scale: NULL or [64] (see rpe_pack_desc.scale) */
int rpe_pack_stem_weight(int dtype, const float* w_oihw, const float* scale, void* out, void* stream);
int rpe_unpack_stem_grad(const float* d_packed, float* dw_oihw, void* stream);

/* replaces: the per-tensor .cuda() staging of util/learn_utils.py:130-138 for `img`
 * ((B,3,H,W) fp32 NCHW, util/data_utils.py:62-73) -> the interior of the zero-bordered NHWC4 image x4 (see rpe_stem_conv_fwd). */
int rpe_stage_image_nhwc4(int dtype, const float* img_nchw, void* out, int B, int H, int W, void* stream);

/* replaces: the CPU image transform ToPILImage -> Resize(256) -> CenterCrop(224) -> ToTensor -> Normalize
 * (util/data_utils.py:48-54) for frames whose shorter side already equals the resize target: uint8 [B][Hs][Ws][3] frames are
 * centre-cropped to (H, W), normalised with the HOST arrays mean3/std3 and written as NHWC4 in the compute dtype. */
int rpe_stage_frames_u8(int dtype, const unsigned char* frames, void* out, int B, int Hs, int Ws, int H, int W, const float* mean3_host,
                        const float* std3_host, void* stream);
/* The same for frames whose shorter side is not the transform's resize target: Pillow's antialiased 8-bit bilinear resample
 * (Resize(256) on a PIL image, util/data_utils.py:48-54) to Hr x Wr in its own 22-bit fixed-point arithmetic, then the crop of
 * H x W at (top, left) and the normalisation.  xb/yb: [Wr]/[Hr] x (first tap, tap count); xk/yk: [Wr][ksx] / [Hr][ksy] int32 weights
 * (device pointers; a pass whose size does not change takes nulls); tmp: B*Hs*Wr*3 bytes of device scratch. */
int rpe_stage_frames_u8_resized(int dtype, const unsigned char* frames, void* out, int B, int Hs, int Ws, int Hr, int Wr, int top, int left, int H, int W,
                                const int* xb, const int* xk, int ksx, const int* yb, const int* yk, int ksy, unsigned char* tmp, const float* mean3_host,
                                const float* std3_host, void* stream);

/* ------------------------------------------------------------------ batch norm */
/* replaces: nn.BatchNorm2d (train mode: biased batch variance, eps, momentum with
 * unbiased running variance) + the in-place nn.ReLU and `out += identity` of the
 * torchvision Bottleneck. */
/* dpart: RPE_BN_DPART_DOUBLES(C) doubles of scratch for the staged (deterministic) partial-sum reduction: 64 doubles that hold
 * the arrival counters of the fused reduce+finalize launch, then the slice sums.  The first 64 doubles must be ZERO before the
 * first use (the kernels leave them zero again); two launches that may run concurrently need separate dpart buffers. */
#define RPE_BN_DPART_DOUBLES(C) ((long)RPE_BN_MAX_SLICES * 2 * (C) + 64)
#define RPE_BN_MAX_SLICES 256
int rpe_bn_finalize(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, long long* num_batches, float momentum, float eps, float* scale, float* shift,
                    float* save_mean, float* save_invstd, double* dpart, void* stream);
int rpe_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                       float* scale, float* shift, void* stream);
/* out = relu?(y*scale + shift (+ residual)) */
int rpe_bn_apply(int dtype, const void* y, const void* residual, void* out, const float* scale, const float* shift, long rows, int C,
                 int relu, void* stream);
/* the same, also writing the ReLU mask as one byte per 8 channels ([rows][C/8], bit j = out[8k+j] > 0): the backward reads it
 * (1/16 of the bytes of `out`) instead of `out` to mask the gradient (rpe_bn_bwd_epilogue.a_mask).  C % 8 == 0. */
int rpe_bn_apply_mask(int dtype, const void* y, const void* residual, void* out, const float* scale, const float* shift, long rows, int C,
                      unsigned char* relu_mask, void* stream);
/* the same with a residual that is itself a raw conv output under its own BatchNorm (the projection shortcut: downsample.0 ->
 * downsample.1, no ReLU): out = [relu](y*scale + shift + res_y*res_scale + res_shift), optional packed mask -- the shortcut's own
 * apply pass and normalised copy disappear (torchvision Bottleneck.forward: `identity = self.downsample(x)`; `out += identity`) */
int rpe_bn_apply_res_bn(int dtype, const void* y, const void* res_y, const float* res_scale, const float* res_shift, void* out, const float* scale,
                        const float* shift, long rows, int C, int relu, unsigned char* relu_mask, void* stream);
/* dz = dA * (a_out > 0) (a_out null: no ReLU); dgamma, dbeta; dy = BN backward of dz; dz_out (nullable) = dz.
 * part: >= 2*1024*C floats of scratch; c1c2: 2*C floats of scratch; dpart: as for rpe_bn_finalize. */
int rpe_bn_backward(int dtype, const void* dA, const void* a_out, const void* y, const float* mean, const float* invstd,
                    const float* gamma, float* dgamma, float* dbeta, void* dy, void* dz_out, long rows, int C, float* part,
                    long part_floats, float* c1c2, double* dpart, void* stream);

/* The reduction half of rpe_bn_backward alone (dgamma, dbeta, c1c2 = mean(dz), mean(dz * xhat)): for a BatchNorm whose apply pass is
 * folded into its consumers and whose sums no fused data-gradient epilogue has produced (the stride-1 projection shortcut). */
int rpe_bn_backward_reduce(int dtype, const void* dA, const void* a_out, const void* y, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma, float* dbeta, long rows, int C, float* part, long part_floats, float* c1c2, double* dpart, void* stream);
/* second half of the fused form: partial sums -> dgamma, dbeta; dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)).
 * dy may alias dz. */
int rpe_bn_backward_from_dz(int dtype, const void* dz, const void* y, const float* mean, const float* invstd, const float* gamma,
                            const float* stats_part, int tiles, float* dgamma, float* dbeta, void* dy, long rows, int C, float* c1c2,
                            double* dpart, void* stream);

/* ------------------------------------------------------------------ pooling */
/* replaces: nn.MaxPool2d(3, 2, 1) of the ResNet stem; idx keeps the winning tap. */
int rpe_maxpool3x3s2_fwd(int dtype, const void* x, void* out, unsigned char* idx, int B, int H, int W, int C, void* stream);
/* BatchNorm apply + ReLU and the 3x3 / stride 2 / pad 1 max pool behind it in one pass over the raw conv output y [B][H][W][C] (H, W even):
 * a = relu(y * scale + shift) [B][H][W][C], out = maxpool(a) [B][H/2][W/2][C], idx = winner taps -- bitwise what rpe_bn_apply followed by
 * rpe_maxpool3x3s2_fwd give (replaces bn1 -> relu -> maxpool of torchvision's ResNet stem, util/model_utils.py:136). */
int rpe_bn_apply_maxpool3x3s2(int dtype, const void* y, const float* scale, const float* shift, void* a, void* out, unsigned char* idx, int B, int H,
                              int W, int C, void* stream);
/* the same pass with the bn1 aux head riding along (see rpe_resnet50_set_aux_head); 64 channels; `a` may be NULL (not written) */
int rpe_bn_apply_maxpool3x3s2_aux(int dtype, const void* y, const float* scale, const float* shift, void* a, void* out, unsigned char* idx, int B, int H,
                                  int W, const float* aux_w, const float* aux_bias, const float* depth_feat, float* aux_out, long ld_aux_out, float* aux_raw,
                                  unsigned char* aux_idx, void* stream);
int rpe_maxpool3x3s2_bwd(int dtype, const void* dout, const unsigned char* idx, const void* addend, void* dx, int B, int H, int W, int C,
                         void* stream);
/* Stem backward in two passes over y (fused form of rpe_maxpool3x3s2_bwd + the aux head's scatter + rpe_bn_backward for the
 * 64-channel bn1): the gradient of a1 = relu(bn1(y)) is gathered on the fly from dpool / pool_idx (B, H/2, W/2, 64) and,
 * when aux_dout != NULL, from the aux head (gradient dout[b*aux_ld + pos] (x aux_depth_feat) of the winner pixel aux_idx of
 * each 2x2 window, times aux_w[c]); the ReLU mask is recomputed from y*scale + shift.  part: >= 128 floats per block of the
 * reduction pass (up to 8192 blocks are used when it is large enough); c1c2: 128
 * floats; dpart as for rpe_bn_finalize.  replaces: autograd through torchvision's bn1 / relu / maxpool and the aux branch
 * (models/naive.py:203-211,223-231). */
int rpe_stem_bwd(int dtype, const void* dpool, const unsigned char* pool_idx, const void* y, const float* scale, const float* shift, const float* mean,
                 const float* invstd, const float* gamma, const float* aux_dout, long aux_ld, const float* aux_depth_feat,
                 const unsigned char* aux_idx, const float* aux_w, float* dgamma, float* dbeta, void* dy, int B, int H, int W, float* part,
                 long part_floats, float* c1c2, double* dpart, void* stream);
/* replaces: nn.AdaptiveAvgPool2d((1,1)) + flatten; output fp32 [B][C] */
int rpe_avgpool_fwd(int dtype, const void* x, float* out, int B, int HW, int C, void* stream);
int rpe_avgpool_bwd(int dtype, const float* dout, void* dx, int B, int HW, int C, void* stream);

/* ------------------------------------------------------------------ heads */
/* replaces: aux_nets[i] = Conv2d(64->1,1x1) -> MaxPool2d(2) -> Flatten and the
 * `aux * depth` product (models/naive.py:223-231,318-330; time_sensitive.py:377-385,477-488).
 * out[b*ld_out + p] = max2x2(a1 . w + bias) * (depth_feat ? depth_feat[b][p] : 1); raw keeps the
 * un-multiplied value, idx the winning pixel. */
int rpe_aux_head_fwd(int dtype, const void* a1, const float* w, const float* bias, const float* depth_feat, float* out, long ld_out,
                     float* raw, unsigned char* idx, int B, int H, int W, void* stream);
/* d_a1 (dense gradient of a1, zero except the winner pixels) may be NULL when the stem backward gathers it itself (rpe_stem_bwd) */
int rpe_aux_head_bwd(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                     const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, void* stream);
/* deterministic form: the parameter gradients leave every block as plain stores into the workspace and are added in block order by a
 * second tiny launch (dw[64], dbias OVERWRITTEN, bitwise reproducible; the atomic form above accumulates) */
long rpe_aux_head_bwd_workspace_floats(int dtype, int B, int H, int W);
int rpe_aux_head_bwd_det(int dtype, const float* dout, long ld_dout, const void* a1, const float* w, const float* depth_feat, const float* raw,
                         const unsigned char* idx, void* d_a1, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, float* workspace,
                         long workspace_floats, void* stream);
/* ... and from the RAW stem conv output y with bn1's scale / shift (the activated tensor was not written: rpe_bn_apply_maxpool3x3s2_aux) */
int rpe_aux_head_bwd_det_y(int dtype, const float* dout, long ld_dout, const void* y, const float* bn_scale, const float* bn_shift, const float* w,
                           const float* depth_feat, const float* raw, const unsigned char* idx, float* dw, float* dbias, float* d_depth_feat, int B, int H,
                           int W, float* workspace, long workspace_floats, void* stream);
/* replaces: depth_nets[i] = AvgPool2d(2) x2 -> InstanceNorm2d(1, affine) -> Flatten (models/naive.py:233-240) */
int rpe_depth_head_fwd(const float* depth, const float* w, const float* b, float* feat, float* xhat, int B, int H, int W, void* stream);
int rpe_depth_head_bwd(const float* d_feat, const float* xhat, long n, float* dw, float* db, void* stream);
/* The same two heads on any hooked feature map x[B][H][W][C] (feature_layer_nums other than (9,): conv1's raw output, the layer1..3
 * outputs; models/naive.py:196-240): C = 64..1024 in whole 16-byte chunks (a power of two of them).  The backward writes a DENSE
 * gradient d_x (zeroed inside: pixels outside the 2x2 windows and the windows' losers stay zero); the depth head pools
 * `pools` = int(log4(224^2 / (H W / 4))) times, flooring odd sizes as AvgPool2d does. */
int rpe_aux_head_fwd_c(int dtype, const void* x, int C, const float* w, const float* bias, const float* depth_feat, float* out, long ld_out,
                       float* raw, unsigned char* idx, int B, int H, int W, void* stream);
int rpe_aux_head_bwd_c(int dtype, const float* dout, long ld_dout, const void* x, int C, const float* w, const float* depth_feat, const float* raw,
                       const unsigned char* idx, void* d_x, float* dw, float* dbias, float* d_depth_feat, int B, int H, int W, void* stream);
int rpe_depth_head_fwd_pools(const float* depth, const float* w, const float* b, float* feat, float* xhat, int B, int H, int W, int pools, void* stream);
/* dst += src over n elements of the compute dtype (n a multiple of the 16-byte chunk) */
int rpe_tensor_add(int dtype, void* dst, const void* src, long n, void* stream);

/* replaces: nn.Linear (+ F.relu) of the proprio-fusion MLP (models/naive.py:343-345), the
 * ResNet fc (util/model_utils.py:141), the LSTM input/recurrent GEMMs and the fc heads
 * (models/time_sensitive.py:420-423,510).  y[M][N] = x[M][K] w[N][K]^T (+bias) (+addend) (relu).
 * The data gradient is the same call with the transposed weight. */
int rpe_linear_fwd(int dtype, const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int M, int N, int K,
                   int relu, const void* addend, int ld_add, void* stream);
/* the same with a caller-provided workspace: launches with few output tiles and a long reduction (the heads' layers at a few hundred
 * rows: 256 x 3655 -> 1024 is 64 tiles of 64 x 64 walking 229 K steps alone) are split along K, partial tiles added in a fixed order
 * by a second small launch (as rpe_conv2d_fwd_affine_ws).  The query returns 0 for shapes that are not split. */
long rpe_linear_fwd_workspace_bytes(int dtype, int M, int N, int K);
int rpe_linear_fwd_ws(int dtype, const void* x, int ldx, const void* w, int ldw, const float* bias, void* y, int ldy, int M, int N, int K,
                      int relu, const void* addend, int ld_add, void* workspace, long workspace_bytes, void* stream);
/* dw[N][K] (fp32, ld lddw) += dy[M][N]^T x[M][K]  (atomic accumulation) */
int rpe_linear_wgrad(int dtype, const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int M, int N, int K, void* stream);
/* deterministic form (see rpe_conv2d_wgrad_det): dw = dy^T x, or dw += dy^T x when `accumulate` (one fixed-order add per element) */
long rpe_linear_wgrad_workspace_bytes(int dtype, int M, int N, int K);
int rpe_linear_wgrad_det(int dtype, const void* dy, int lddy, const void* x, int ldx, float* dw, int lddw, int M, int N, int K, int accumulate,
                         void* workspace, long workspace_bytes, void* stream);
int rpe_transpose_f32(const float* in, float* out, int rows, int cols, int ldi, int ldo, void* stream);
int rpe_relu_bwd(const float* out, const float* dy, float* dx, long n, void* stream);
int rpe_colsum(const float* x, long rows, int cols, int ld, float* out, int accumulate, void* stream);
int rpe_copy2d(const float* src, int ld_src, float* dst, int ld_dst, long rows, int cols, void* stream);

/* replaces: the pointwise part of nn.LSTM (models/time_sensitive.py:212,235,501,759,768); gate order i,f,g,o.
 * gates[N][4H] holds x W_ih^T + h W_hh^T (no biases) and is overwritten with the activated gates. */
int rpe_lstm_cell_fwd(float* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* c_out, float* h_out, int N, int Hd,
                      void* stream);
int rpe_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c_cur, const float* dh, float* dc_io, float* dgates, int N,
                      int Hd, void* stream);

/* replaces: PoseDistanceLoss.forward (models/losses.py:47-128) and its autograd backward.
 * metric 0 l2 / 1 l1 / 2 linf / 3 combined; mode 0 position / 1 pose.
 * out3 = { loss, sum_i sqrt(|dp_i|^2+eps) , sum_i |angle_i| } -- the last two are the "val" mode
 * outputs (losses.py:95-113) computed on device instead of the per-sample numpy loop.
 * grad (nullable) = d loss / d pred. */
int rpe_pose_loss(const float* pred, const float* truth, long n, int metric, int mode, float scale, float alpha, float eps, float* out3,
                  float* grad, void* stream);

/* replaces: torch.optim.Adam(model.parameters(), lr).step() (scripts/train_model.py:228,
 * util/learn_utils.py:179) over one flat parameter / gradient / moment buffer. */
int rpe_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps, int step,
                  void* stream);

/* Dynamic loss scaling for the RPE_F16 compute path (BASELINE config C5; the reference trains in fp32 and has no counterpart:
 * util/learn_utils.py:152-179 is the loop these three calls slot into, between loss.backward() and optimizer.step()).
 * state: 8 device floats {scale, 1/scale, found_inf, skip, finite-step streak, optimizer steps taken, -, -}; nothing is read
 * back by the host.  unscale: grads *= 1/scale, found_inf |= any non-finite.  update: found_inf -> scale *= backoff, skip = 1;
 * else steps += 1, and every `growth_interval` finite steps scale *= growth.  adam_step_amp: rpe_adam_step whose step count and
 * skip decision are read from `state` on the device. */
int rpe_amp_unscale(float* grads, long n, float* state, void* stream);
int rpe_amp_update(float* state, float growth_factor, float backoff_factor, int growth_interval, void* stream);
int rpe_adam_step_amp(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps, const float* state,
                      void* stream);

/* ------------------------------------------------------------------ ResNet-50 trunk engine */
/* One object = one (batch, dtype) plan for the whole torchvision-shaped ResNet-50 body:
 * stage image -> conv1/bn1/relu -> maxpool -> 16 bottlenecks -> avgpool -> fc, forward and
 * backward, launched back to back on one stream with no host round trips.
 * replaces: self.feature_net(img) and its autograd backward (models/naive.py:316,
 * models/time_sensitive.py:472) including the bn1 forward hook (models/naive.py:211,282-283). */
typedef struct rpe_resnet50 rpe_resnet50_t;

#define RPE_RESNET50_NUM_CONV 53
/* parameter table order: see rpe_resnet50_param_name(); 161 fp32 tensors (53 conv w, 53 x (gamma, beta), fc w, fc b) */
#define RPE_RESNET50_NUM_PARAMS 161
/* buffer table: 53 x (running_mean, running_var) fp32 + 53 num_batches_tracked (int64) */
#define RPE_RESNET50_NUM_BUFFERS 106

int rpe_resnet50_create(rpe_resnet50_t** out, int batch, int height, int width, int dtype, int latent_dim);
/* the same engine for every ResNet the reference's import_resnet offers (util/model_utils.py:130-136): the bottleneck networks
 * depth 50 ([3,4,6,3] blocks), 101 ([3,4,23,3]) or 152 ([3,8,36,3]), and the BasicBlock networks 18 ([2,2,2,2] blocks of two 3x3
 * convs, identity shortcut in layer1.0, fc input 512) or 34 ([3,4,6,3]).  Parameter / buffer tables in torchvision's order for that
 * depth (rpe_resnet50_param_name); the rpe_resnet50_* entry points below take the handle of any depth.  The BasicBlock plan is
 * the general launch sequence (conv + statistics, finalize, apply; fused data gradient + BatchNorm reduction; dz, y -> dy; weight
 * gradients on the second stream) -- the bottleneck-only forms (folded conv3 backward, y3-free blocks) do not apply to it. */
int rpe_resnet_create(rpe_resnet50_t** out, int depth, int batch, int height, int width, int dtype, int latent_dim);
void rpe_resnet50_destroy(rpe_resnet50_t* e);
/* bytes of device workspace the engine needs (activations, gradients, packed weights, scratch) */
long rpe_resnet50_workspace_bytes(const rpe_resnet50_t* e);
/* state_dict key (torchvision naming, e.g. "layer1.0.conv1.weight") and element count of parameter i */
const char* rpe_resnet50_param_name(const rpe_resnet50_t* e, int i);
long rpe_resnet50_param_numel(const rpe_resnet50_t* e, int i);
/* host arrays of device pointers: params/grads [NUM_PARAMS] fp32 (conv weights in channels_last
 * storage = [Co][kh][kw][Ci]; conv1.weight plain OIHW), running stats [2*53], num_batches [53] */
int rpe_resnet50_bind(rpe_resnet50_t* e, void* workspace, long workspace_bytes, float* const* params_host, float* const* grads_host,
                      float* const* running_host, long long* const* num_batches_host);
/* re-pack compute-dtype copies of the weights (call after every optimizer step / load_state_dict) */
int rpe_resnet50_pack_weights(rpe_resnet50_t* e, void* stream);
/* The fp32 masters or the BN running statistics bound to this engine were changed by someone else (an optimizer step or a
 * training forward that ran through ANOTHER engine object bound to the same tensors, load_state_dict): every cached weight
 * copy -- the compute-dtype training copies and the BN-folded inference copies -- is rebuilt by the next forward.
 * replaces: nothing in the reference (torch modules read their parameters in place, util/model_utils.py:136-141). */
int rpe_resnet50_weights_changed(rpe_resnet50_t* e);
/* The engine's second stream (weight gradients in the backward, the projection shortcuts in the training forward) is PROBED when it is
 * created (first training pass): does a kernel on it run while a kernel of the caller's stream executes?  HIP deals streams to a few
 * hardware queues in creation order, and a stream that shares a queue (or, at low priority, a pipe) with the caller's serialises or
 * starves (Schedule).  candidates: streams created until one overlapped (1..4, 0 before the first training pass);
 * concurrent: 1 the stream in use overlapped, 0 none of four did, -1 not probed (RPE_NO_SIDE_PROBE=1, or created under capture). */
int rpe_resnet50_side_stream_info(const rpe_resnet50_t* e, int* candidates, int* concurrent);
/* The bn1 aux head (aux_nets[0]: Conv2d(64 -> 1, 1x1) + MaxPool2d(2) + Flatten [x the depth feature], models/naive.py:223-231,318-330)
 * computed BY the next training forward, inside the stem's BatchNorm-apply + ReLU + max-pool pass (rpe_bn_apply_maxpool3x3s2_aux): the
 * activated tensor relu(bn1(conv1 x)) is then never written (rpe_resnet50_early_feature returns NULL after such a forward) and the
 * separate aux launch with its re-read disappears.  out: the aux columns of the fused feature rows (row pitch ld_out floats); raw / idx:
 * [B][(H/4)(W/4)] window maxima and winner taps for the backward.  The setting is consumed by ONE forward; w = NULL clears it.
 * rpe_resnet50_aux_head_bwd: the head's parameter gradients (dw [64], dbias, d_depth_feat) from the raw stem output, as
 * rpe_aux_head_bwd_det computes them from a1 (workspace: rpe_aux_head_bwd_workspace_floats); its gradient towards the stem still
 * travels in compact form through rpe_resnet50_set_aux_grad. */
int rpe_resnet50_set_aux_head(rpe_resnet50_t* e, const float* w, const float* bias, const float* depth_feat, float* out, long ld_out, float* raw,
                              unsigned char* idx);
int rpe_resnet50_aux_head_bwd(rpe_resnet50_t* e, const float* dout, long ld_dout, const float* w, const float* depth_feat, const float* raw,
                              const unsigned char* idx, float* dw, float* dbias, float* d_depth_feat, float* workspace, long workspace_floats, void* stream);
/* img: (B,3,H,W) fp32 NCHW.  features: fp32 [B][ld_features] (first latent_dim columns written).
 * training != 0: batch statistics + running-stat update and everything backward needs is kept. */
int rpe_resnet50_forward(rpe_resnet50_t* e, const float* img_nchw, float* features, long ld_features, int training, void* stream);
/* the same from raw simulator frames (uint8 [B][Hs][Ws][3]): crop + normalise + stage in one kernel (rpe_stage_frames_u8) */
int rpe_resnet50_forward_u8(rpe_resnet50_t* e, const unsigned char* frames, int Hs, int Ws, const float* mean3_host, const float* std3_host,
                            float* features, long ld_features, int training, void* stream);
/* ... with the resize in front (rpe_stage_frames_u8_resized); `rs`: its geometry and device tables */
typedef struct {
    int Hr, Wr, top, left, ksx, ksy;
    const int *xb, *xk, *yb, *yk;
    unsigned char* tmp;
} rpe_resize_plan;
int rpe_resnet50_forward_u8_resized(rpe_resnet50_t* e, const unsigned char* frames, int Hs, int Ws, const rpe_resize_plan* rs, const float* mean3_host,
                                    const float* std3_host, float* features, long ld_features, int training, void* stream);
/* the hooked early feature relu(bn1(conv1 x)): NHWC [B][H/2][W/2][64] in the compute dtype (inside the workspace) */
const void* rpe_resnet50_early_feature(const rpe_resnet50_t* e);
/* gradient buffer of the early feature; the caller writes d(loss)/d(early) there (or passes use_d_early = 0) */
void* rpe_resnet50_early_grad(rpe_resnet50_t* e);
/* parameter gradients are WRITTEN (not accumulated) into the bound grad tensors */
int rpe_resnet50_backward(rpe_resnet50_t* e, const float* d_features, long ld_d_features, int use_d_early, void* stream);
/* Optional measurement aid: HIP events around every launch of the plan (on the launch stream), summed per
 * category.  enable=1 clears the counters.  Used by bench.py for the roofline line; off in timed regions. */
enum { RPE_PROF_CONV_FWD = 0, RPE_PROF_CONV_DGRAD = 1, RPE_PROF_CONV_WGRAD = 2, RPE_PROF_BN_FWD = 3, RPE_PROF_BN_BWD = 4,
       RPE_PROF_OTHER = 5, RPE_PROF_NUM = 6 };
int rpe_resnet50_profile(rpe_resnet50_t* e, int enable);
int rpe_resnet50_profile_read(rpe_resnet50_t* e, float* ms, int* launches, double* flops, double* bytes);
/* the same spans aggregated per kernel symbol: text lines "name;launches;ms;flops;bytes" (algorithmic work) written to buf; returns bytes written */
long rpe_resnet50_profile_kernels(rpe_resnet50_t* e, char* buf, long buflen);
/* short name of the implicit-GEMM kernel instance the last conv / Linear call on this thread launched,
 * e.g. "nt_kernel<bf16,2,128,4,0,3,1>" = <dtype, waves_m, BN, chunks per K-row, mode, ring depth, role> */
const char* rpe_last_kernel_name(void);
/* Walk direction of the calling thread's NEXT launches of the direction-aware kernels (the implicit-GEMM kernels and the BatchNorm apply
 * passes): every XCD owns one contiguous eighth of a launch's row tiles / spans and walks it upwards (mode 0, the default), downwards
 * (1), or alternately launch by launch starting upwards (2) -- a consumer that walks against its producer's direction reads what the
 * producer wrote last, i.e. what the 256-MiB Infinity Cache still holds of a larger tensor, first (tools/micro/mall_order.hip).
 * Results do not depend on it (partial sums are indexed by tile).  The engine sets mode 2 around its training step and restores the
 * caller's mode; RPE_NO_WALK_ALT=1 keeps it at 0.  No reference counterpart (scheduling only). */
void rpe_set_walk_direction(int mode);
/* Narrowest feature map (in pixels) whose 3x3 / stride-1 / pad-1 DETERMINISTIC weight gradient (rpe_conv2d_wgrad_det, 16-bit element types,
 * out_c % 64 == 0, in_c % 32 == 0 [% 64 for 64 output channels], width <= 60) runs in the halo form (csrc/wgrad_halo.hip: all nine taps in
 * one workgroup, x streamed once through a ring of pixels in LDS; the reduction walks the zero-padded pixel grid, which costs
 * (H + 2)(W + 2) / (H W) of the useful work -- hence a lower bound on the width).  Default 28 (RPE_WGRAD_HALO_MINW); returns the previous
 * value.  Process-wide; set it before sizing workspaces (rpe_conv2d_wgrad_workspace_bytes follows the choice).  Scheduling / summation
 * order only: no reference counterpart. */
int rpe_conv2d_wgrad_halo_min_width(int width);
/* The same backward in stages, so a data-parallel caller can all-reduce finished gradients while the rest is computed:
 * begin (fc + avgpool) -> blocks(count, join=1) ... until all 16 blocks are done -> end (stem).  After blocks(.., join=1)
 * every gradient of the blocks processed so far is complete in stream order on `stream`. */
int rpe_resnet50_backward_begin(rpe_resnet50_t* e, const float* d_features, long ld_d_features, void* stream);
int rpe_resnet50_backward_blocks(rpe_resnet50_t* e, int count, int join, void* stream);
int rpe_resnet50_backward_end(rpe_resnet50_t* e, int use_d_early, void* stream);
/* Backward of a FROZEN trunk -- import_resnet freezes every ResNet parameter when feature_extract and use_pretrained are both set
 * (util/model_utils.py:110-113,136-137; the setting of every published job: scripts/train_no.sbatch:83, train_tdo.sbatch:83,
 * train_tdo_v2.sbatch:84) and then installs a fresh, trainable fc (util/model_utils.py:140-141): only that fc's weight and bias take
 * a gradient, nothing flows into the body (no data gradient, no BN backward, no conv weight gradients: ~2/3 of the step's work).
 * The forward stays the training forward (BatchNorm on batch statistics, running statistics updated), as torch runs it. */
int rpe_resnet50_backward_frozen(rpe_resnet50_t* e, const float* d_features, long ld_d_features, void* stream);
/* Instead of a dense early-feature gradient tensor (rpe_resnet50_early_grad), hand the aux head's gradient to the next backward
 * in its compact form (see rpe_stem_bwd); the pointers must stay valid until that backward's end stage has been enqueued.
 * aux_dout == NULL clears it. */
int rpe_resnet50_set_aux_grad(rpe_resnet50_t* e, const float* aux_dout, long aux_ld, const float* aux_depth_feat, const unsigned char* aux_idx,
                              const float* aux_w);
/* debugging / parity: device pointer, rows and channels of a named intermediate (e.g. "layer1.0.y1") */
int rpe_resnet50_tensor(const rpe_resnet50_t* e, const char* name, const void** ptr, long* rows, int* channels);
/* Hooks other than bn1.  set_hook_grad: a dense gradient (compute dtype, the tensor's own NHWC shape) of conv1's raw output (layer 0:
 * `conv1.y`) or of the output of layer1..3 (`layerN.<last>.conv3.a`, see rpe_resnet50_tensor) that the NEXT backward adds where that
 * tensor's gradient is formed (layer outputs: into the shortcut gradient of the following stage-entry block; conv1: behind BN1's
 * backward, which then runs unfused); cleared by that backward; the tensor must stay alive until it has been enqueued.
 * set_stem_raw: the inference forward keeps conv1's raw output (conv and BN as two launches instead of the folded one). */
int rpe_resnet50_set_hook_grad(rpe_resnet50_t* e, int layer, const void* dense_grad);
int rpe_resnet50_set_stem_raw(rpe_resnet50_t* e, int on);

#ifdef __cplusplus
}
#endif
#endif /* RPE_HIP_H */
