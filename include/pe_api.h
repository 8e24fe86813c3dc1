/*
 * pe_api.h — C ABI of libpe_hip.so, the MI355X (gfx950) pose-estimation hot path.
 *
 * The reference (preritj/pose_estimation) is Python-only and exposes no FFI;
 * the files that held these units are named by its project manifest but are
 * absent from the snapshot.  Each entry point below cites the reference
 * location whose work it takes over.
 *
 * Conventions
 *   - every function returns 0 (PE_OK) or a negative PE_E_* code; the message
 *     of the last failure on the calling thread is pe_last_error();
 *   - all tensor arguments are caller-owned DEVICE pointers unless a parameter
 *     is documented as host memory; the library allocates nothing;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every compute call is asynchronous on that stream and never synchronises;
 *   - sizes are int32 element counts, strides are int64 element strides.
 */
#ifndef PE_API_H
#define PE_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PE_OK             0
#define PE_E_INVALID     (-1)  /* bad argument (null pointer, negative size, misalignment) */
#define PE_E_HIP         (-2)  /* a HIP runtime call or kernel launch failed */
#define PE_E_UNSUPPORTED (-3)  /* valid request outside what the kernels cover */

#define PE_DTYPE_F32  0
#define PE_DTYPE_BF16 1

#define PE_MAX_RADIUS 31       /* largest Gaussian window radius pe_render/pe_mse accept */

typedef void* pe_stream_t;

/* Library ABI version (major*1000 + minor). */
int32_t pe_version(void);

/* Message of the most recent failure on this thread ("" if none). Never NULL. */
const char* pe_last_error(void);

/* Number of HIP devices visible to the library, or PE_E_HIP. Host-only query. */
int32_t pe_device_count(void);

/* ------------------------------------------------------------------------
 * Launch tape: a recorded call list for steps whose shapes and buffers do not change.
 * Replaces nothing in the reference, whose session loop (train.py, absent; test.ipynb:1060
 * drives it) re-runs a compiled TensorFlow graph per step: this is the same idea for this
 * library's launches.  Between pe_tape_begin and pe_tape_end every kernel (and memset, and
 * pe_stream_wait_stream) that the calling thread enqueues through this library is executed as
 * usual AND stored with a copy of its arguments; pe_tape_replay enqueues records
 * [first, first + count) again, on the streams and with the pointers they were recorded with,
 * at the cost of one hipLaunchKernel each.  The caller keeps every buffer the recorded calls
 * named alive and at the same address, and replays whatever it enqueued itself between the
 * recorded calls (pe_tape_length tells where in the tape it stood).  Entry points whose
 * arguments change from call to call (the optimizer's step number) are left outside the
 * recorded region.  Two kinds of call REFUSE to run while the calling thread records
 * (PE_E_UNSUPPORTED), because a replay of them would be silently wrong: pe_allreduce_bucket
 * (RCCL enqueues the collective itself, only the scale kernel would land on the tape) and the
 * one-launch batch-norm forms that take a `sync` buffer (pe_bn_finalize_apply_bf16,
 * pe_bn_stats_apply_bf16, pe_bn_bwd_bf16 / pe_bn_bwd_from_partials_bf16 with sync != NULL:
 * their arrival-counter base is a host-side count that a replay would not advance).
 * One tape may record per thread at a time.
 * ------------------------------------------------------------------------ */
typedef void* pe_tape_t;
int32_t pe_tape_begin(void);
/* Records stored so far on the recording tape, or -1 when none is recording. */
int64_t pe_tape_length(void);
int32_t pe_tape_end(pe_tape_t* tape);
int32_t pe_tape_replay(pe_tape_t tape, int64_t first, int64_t count);
int32_t pe_tape_destroy(pe_tape_t tape);
/* `waiter` waits for everything enqueued on `waited` so far (event record + stream wait);
 * recorded on a tape like a launch. */
int32_t pe_stream_wait_stream(pe_stream_t waiter, pe_stream_t waited);

/* ------------------------------------------------------------------------
 * a1. Heat-map peak decode.
 * Replaces: the heat-map -> keypoint step of inference.py / utils/ops.py
 * (PoseEstimation.pyproj:28,37; both absent).  The notebook's nearest call is
 * heatmaps[:, :, kp].nonzero() at test.ipynb:931.
 *
 * hm is a logical [B,K,H,W] float32 tensor addressed as
 *   hm[b*sB + k*sK + y*sH + x*sW]
 * so both NCHW (sW = 1) and the reference's NHWC label layout (sK = 1,
 * test.ipynb:931) are accepted.  Per (b,k):
 *   idx  = lowest y*W + x attaining the maximum          -> idx  [B,K] int32
 *   conf = that maximum                                   -> conf [B,K] float32
 *   (x,y)= (idx % W, idx / W); if refine != 0 and 1 < x < W-1 and 1 < y < H-1
 *          each coordinate moves 0.25 towards its larger neighbour (none on a
 *          tie); if conf <= 0 both are 0                  -> xy [B,K,2] float32
 * Inputs are assumed finite.  What a map with non-finite elements gives is
 * nevertheless fixed (every compare is an ordered float compare):
 *   - a NaN never wins (numpy.argmax would return the first NaN): idx is the
 *     lowest index of the largest non-NaN element above -inf;
 *   - a map with no element above -inf (NaN and/or -inf only) gives idx 0,
 *     conf = hm[0] (NaN if that is NaN) and xy = (0, 0);
 *   - a NaN neighbour is neither larger nor smaller than the opposite one, so
 *     the coordinate of that axis does not move.
 * ---------------------------------------------------------------------- */
int32_t pe_decode_f32(const float* hm, int32_t B, int32_t K, int32_t H, int32_t W,
                      int64_t sB, int64_t sK, int64_t sH, int64_t sW,
                      int32_t refine, float* xy, float* conf, int32_t* idx,
                      pe_stream_t stream);

/* ------------------------------------------------------------------------
 * a2. Gaussian target rendering.
 * Replaces: the label-map rendering inside the tf.data map function of
 * dataset/pose_data.py / utils/ops.py (PoseEstimation.pyproj:27,37; absent),
 * whose output is visible at test.ipynb:606-614, 919-922.
 *
 * pe_gaussian_table_f32 is HOST-only: table[i] = (float)exp(-(double)i*i /
 * (2*sigma*sigma)), i = 0..radius, written to host memory `table`
 * (radius+1 floats).  It is the table the render / fused-loss kernels use.
 *
 * pe_render_targets_f32: joints [B,K,3] float32 = (x, y, v) in input pixels.
 *   mu      = floor(coord / stride + 0.5)       (float32 arithmetic)
 *   weight  = 0 if !(v > 0), if a coordinate is not finite, or if the centre
 *             cell mu is off the map (the label of test.ipynb:606-653 leaves
 *             the heat map of such a joint empty); else 1
 *   target  = table[|x-mu_x|] * table[|y-mu_y|] inside the window, 0 elsewhere,
 *             all 0 when weight == 0
 * target is a logical [B,K,H,W] float32 tensor addressed through
 * (tB,tK,tH,tW) like hm above; weight is [B,K] float32, contiguous.
 * ---------------------------------------------------------------------- */
int32_t pe_gaussian_table_f32(float sigma, int32_t radius, float* table);

int32_t pe_render_targets_f32(const float* joints, int32_t B, int32_t K, int32_t H, int32_t W,
                              float stride, float sigma, int32_t radius,
                              float* target, int64_t tB, int64_t tK, int64_t tH, int64_t tW,
                              float* weight, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * Vector-map labels and their loss (the reference's second dense label).
 * Replaces: the `vecmaps` label rendering of dataset/pose_data.py
 * (PoseEstimation.pyproj:27; absent), whose output is printed at
 * test.ipynb:634-653 and read back by the drawing cell as vecmaps[y, x, 4*i]
 * (x offset) and vecmaps[y, x, 4*i + 1] (y offset), in cells.
 *
 * pairs_host: HOST array of 2*num_pairs joint indices (num_pairs <= 32).
 * vec is a logical [B, 4*num_pairs, H, W] float32 tensor addressed through
 * element strides (sB,sC,sH,sW); NHWC is (H*W*C, 1, W*C, C).  For pair
 * i = (k1, k2) with centres c = floor(coord/stride + 0.5):
 *   channel 4i, 4i+1   : cells with |x-c1.x| <= radius and |y-c1.y| <= radius
 *                        hold (c2.x - x, c2.y - y)
 *   channel 4i+2, 4i+3 : cells of k2's window hold (c1.x - x, c1.y - y)
 * everything else is 0.  Windows are clipped to the map and a centre may lie
 * off the map; both joints must be labelled (v > 0) and finite.
 *
 * pe_vecmap_loss_fwd_bwd: pred / grad contiguous [B, 4*num_pairs, H, W],
 * float32 or bf16;  L = 0.5/n * sum over window cells of m * (p - t)^2,
 * grad = m * (p - t) / n inside windows and 0 elsewhere, n = B*4*num_pairs*H*W.
 * The label is rebuilt in registers, never read from memory.  mask [B,H,W]
 * uint8 (NULL = none); per_channel: 4*num_pairs floats, mean == loss (may be
 * NULL); grad may be NULL; workspace >= pe_vecmap_workspace_bytes(B, num_pairs).
 * ---------------------------------------------------------------------- */
int32_t pe_render_vecmaps_f32(const float* joints, int32_t B, int32_t K,
                              const int32_t* pairs_host, int32_t num_pairs, int32_t H, int32_t W,
                              float stride, int32_t radius, float* vec, int64_t sB, int64_t sC,
                              int64_t sH, int64_t sW, pe_stream_t stream);

size_t pe_vecmap_workspace_bytes(int32_t B, int32_t num_pairs);

int32_t pe_vecmap_loss_fwd_bwd(const void* pred, int32_t pred_dtype, const float* joints,
                               const uint8_t* mask, int32_t B, int32_t K,
                               const int32_t* pairs_host, int32_t num_pairs, int32_t H, int32_t W,
                               float stride, int32_t radius, float* loss, float* per_channel,
                               void* grad, void* workspace, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * a3. Masked per-joint MSE loss and its gradient.
 * Replaces: the heat-map loss of model/base_model.py (PoseEstimation.pyproj:30;
 * absent); the per-pixel ignore mask is the `masks` label of test.ipynb:949.
 *
 *   L    = 0.5/(B*K*H*W) * sum_{b,k,y,x} m[b,y,x] * (w[b,k] * (p - t))^2
 *   grad = w[b,k]^2 * m[b,y,x] * (p - t) / (B*K*H*W)
 *
 * pred / grad: contiguous [B,K,H,W], float32 (PE_DTYPE_F32) or bf16
 *              (PE_DTYPE_BF16); grad may be NULL (loss only).
 * Exactly one of `target` (contiguous [B,K,H,W] float32) and `joints`
 * ([B,K,3] float32) is non-NULL.  With `joints` the target is rendered on the
 * fly exactly as pe_render_targets_f32 would (stride, sigma, radius) and never
 * touches memory; `weight` is then an OUTPUT ([B,K], may be NULL).  With
 * `target`, `weight` is an INPUT [B,K] (NULL = all ones).
 * mask: [B,H,W] uint8, 0 = ignore pixel; NULL = no mask.
 * loss: 1 float32; per_joint: K float32 with mean(per_joint) == loss (may be NULL).
 * workspace: device scratch of at least pe_mse_workspace_bytes(B,K) bytes.
 * The reduction order is fixed, so results are run-to-run bit-stable.
 * ---------------------------------------------------------------------- */
size_t pe_mse_workspace_bytes(int32_t B, int32_t K);

int32_t pe_mse_fwd_bwd(const void* pred, int32_t pred_dtype,
                       const float* target, const float* joints,
                       float* weight, const uint8_t* mask,
                       int32_t B, int32_t K, int32_t H, int32_t W,
                       float stride, float sigma, int32_t radius,
                       float* loss, float* per_joint, void* grad,
                       void* workspace, pe_stream_t stream);

/* pe_mse_fwd_bwd without the workspace argument and without any allocation.
 * No counterpart in the reference: it is for callers that cannot allocate or
 * size device memory around a call (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  Same arguments, same results bit for
 * bit; the per-map partial sums go to a device array inside the library
 * (PE_MSE_STATIC_MAX_MAPS floats = 256 KB, one copy per GPU, held until the
 * process ends).  B*K > PE_MSE_STATIC_MAX_MAPS returns PE_E_UNSUPPORTED.  The
 * call uses the scratch of the CURRENT device, so `stream` must belong to it.
 * All calls on one GPU share that array: two calls that overlap on one GPU
 * (different streams, or threads) race; calls on one stream are ordered.  The
 * first call on a device looks up the array's address (hipGetSymbolAddress,
 * not an allocation); make that first call outside a graph capture. */
#define PE_MSE_STATIC_MAX_MAPS 65536
int32_t pe_mse_fwd_bwd_static(const void* pred, int32_t pred_dtype,
                              const float* target, const float* joints,
                              float* weight, const uint8_t* mask,
                              int32_t B, int32_t K, int32_t H, int32_t W,
                              float stride, float sigma, int32_t radius,
                              float* loss, float* per_joint, void* grad,
                              pe_stream_t stream);

/* ------------------------------------------------------------------------
 * a6. Optimizer steps over a flat float32 parameter arena.
 * Replaces: the optimizer of train.py / model/base_model.py
 * (PoseEstimation.pyproj:30,34; absent); hyper-parameters at
 * test.ipynb:1001-1008 (adam, beta1 0.9, beta2 0.999, epsilon 1e-8, lr 1e-3).
 *
 * pe_adam_step: for i < n, with g = grad[i]*grad_scale,
 *   m = beta1*m + (1-beta1)*g;  v = beta2*v + ((1-beta2)*g)*g
 *   p -= (step_size*m) / (sqrt(v)*denom_scale + eps)
 * step_size/denom_scale carry the bias correction and are computed by the
 * caller: TF-1 Adam is (lr*sqrt(1-b2^t)/(1-b1^t), 1); the eps-on-v_hat form is
 * (lr/(1-b1^t), 1/sqrt(1-b2^t)).  p_bf16 (may be NULL) receives the bf16
 * working copy of the updated parameters.
 *
 * pe_sgd_step: mom = momentum*mom + g; p -= lr*mom.
 * All pointers must be 16-byte aligned (p_bf16: 8-byte).
 * ---------------------------------------------------------------------- */
int32_t pe_adam_step(float* p, float* m, float* v, const float* grad, uint16_t* p_bf16,
                     int64_t n, float beta1, float beta2, float eps,
                     float step_size, float denom_scale, float grad_scale,
                     pe_stream_t stream);

int32_t pe_sgd_step(float* p, float* mom, const float* grad, uint16_t* p_bf16,
                    int64_t n, float lr, float momentum, float grad_scale,
                    pe_stream_t stream);

/* ------------------------------------------------------------------------
 * a4 / a5. Convolution backbone and deconvolution head (bf16 MFMA implicit GEMM).
 * Replaces: nets/mobilenet_v2.py, model/mobilenet_pose.py, model/base_model.py
 * (PoseEstimation.pyproj:30-32; absent) - the conv / deconv arithmetic the
 * reference delegated to TensorFlow 1.x kernels.
 *
 * Tensors: activations are NHWC bf16; packed weights are bf16 with the
 * reduction axis contiguous:
 *   conv    : [Cout][KH][KW][Cin]
 *   deconv  : [4 phases][Cout][2x2 taps][Cin]   (phase = 2*py + px of the output
 *             pixel parity; tap (a, b) of phase (py, px) is kernel element
 *             kh = (py ? 2 - 2a : 1 + 2a) ... see pe_pack layout in ops.py)
 * Accumulation is fp32.  Epilogue: y = acc*scale[c] + shift[c] (+ residual)
 * (-> ReLU) -> bf16; scale/shift/residual may be NULL.
 * Cin must be a multiple of 8 (a tap's run is walked in steps of 64; the part of the last
 * step past the run reads zeros, nothing is padded in memory; the transposed conv needs a
 * multiple of 64), Cout a multiple of 4 (except f32-NCHW output).
 * Narrow inputs (pe_conv2d_fwd_bf16 only): Cin of 8, 16 or 32 is accepted with weights
 * packed [Cout][KH][ceil(KW/P)][P*Cin], P = 64/Cin: run c of a filter row holds taps
 * kw = c*P .. c*P+P-1 pixel by pixel (zero where kw >= KW), because P horizontally
 * adjacent NHWC pixels are one contiguous 64-element run.
 * ---------------------------------------------------------------------- */
typedef struct pe_conv_desc {
    int32_t B, H, W, Cin;       /* input  [B,H,W,Cin] */
    int32_t Cout, KH, KW;       /* filter */
    int32_t stride, pad;
    int32_t relu;               /* 1: ReLU in the epilogue */
    int32_t out_f32_nchw;       /* 0: y is bf16 NHWC [B,Ho,Wo,Cout]; 1: float32 NCHW [B,Cout,Ho,Wo] */
} pe_conv_desc;

int32_t pe_conv2d_fwd_bf16(const pe_conv_desc* d, const void* x, const void* w,
                           const float* scale, const float* shift, const void* residual,
                           void* y, pe_stream_t stream);

/* Up to three convolutions over the same pixels in one launch (inference): stage 0 is any
 * stride-1 or strided conv pe_conv2d_fwd_bf16 takes with Cin a multiple of 64; every later stage
 * is a 1x1 stride-1 conv whose input is the previous stage's output y (its desc.B/H/W/Cin must
 * say so).  A workgroup computes its 256 output pixels of stage 0, then the same pixels of
 * stage 1 from what it has just stored, and so on, so a bottleneck's conv2 -> conv3 -> next
 * conv1 costs one launch and the intermediate tensors are read back from the cache their
 * producer filled.  Every stage's y is written in full, bit for bit what the stage would give
 * alone.  Each stage has its own epilogue operands (scale, shift, residual may be NULL; relu
 * in desc).  bf16 NHWC outputs only, Cout a multiple of 8.
 * Replaces: consecutive layer calls of the reference's network files (nets/mobilenet_v2.py,
 * model/mobilenet_pose.py; PoseEstimation.pyproj:31-32, files absent). */
typedef struct pe_conv_stage {
    pe_conv_desc desc;
    const void* w;
    const float* scale;
    const float* shift;
    const void* residual;
    void* y;
} pe_conv_stage;

int32_t pe_conv2d_fwd_chain_bf16(const void* x, const pe_conv_stage* stages, int32_t nstage,
                                 pe_stream_t stream);

/* ConvTranspose 4x4, stride 2, pad 1: x [B,H,W,Cin] -> y [B,2H,2W,Cout], four parity phases. */
int32_t pe_deconv4x4s2_fwd_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                int32_t relu, const void* x, const void* w_phases,
                                const float* scale, const float* shift, void* y,
                                pe_stream_t stream);

/* The transposed conv above followed by a 1x1 conv over its output (the pose head's last
 * deconv + the final conv to K heat maps, model/mobilenet_pose.py, PoseEstimation.pyproj:31;
 * absent), in ONE launch: the deconv's [B,2H,2W,Cout] output tile stays in LDS, only
 *   out[b][k][y][x] = head_bias[k] + sum_c head_w[k][c] * bf16(relu?(deconv(x)*scale+shift))[b,y,x,c]
 * is written, float32 NCHW [B,head_N,2H,2W].  Bit-identical to pe_deconv4x4s2_fwd_bf16 followed
 * by pe_conv2d_fwd_bf16(out_f32_nchw = 1) on the same operands, without the 2 x B*4HW*Cout*2-byte
 * round trip through memory.  Cout must be 128, 256, 384 or 512, Cin a multiple of 64,
 * 1 <= head_N <= 32; partial sums over the first channel tiles are parked in out_f32_nchw itself;
 * head_w is bf16 [head_N][Cout], 16-byte aligned; head_bias float32 [head_N] or NULL. */
int32_t pe_deconv4x4s2_head_fwd_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                     int32_t relu, const void* x, const void* w_phases,
                                     const float* scale, const float* shift, const void* head_w,
                                     const float* head_bias, int32_t head_N, float* out_f32_nchw,
                                     pe_stream_t stream);

/* 7x7 stride-2 pad-3 stem on a 3-channel image, computed as a 4x4 stride-1 conv over the
 * 2x2 space-to-depth image: x_s2d [B,H/2,W/2,16] bf16 (12 real channels (ry,rx,c) + 4 zero),
 * w [Cout][4 rows][4 cols x 16] bf16 -> y [B,H/2,W/2,Cout]. */
int32_t pe_stem7x7s2_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout, int32_t relu,
                              const void* x_s2d, const void* w, const float* scale,
                              const float* shift, void* y, pe_stream_t stream);

/* The stem above (with ReLU) followed by the 3x3 stride-2 max pool, in ONE launch (inference):
 * x_s2d [B,H2,W2,16] -> y_pooled [B,H2/2,W2/2,Cout]; the stem's own [B,H2,W2,Cout] map is
 * never written.  Equals pe_maxpool3x3s2_fwd_bf16(pe_stem7x7s2_fwd_bf16(relu = 1)) bit for bit.
 * Replaces the same reference code as the two entry points it fuses (nets/mobilenet_v2.py,
 * PoseEstimation.pyproj:32; absent).  Cout must be 64; H2 and W2 even. */
int32_t pe_stem7x7s2_pool_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                   const void* x_s2d, const void* w, const float* scale,
                                   const float* shift, void* y_pooled, pe_stream_t stream);

/* 3x3 stride-1 pad-1 conv with 64 input and 64 output channels (+ scale, shift, relu as in
 * pe_conv2d_fwd_bf16), x, y [B,H,W,64] bf16 NHWC, w [64][3][3][64] bf16: the 3x3 layers of the
 * first backbone stage (nets/mobilenet_v2.py, PoseEstimation.pyproj:32; absent) for inference.
 * One persistent workgroup per CU keeps the whole filter in LDS and stages each input patch
 * once instead of once per tap; two groups of waves alternate between multiplying one tile and
 * converting / storing the previous one.  H, W <= 8192 (PE_E_UNSUPPORTED beyond).  Equals
 * pe_conv2d_fwd_bf16 on the same operands bit for bit. */
int32_t pe_conv3x3_c64_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x, const void* w,
                                const float* scale, const float* shift, int32_t relu, void* y,
                                pe_stream_t stream);

/* The 32-channel twin: 3x3 stride-1 pad-1 conv with 32 input and 32 output channels (+ scale,
 * shift, residual, relu as in pe_conv2d_fwd_bf16), x, residual, y [B,H,W,32] bf16 NHWC, w in
 * the narrow packing [32][3][2][64] that pe_conv2d_fwd_bf16 takes for 32 input channels: the
 * 3x3 layers of the highest-resolution branch of a multi-branch backbone (the files under nets/,
 * PoseEstimation.pyproj:31-32; absent) for inference.  Persistent workgroups with the whole
 * filter in LDS (60 KB per workgroup: they share a CU with other launches).  H, W <= 8192.
 * Equals pe_conv2d_fwd_bf16 on the same operands bit for bit. */
int32_t pe_conv3x3_c32_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                const void* w_narrow, const float* scale, const float* shift,
                                const void* residual, int32_t relu, void* y, pe_stream_t stream);

/* The same 3x3 layer followed, in the same launch, by the pointwise (1x1) layer that reads it:
 * y = act3( conv1x1( bf16( act(conv3x3(x, w) * scale + shift) ), w3 ) * scale3 + shift3 + residual ),
 * w3 [Cout3][64] bf16, scale3 / shift3 [Cout3] float32 (NULL = 1 / 0), residual and y
 * [B,H,W,Cout3] bf16, relu3 0 or 1: conv2 and conv3 of a bottleneck of the first backbone
 * stage (nets/mobilenet_v2.py, PoseEstimation.pyproj:32; absent).  The 64-channel tensor between
 * the two layers is never written: a finished tile stays in registers, rounded to bf16 exactly
 * as the separate launch would store it, and is the activation operand of the 1x1 layer.
 * Cout3 must be 256, scale / shift must be given (folded batch norm) and relu must be 1
 * (PE_E_UNSUPPORTED otherwise).  Equals pe_conv2d_fwd_bf16(1x1, residual, relu3) of
 * pe_conv3x3_c64_fwd_bf16 bit for bit. */
int32_t pe_conv3x3_c64_pw_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x, const void* w,
                                   const float* scale, const float* shift, int32_t relu,
                                   const void* w3, int32_t Cout3, const float* scale3,
                                   const float* shift3, const void* residual, int32_t relu3,
                                   void* y, pe_stream_t stream);

/* The same pair of layers for the FIRST bottleneck of a stage, whose residual is itself a
 * pointwise projection of the block's input x0 [B,H,W,64] (the "downsample" branch):
 * y = relu( conv1x1(bf16(act(conv3x3(x))), w3) * scale3 + shift3
 *           + bf16( conv1x1(x0, wd) * scaled + shiftd ) ),
 * wd [Cout3][64] bf16, scaled / shiftd [Cout3] float32 (NULL = 1 / 0).  The projection is
 * computed per output tile inside the launch, rounded to bf16 exactly as its own launch would
 * store it; its Cout3-channel tensor is never written or read.  Same restrictions as
 * pe_conv3x3_c64_pw_fwd_bf16; equals the three separate launches bit for bit. */
int32_t pe_conv3x3_c64_pw_proj_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                        const void* w, const float* scale, const float* shift,
                                        int32_t relu, const void* w3, int32_t Cout3,
                                        const float* scale3, const float* shift3, const void* x0,
                                        const void* wd, const float* scaled, const float* shiftd,
                                        void* y, pe_stream_t stream);

/* pe_conv3x3_c64_pw_fwd_bf16 (relu3 = 1) for a bottleneck that is FOLLOWED by another one of the
 * same stage: the launch also computes the next block's first pointwise layer on the tile it
 * has just finished,
 * y_next = relu( conv1x1(y, w_next) * scale_next + shift_next ),
 * w_next [Cout_next][Cout3] bf16, scale_next / shift_next [Cout_next] float32 (NULL = 1 / 0),
 * y_next [B,H,W,Cout_next] bf16 (conv1 of the next bottleneck, nets/mobilenet_v2.py,
 * PoseEstimation.pyproj:32; absent).  y is written as before (the next block adds it back as
 * its residual) but not read again: each 32-channel chunk of a finished tile, rounded to bf16
 * as stored, is one K step of the next layer while it is still in registers.  Cout3 must be 256
 * and Cout_next 64; otherwise as pe_conv3x3_c64_pw_fwd_bf16.  Equals that call followed by
 * pe_conv2d_fwd_bf16(1x1, relu) on y bit for bit. */
int32_t pe_conv3x3_c64_pw_next_fwd_bf16(int32_t B, int32_t H, int32_t W, const void* x,
                                        const void* w, const float* scale, const float* shift,
                                        int32_t relu, const void* w3, int32_t Cout3,
                                        const float* scale3, const float* shift3,
                                        const void* residual, void* y, const void* w_next,
                                        int32_t Cout_next, const float* scale_next,
                                        const float* shift_next, void* y_next, pe_stream_t stream);

/* 3x3 stride-2 pad-1 stem on a 3-channel image (MobileNetV2's first conv; the reference's
 * backbone, nets/mobilenet_v2.py, PoseEstimation.pyproj:32, absent), computed over the same
 * space-to-depth image as a 2-row conv: w [Cout][2 rows][4 cols x 16] bf16 (columns 2, 3 zero),
 * x_s2d [B,H/2,W/2,16] -> y [B,H/2,W/2,Cout].  relu: 0 none, 1 ReLU, 2 ReLU6 (also accepted by
 * every conv entry above). */
int32_t pe_stem3x3s2_fwd_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout, int32_t relu,
                              const void* x_s2d, const void* w, const float* scale,
                              const float* shift, void* y, pe_stream_t stream);

/* Depthwise 3x3 conv, NHWC bf16, stride 1 or 2, dilation >= 1, pad = dilation:
 * y[b,oy,ox,c] = act(scale[c] * sum_{ky,kx} w[ky*3+kx][c] * x[b, oy*s+(ky-1)*d, ox*s+(kx-1)*d, c]
 *                    + shift[c]);  w is [9][C] bf16, scale/shift float32 [C] or NULL,
 * y [B, (H-1)/s+1, (W-1)/s+1, C].  C % 8 == 0.  Replaces the depthwise layers of the
 * reference's nets/mobilenet_v2.py (absent). */
int32_t pe_depthwise3x3_fwd_bf16(const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                                 int32_t stride, int32_t dilation, const void* w,
                                 const float* scale, const float* shift, int32_t relu, void* y,
                                 pe_stream_t stream);

/* Backward of the depthwise 3x3 conv (same geometry arguments: H, W are the INPUT size).
 *   bwd_data  : dx [B,H,W,C] bf16 from dy [B,Ho,Wo,C] bf16 and the forward weight w [9][C].
 *   bwd_weight: dw (float32 [9][C], caller zeroes it) += sum over cells of dy * shifted x, in
 *               two ordered stages through `workspace`
 *               (pe_depthwise3x3_bwd_weight_workspace_bytes(C)): bit-stable from run to run. */
int32_t pe_depthwise3x3_bwd_data_bf16(const void* dy, int32_t B, int32_t H, int32_t W, int32_t C,
                                      int32_t stride, int32_t dilation, const void* w, void* dx,
                                      pe_stream_t stream);
size_t pe_depthwise3x3_bwd_weight_workspace_bytes(int32_t C);
int32_t pe_depthwise3x3_bwd_weight_bf16(const void* x, const void* dy, int32_t B, int32_t H,
                                        int32_t W, int32_t C, int32_t stride, int32_t dilation,
                                        float* dw, void* workspace, size_t workspace_bytes,
                                        pe_stream_t stream);

/* uint8 RGB [B,H,W,3] -> (x/255 - mean[c]) / std[c] -> bf16 space-to-depth [B,H/2,W/2,16]. */
int32_t pe_preprocess_u8_s2d_bf16(const uint8_t* img, int32_t B, int32_t H, int32_t W,
                                  const float* mean3, const float* std3, void* x_s2d,
                                  pe_stream_t stream);

/* Input pipeline in front of the stem.  Replaces the resize / augmentation / normalisation of
 * the reference's tf.data map function (dataset/pose_data.py, PoseEstimation.pyproj:27; absent;
 * configuration printed at test.ipynb:995-1010: flip_left_right, random_crop, scale_range
 * [0.5, 2.0], image_resize [320, 320]).
 * img [B,Hs,Ws,3] uint8; params_device: DEVICE float32 [B][5] = (sx, sy, tx, ty, flip) with
 *   u = (xf + 0.5)*sx + tx - 0.5,  v = (y + 0.5)*sy + ty - 0.5,  xf = flip ? W-1-x : x
 * the source coordinates of output pixel (x, y), sampled bilinearly (outside the frame = 0),
 * then (value/255 - mean[c]) / std[c] (HOST arrays), written as the bf16 space-to-depth image
 * x_s2d [B,H/2,W/2,16] the stem entries read.  A plain resize is sx = Ws/W, sy = Hs/H, t = 0.
 * pe_transform_joints_f32 applies the inverse map to joints [B,K,3] (x, y, v):
 *   xo = (x + 0.5 - tx)/sx - 0.5 (mirrored to W-1-xo when flipped), likewise y; when flipped,
 *   output joint k is read from input joint flip_perm_host[k] (HOST array of K indices, NULL =
 *   identity), which exchanges left and right joints.  joints_out must not alias joints_in. */
int32_t pe_augment_u8_s2d_bf16(const uint8_t* img, int32_t B, int32_t Hs, int32_t Ws,
                               const float* params_device, int32_t H, int32_t W,
                               const float* mean3, const float* std3, void* x_s2d,
                               pe_stream_t stream);

int32_t pe_transform_joints_f32(const float* joints_in, int32_t B, int32_t K,
                                const float* params_device, int32_t W,
                                const int32_t* flip_perm_host, float* joints_out,
                                pe_stream_t stream);

/* The per-pixel ignore mask (`labels['masks']`, test.ipynb:949) under the same map: mask_in
 * [B,Hms,Wms] uint8 covers the SOURCE frame in cells of src_stride pixels, mask_out [B,Hm,Wm]
 * covers the output image in cells of `stride` pixels.  The centre of output cell (x, y),
 *   xc = (x + 0.5)*stride - 0.5 (mirrored to (Wm*stride - 1) - xc when flipped), yc likewise,
 * goes through the image map (u, v) and reads source cell (floor((u + 0.5)/src_stride),
 * floor((v + 0.5)/src_stride)); a centre outside the source frame gives `fill` (0..255).
 * float32 arithmetic in this order; mask_out must not alias mask_in. */
int32_t pe_augment_mask_u8(const uint8_t* mask_in, int32_t B, int32_t Hms, int32_t Wms,
                           float src_stride, const float* params_device, int32_t Hm, int32_t Wm,
                           float stride, int32_t fill, uint8_t* mask_out, pe_stream_t stream);

/* 3x3 stride-2 pad-1 max pool, NHWC bf16: [B,H,W,C] -> [B,H/2,W/2,C] (H, W even; C % 8 == 0).
 * idx (may be NULL; training passes it): uint8 [B,H/2,W/2,C], the window position ky*3+kx of
 * each output's first maximum in row-major scan order. */
int32_t pe_maxpool3x3s2_fwd_bf16(const void* x, int32_t B, int32_t H, int32_t W, int32_t C,
                                 void* y, uint8_t* idx, pe_stream_t stream);

/* Backward of the pool above: dx [B,H,W,C] from the forward pass's idx and dy [B,H/2,W/2,C];
 * a tie inside a window sends the gradient to its first maximum (row-major scan). */
int32_t pe_maxpool3x3s2_bwd_bf16(const uint8_t* idx, const void* dy, int32_t B, int32_t H,
                                 int32_t W, int32_t C, void* dx, pe_stream_t stream);

/* Backward-data.  d describes the FORWARD conv (stride 1 or 2); dy is [B,Ho,Wo,Cout] bf16;
 * w_t is the transposed weight [Cin][KH*KW][Cout] bf16 (see pe_transpose_weights_bf16);
 * dx [B,H,W,Cin] bf16 = conv_bwd_data(dy) (+ residual, same shape as dx, may be NULL). */
int32_t pe_conv2d_bwd_data_bf16(const pe_conv_desc* d, const void* dy, const void* w_t,
                                const void* residual, void* dx, pe_stream_t stream);

/* When the forward conv has 8, 16 or 32 OUTPUT channels, one 64-element K run of the
 * backward-data GEMM spans P = 64/Cout adjacent dy pixels and w_t is [Cin][nslots][P*Cout]:
 * filter tap (kh, kw) lives in sub-block sub_of_tap[kh*KW+kw] (Cout elements) of slot
 * slot_of_tap[kh*KW+kw]; sub-blocks no tap maps to must hold zeros.  This call returns that
 * mapping (host arrays of KH*KW entries) and the slot count for a filter geometry. */
int32_t pe_conv2d_bwd_data_narrow_layout(int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                         int32_t Cout, int32_t* slot_of_tap, int32_t* sub_of_tap,
                                         int32_t* nslots);

/* Backward-data of the 4x4/2 transposed conv: dy [B,2H,2W,Cout], w_t [Cin][16][Cout] where
 * slot = phase*4 + tap of the forward packing; dx [B,H,W,Cin]. */
int32_t pe_deconv4x4s2_bwd_data_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                     const void* dy, const void* w_t, void* dx,
                                     pe_stream_t stream);

/* Backward-weight: dw (float32, same layout as the packed forward weight) += x (*) dy.
 * The caller zeroes dw.  The pixel range is split over workgroups; each split writes a partial
 * dw into `workspace` (device memory, at least pe_wgrad_workspace_bytes(elements of the
 * largest dw passed)) and an ordered reduction folds them, so results are bit-stable from run
 * to run.  ldy = channels per pixel of dy (>= Cout rounded up to 8; lets a zero-padded
 * gradient of a narrow layer be used).  With 8, 16 or 32 input channels dw has the forward
 * conv's narrow packing [Cout][KH][ceil(KW/P)][P*Cin]; elements of a run that lie past the filter
 * width receive values that carry no parameter and the caller masks them off (the stems'
 * structural zeros behave the same way). */
size_t pe_wgrad_workspace_bytes(int64_t weight_elems);
int32_t pe_conv2d_bwd_weight_bf16(const pe_conv_desc* d, const void* x, const void* dy,
                                  int32_t ldy, float* dw, void* workspace,
                                  size_t workspace_bytes, pe_stream_t stream);

/* The weight gradients of up to PE_WGRAD_MAX_BATCH convs in one launch plus one fold launch
 * (model/base_model.py's backward, absent: TensorFlow ran a layer's filter gradient as one op;
 * at batch 64 a single layer's gradient is microseconds of work inside ~19 us of launch, fill,
 * drain and fold latency, so the layers whose gradients are due together share a launch).
 * Each job is what one pe_conv2d_bwd_weight_bf16 call takes; results differ from that call's
 * only in the order of the pixel splits' sum, which is fixed by the batch's shapes alone.
 * workspace: pe_wgrad_batched_workspace_bytes(sum over jobs of Cout*KH*KW*Cin) bytes. */
#define PE_WGRAD_MAX_BATCH 8
typedef struct pe_wgrad_job {
    pe_conv_desc desc;
    const void* x;      /* forward input  [B,H,W,Cin]   bf16 */
    const void* dy;     /* output gradient [B,Ho,Wo,ldy] bf16 */
    float* dw;          /* [Cout,KH,KW,Cin] float32, accumulated into */
    int32_t ldy;
    int32_t reserved;
} pe_wgrad_job;
size_t pe_wgrad_batched_workspace_bytes(int64_t total_weight_elems);
int32_t pe_conv2d_bwd_weight_batched_bf16(const pe_wgrad_job* jobs, int32_t njobs, void* workspace,
                                          size_t workspace_bytes, pe_stream_t stream);
/* The transposed conv's weight gradient: its four output parities are four gradients of equal
 * size and go out as one batched launch, so the workspace is
 * pe_wgrad_batched_workspace_bytes(16*Cout*Cin) bytes. */
int32_t pe_deconv4x4s2_bwd_weight_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                       int32_t Cout, const void* x, const void* dy,
                                       float* dw_phases, void* workspace,
                                       size_t workspace_bytes, pe_stream_t stream);
int32_t pe_stem7x7s2_bwd_weight_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                     const void* x_s2d, const void* dy, float* dw,
                                     void* workspace, size_t workspace_bytes,
                                     pe_stream_t stream);
/* Same for the 3x3/2 stem of pe_stem3x3s2_fwd_bf16 (dw [Cout][2][64]). */
int32_t pe_stem3x3s2_bwd_weight_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                     const void* x_s2d, const void* dy, float* dw,
                                     void* workspace, size_t workspace_bytes,
                                     pe_stream_t stream);

/* out[c][t0 + t][a] = in[a][t][c] (bf16), out rows are slots*apitch elements long.  Builds the
 * backward-data weight layouts from the forward packings. */
int32_t pe_transpose_weights_bf16(const void* in, int32_t A, int32_t T, int32_t C, void* out,
                                  int32_t slots, int32_t apitch, int32_t t0, pe_stream_t stream);

/* The same transpose for many weights in one launch.  jobs_device: DEVICE array of njobs
 * records; tile_begin is the running sum of T * ceil(A/64) * ceil(C/64) over the jobs before
 * it (ascending), total_tiles the sum over all jobs.  A job may move a window of the source's
 * taps (src_T, src_t0): out[c][t0 + t][a] = in[a][src_t0 + t][c], t < T. */
typedef struct pe_transpose_job {
    const void* in;   /* [A][T][C] bf16 */
    void* out;        /* [C][slots][apitch] bf16 */
    int32_t A, T, C, slots, apitch, t0;
    int32_t tile_begin;
    int16_t src_T;    /* taps per row of the source when the job moves only a window of them
                         (source is [A][src_T][C]); 0 = T */
    int16_t src_t0;   /* first source tap of the window */
} pe_transpose_job;

int32_t pe_transpose_weights_batched_bf16(const pe_transpose_job* jobs_device, int32_t njobs,
                                          int32_t total_tiles, pe_stream_t stream);

/* Loss-gradient hand-over: float32 [B][K][HW] -> bf16 [B*HW][Kpad] (zero padded), and
 * dbias[k] += sum over pixels (dbias may be NULL; the caller zeroes it). */
int32_t pe_final_grad_prepare_bf16(const float* g_nchw, int32_t B, int32_t K, int32_t HW,
                                   int32_t Kpad, void* out_nhwc, float* dbias,
                                   pe_stream_t stream);

/* a[i] *= b[i] (float32); used to keep structurally-zero packed weights at zero gradient. */
int32_t pe_mul_f32(float* a, const float* b, int64_t n, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * Batch normalisation in training mode over NHWC bf16 viewed as [M][C], M = B*H*W.
 * Replaces: the batch-norm layers of nets/mobilenet_v2.py (PoseEstimation.pyproj:32; absent).
 *   stats : mean/var (biased) of z per channel -> mean, rstd = 1/sqrt(var+eps),
 *           scale = gamma*rstd, shift = beta - mean*scale; running buffers (may be NULL)
 *           move by `momentum` towards (mean, unbiased var).
 *   apply : y = act( z*scale + shift (+ residual) ) in bf16; relu 0: none, 1: ReLU, 2: ReLU6.
 *   bwd   : g = da masked by (a > 0) (relu = 1) or (0 < a < 6) (relu = 2) when a != NULL;
 *           dbeta = sum g; dgamma = sum g*xhat;
 *           dz = gamma*rstd*(g - dbeta/M - xhat*dgamma/M); dmasked (may be NULL) receives g.
 *           For a layer without a residual the mask can be rebuilt from z instead of read: pass
 *           a = NULL and act_scale / act_shift = the scale / shift its forward apply used (the
 *           stored activation is recomputed with the same operations and rounding, so the
 *           result is bit-identical and one tensor read per pass is saved); both NULL with
 *           a = NULL means no activation.
 *           frozen != 0 (pe_bn_bwd_bf16, pe_bn_bwd_from_partials_bf16): mean and rstd are the
 *           layer's running statistics, constants of the step, so dz = gamma*rstd*g; dgamma and
 *           dbeta as above.
 *           sync (pe_bn_bwd_bf16, pe_bn_bwd_from_partials_bf16, pe_bn_finalize_apply_bf16,
 *           pe_bn_stats_apply_bf16): DEVICE buffer of pe_bn_sync_bytes() bytes, 128-byte aligned,
 *           prepared once with pe_bn_sync_init.  With it the per-channel finalize and the
 *           element-wise pass are ONE launch: the first C/4 workgroups fold the partial sums and
 *           publish the coefficients, the others wait for them inside the launch on an arrival
 *           counter in the buffer.  The counter only grows and the library keeps its value on
 *           the host, so calls that share a buffer must come from one host thread at a time, on
 *           one stream, and cannot be captured into a HIP graph (a replay would repeat the old
 *           starting value); NULL keeps the separate finalize launch.  The int32 at byte offset
 *           2048 counts waits that timed out (such a workgroup folds for itself; never expected).
 * C must be a multiple of 8, at most 2048.  workspace: pe_bn_workspace_bytes(M, C).
 * Reductions are ordered: results are bit-stable from run to run.
 * ---------------------------------------------------------------------- */
size_t pe_bn_workspace_bytes(int64_t M, int32_t C);
int32_t pe_bn_stats_bf16(const void* z, int64_t M, int32_t C, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* rstd, float* scale,
                         float* shift, void* workspace, pe_stream_t stream);
int32_t pe_bn_apply_bf16(const void* z, int64_t M, int32_t C, const float* scale,
                         const float* shift, const void* residual, int32_t relu, void* y,
                         pe_stream_t stream);
int32_t pe_bn_bwd_bf16(const void* da, const void* a, int32_t relu, const void* z, int64_t M, int32_t C,
                       const float* gamma, const float* mean, const float* rstd,
                       const float* act_scale, const float* act_shift, float* dgamma,
                       float* dbeta, void* dz, void* dmasked, void* workspace,
                       int32_t frozen, int32_t* sync, pe_stream_t stream);

/* Folded constants of batch-norm layers that run on their running statistics (inference, or
 * training with frozen statistics), for any number of layers in one launch:
 *   rstd = 1/sqrt(running_var + eps); scale = gamma*rstd; shift = beta - running_mean*scale;
 *   mean / rstd (may be NULL) receive running_mean / rstd: the rows pe_bn_bwd_bf16 reads.
 * jobs_device: DEVICE array of njobs records; every pointer in it is a device pointer to C floats.
 * Replaces the inference graph TensorFlow builds from the same variables (is_training false,
 * test.ipynb:1004; nets/mobilenet_v2.py, PoseEstimation.pyproj:32, absent). */
typedef struct pe_bn_fold_job {
    const float* gamma;
    const float* beta;
    const float* running_mean;
    const float* running_var;
    float* scale;
    float* shift;
    float* mean;
    float* rstd;
    int32_t C;
    int32_t reserved;
} pe_bn_fold_job;
int32_t pe_bn_fold_f32(const pe_bn_fold_job* jobs_device, int32_t njobs, float eps,
                       pe_stream_t stream);

/* Backward-data fused with the first half of the batch-norm backward of the layer BELOW: the
 * result is the gradient da of an activation a = act(z * act_scale + act_shift) (relu 1: ReLU,
 * 2: ReLU6; z [B,H,W,Cin] bf16 is that layer's pre-norm conv output, act_scale / act_shift the
 * float32 [Cin] rows its forward normalise pass used).  Instead of da the call stores
 * g = (0 < a < act_hi) ? da : 0, with a rebuilt from z by the same operations and rounding as
 * pe_bn_apply_bf16, and every GEMM tile writes one partial row {sum g, sum g*z}[Cin] into
 * bn_workspace (pe_bn_workspace_bytes(B*H*W, Cin); *rows_out = number of rows, host memory).
 * pe_bn_bwd_from_partials_bf16 then folds the rows in a fixed order and produces dz, dgamma and
 * dbeta: together they do what pe_conv2d_bwd_data_bf16 + pe_bn_bwd_bf16(act_scale, act_shift)
 * do, with one read of da and z and one launch less.  g equals the masked da bit for bit; the
 * sums differ from pe_bn_bwd_bf16's in summation order only.  No residual; Cin % 8 == 0; a
 * stride-2 conv needs every input parity to have a tap (3x3 yes, 1x1 no).
 * Replaces: the gradient graph TensorFlow derived for model/base_model.py, nets/mobilenet_v2.py
 * (PoseEstimation.pyproj:30-32, absent). */
int32_t pe_conv2d_bwd_data_bnmask_bf16(const pe_conv_desc* d, const void* dy, const void* w_t,
                                       const void* z, const float* act_scale,
                                       const float* act_shift, int32_t relu, void* g,
                                       void* bn_workspace, size_t workspace_bytes,
                                       int32_t* rows_out, pe_stream_t stream);
/* The same fusion for a RESIDUAL layer below: a = relu(bn(z) + identity) cannot be rebuilt from
 * z, so the mask is read from the stored activation `act` [B,H,W,Cin] bf16 (ReLU only), and the
 * value that is masked is the GEMM's result plus `residual` (NULL: none), i.e. the whole gradient
 * of `act` when this conv is the last of its consumers to be differentiated:
 *   g = act > 0 ? bf16(conv_bwd_data(dy) + residual) : 0,  partial rows {sum g, sum g*z}.
 * Replaces pe_conv2d_bwd_data_bf16(residual) + the reduction pass of pe_bn_bwd_bf16(a = act):
 * a reduction launch and three tensor reads less per residual block; g is bit-identical. */
int32_t pe_conv2d_bwd_data_resmask_bf16(const pe_conv_desc* d, const void* dy, const void* w_t,
                                        const void* residual, const void* act, const void* z,
                                        void* g, void* bn_workspace, size_t workspace_bytes,
                                        int32_t* rows_out, pe_stream_t stream);
int32_t pe_deconv4x4s2_bwd_data_bnmask_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                            int32_t Cout, const void* dy, const void* w_t,
                                            const void* z, const float* act_scale,
                                            const float* act_shift, int32_t relu, void* g,
                                            void* bn_workspace, size_t workspace_bytes,
                                            int32_t* rows_out, pe_stream_t stream);
/* g: the masked gradient and workspace: the partial rows one of the two calls above left
 * (the three coefficient rows of the element-wise pass go behind them). */
int32_t pe_bn_bwd_from_partials_bf16(const void* g, const void* z, int32_t rows, int64_t M,
                                     int32_t C, const float* gamma, const float* mean,
                                     const float* rstd, float* dgamma, float* dbeta, void* dz,
                                     void* workspace, int32_t frozen, int32_t* sync,
                                     pe_stream_t stream);

/* Forward convolutions that also leave the batch-norm statistics of their own output behind,
 * so the training forward pass does not read z again to reduce it.  Same operands as
 * pe_conv2d_fwd_bf16 / pe_deconv4x4s2_fwd_bf16 / pe_stem7x7s2_fwd_bf16 with no epilogue
 * (no scale, shift, residual or ReLU; bf16 NHWC output; Cout % 8 == 0).  Each GEMM tile writes
 * one partial row {sum z, sum z*z}[Cout] (of the bf16-rounded z it stored) into bn_workspace,
 * sized by pe_bn_workspace_bytes(M, Cout); *rows_out (host memory, written before the call
 * returns) receives the number of rows.  pe_bn_finalize_bf16 folds them in a fixed order and
 * does what the second half of pe_bn_stats_bf16 does.  Same layers replaced as above. */
int32_t pe_conv2d_fwd_bnstats_bf16(const pe_conv_desc* d, const void* x, const void* w, void* y,
                                   void* bn_workspace, size_t workspace_bytes, int32_t* rows_out,
                                   pe_stream_t stream);
int32_t pe_deconv4x4s2_fwd_bnstats_bf16(int32_t B, int32_t H, int32_t W, int32_t Cin,
                                        int32_t Cout, const void* x, const void* w_phases,
                                        void* y, void* bn_workspace, size_t workspace_bytes,
                                        int32_t* rows_out, pe_stream_t stream);
int32_t pe_stem7x7s2_fwd_bnstats_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                      const void* x_s2d, const void* w, void* y,
                                      void* bn_workspace, size_t workspace_bytes,
                                      int32_t* rows_out, pe_stream_t stream);
int32_t pe_stem3x3s2_fwd_bnstats_bf16(int32_t B, int32_t H2, int32_t W2, int32_t Cout,
                                      const void* x_s2d, const void* w, void* y,
                                      void* bn_workspace, size_t workspace_bytes,
                                      int32_t* rows_out, pe_stream_t stream);
size_t pe_bn_sync_bytes(void);
int32_t pe_bn_sync_init(int32_t* sync, pe_stream_t stream);
/* pe_bn_finalize_bf16 followed by pe_bn_apply_bf16 on z, in one launch (see `sync` above). */
int32_t pe_bn_finalize_apply_bf16(const void* workspace, int32_t rows, const void* z, int64_t M,
                                  int32_t C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var,
                                  float* mean, float* rstd, float* scale, float* shift,
                                  const void* residual, int32_t relu, void* y, int32_t* sync,
                                  pe_stream_t stream);
/* pe_bn_stats_bf16 followed by pe_bn_apply_bf16 in two launches (reduction, then finalize +
 * element-wise pass): for layers whose conv leaves no statistics behind (depthwise). */
int32_t pe_bn_stats_apply_bf16(const void* z, int64_t M, int32_t C, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* rstd, float* scale,
                               float* shift, const void* residual, int32_t relu, void* y,
                               void* workspace, int32_t* sync, pe_stream_t stream);
int32_t pe_bn_finalize_bf16(const void* workspace, int32_t rows, int64_t M, int32_t C,
                            const float* gamma, const float* beta, float eps, float momentum,
                            float* running_mean, float* running_var, float* mean, float* rstd,
                            float* scale, float* shift, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * Multi-resolution fusion (HRNet-style exchange units of the 384x288 configuration; the
 * reference itself names only nets/mobilenet_v2.py, PoseEstimation.pyproj:32):
 *   y = relu?( base + sum_i nearest_upsample(low_i, 2^log2_factor_i) )
 * base (may be NULL) and y are bf16 NHWC [B,H,W,C]; low_i are bf16 NHWC
 * [B, H >> f_i, W >> f_i, C].  low_host / log2_factor_host are HOST arrays of n_low <= 3
 * device pointers / exponents (1..5).  Summed in float32 in argument order, rounded once.
 * ---------------------------------------------------------------------- */
int32_t pe_fuse_upsample_add_bf16(const void* base, int32_t n_low, const void* const* low_host,
                                  const int32_t* log2_factor_host, int32_t B, int32_t H,
                                  int32_t W, int32_t C, int32_t relu, void* y, pe_stream_t stream);

/* Backward of the fusion.  dy, y (the forward output; NULL when the forward had no ReLU) and
 * dbase are bf16 NHWC [B,H,W,C]; g = dy where y > 0, else 0 (g = dy without a ReLU).
 * dbase (may be NULL; must be NULL when y is NULL, since then the gradient of base is dy
 * itself) receives g; dlow_host[i] receives the sum of g over each 2^f_i x 2^f_i block,
 * [B, H >> f_i, W >> f_i, C], added in float32 in row-major order and rounded once. */
int32_t pe_fuse_upsample_add_bwd_bf16(const void* dy, const void* y, int32_t n_low,
                                      void* const* dlow_host, const int32_t* log2_factor_host,
                                      int32_t B, int32_t H, int32_t W, int32_t C, void* dbase,
                                      pe_stream_t stream);

/* out = a + b over n bf16 elements (n a multiple of 8), summed in float32, rounded once: the
 * accumulation of gradients that reach one activation from two consumers. */
int32_t pe_add_bf16(const void* a, const void* b, int64_t n, void* out, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * Gradient exchange without a host framework (one communicator per process / GPU).
 * Replaces: Horovod's allreduce (test.ipynb:12).  RCCL is loaded on first use; the calls return
 * PE_E_UNSUPPORTED when it cannot be.
 *   pe_comm_unique_id  : one rank makes the 128-byte id and hands it to the others by
 *                        whatever channel launched them.
 *   pe_comm_create     : collective over `world` ranks, on the caller's current GPU.
 *   pe_allreduce_bucket: buf[0..n) = sum over ranks of (scale * buf), in place, asynchronous on
 *                        `stream` (dtype PE_DTYPE_F32 / PE_DTYPE_BF16).  scale = 1 is a plain
 *                        sum: the training step passes 1 and folds 1/world into the optimizer
 *                        (pe_adam_step grad_scale).  Any other scale rides inside the reduction
 *                        (RCCL's pre-multiplied sum; for bf16 the scale is rounded to bf16), so
 *                        no second pass reads the bucket.
 *   pe_grad_pack_bf16 / pe_grad_unpack_bf16: float32 <-> bf16 (round to nearest even) copies of
 *                        n elements, for exchanging a float32 gradient bucket in bf16 (half the
 *                        bytes on the links); 16-byte aligned pointers.
 * The Python layer uses torch.distributed (the same RCCL) by default; these are for consumers
 * that have no such layer.
 * ---------------------------------------------------------------------- */
typedef struct pe_comm pe_comm;
int32_t pe_comm_unique_id(uint8_t id[128]);
int32_t pe_comm_create(int32_t rank, int32_t world, const uint8_t id[128], pe_comm** out);
int32_t pe_allreduce_bucket(pe_comm* comm, void* buf, int64_t n, int32_t dtype, float scale,
                            pe_stream_t stream);
int32_t pe_comm_destroy(pe_comm* comm);
int32_t pe_grad_pack_bf16(const float* src, void* dst_bf16, int64_t n, pe_stream_t stream);
int32_t pe_grad_unpack_bf16(const void* src_bf16, float* dst, int64_t n, pe_stream_t stream);

/* ------------------------------------------------------------------------
 * Detection branch: per-anchor targets from ground-truth boxes, and the loss on them.
 * Replaces: utils/bboxes.py (PoseEstimation.pyproj:35; absent).  Visible in the reference: 1500 =
 * 20*20*3 + 10*10*3 anchors per image with `classes`, `weights`, `regs` labels
 * (test.ipynb:423-464, 504-507), one positive anchor per person (test.ipynb:538-541) and
 * `regs` rows that are corner boxes in [0, 1] (test.ipynb:571-572).
 *   anchors      [A,4] float32 (ymin, xmin, ymax, xmax), any order of anchors
 *   boxes        [B,G,4] float32 corner boxes; box_classes [B,G] int32, <= 0 marks an empty
 *                slot; G <= 64
 *   Each labelled box claims the anchor it overlaps most (lowest anchor on ties, lowest box on
 *   contested anchors).  With pos_iou <= 1, anchors whose best IoU >= pos_iou also take their
 *   best box.  Matched: class of the box, weight 1, regs = the box, match = box index.
 *   Unmatched: class 0, regs 0, match -1, weight 1, or 0 when the best IoU >= ignore_iou.
 * Results are exact (fixed float32 operation order).
 * ---------------------------------------------------------------------- */
int32_t pe_anchor_assign_f32(const float* anchors, int32_t A, const float* boxes,
                             const int32_t* box_classes, int32_t B, int32_t G, float pos_iou,
                             float ignore_iou, float* classes, float* weights, float* regs,
                             int32_t* match, pe_stream_t stream);

/* One person box per image from its labelled keypoints, for batches that carry keypoints but
 * no box annotation.  Replaces: the box the reference's dataset adapters read from the
 * annotation files (dataset/pose_data.py, PoseEstimation.pyproj:27; absent); its format is what
 * test.ipynb:571-572 prints: normalised corners (ymin, xmin, ymax, xmax) in [0, 1].
 *   joints [B,K,3] float32 (x, y in input pixels, v); a joint counts when v > 0.
 *   boxes  [B,1,4] = clamp((min y / H - margin, min x / W - margin, max y / H + margin,
 *                          max x / W + margin), 0, 1), all zeros when no joint is labelled;
 *   box_classes [B,1] int32 = 1, or 0 when no joint is labelled (an empty slot for
 *   pe_anchor_assign_f32).  Exact (fixed float32 operation order). */
int32_t pe_boxes_from_joints_f32(const float* joints, int32_t B, int32_t K, int32_t H, int32_t W,
                                 float margin, float* boxes, int32_t* box_classes,
                                 pe_stream_t stream);

/* Loss of the detection branch (model/base_model.py, PoseEstimation.pyproj:30; absent):
 *   cls = sum_a weights * BCEwithLogits(logit, classes > 0) / N
 *   reg = sum_{positive a} sum_4 huber_1(reg_pred - regs) / N,  N = max(1, positives in batch)
 *   loss4 = {cls + reg_weight*reg, cls, reg, positives}; grad_logit [B,A] and grad_reg [B,A,4]
 *   (either may be NULL) receive d(loss4[0]).  Sums run in a fixed order (bit-stable).
 * workspace: pe_box_loss_workspace_bytes(B). */
size_t pe_box_loss_workspace_bytes(int32_t B);
int32_t pe_box_loss_fwd_bwd(const float* logit, const float* reg_pred, const float* classes,
                            const float* weights, const float* regs, int32_t B, int32_t A,
                            float reg_weight, float* loss4, float* grad_logit, float* grad_reg,
                            void* workspace, pe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PE_API_H */
