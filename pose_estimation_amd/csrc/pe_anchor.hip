// Detection branch labels and loss: per-anchor class / weight / box targets from ground-truth
// boxes, and the objectness + box regression loss with its gradients.
//
// The reference builds these labels in utils/bboxes.py (PoseEstimation.pyproj:35; absent).  What
// its notebook shows: 1500 = 20*20*3 + 10*10*3 anchors per image, per-anchor `classes`,
// `weights`, `regs` (test.ipynb:423-464, 504-507), exactly one positive anchor per person
// (test.ipynb:538-541) and `regs` rows that are corner boxes in [0, 1] (test.ipynb:571-572).
//
// Both kernels are tiny next to the image path (A * G overlaps per image); they exist so that
// the labels are made where the images already are.  All float32 arithmetic is in a fixed order
// (the build disables contraction), so assignment results equal the CPU restatement bit for bit.
#include "pe_common.h"

namespace pe {

constexpr int kMaxBoxes = 64;        // ground-truth slots per image
constexpr int kAnchorsPerThread = 8; // anchors a thread keeps in registers per sweep

struct Box {
    float y0, x0, y1, x1;
};

__device__ __forceinline__ float box_iou(const Box& a, float area_a, const Box& g, float area_g) {
    const float iy0 = fmaxf(a.y0, g.y0), ix0 = fmaxf(a.x0, g.x0);
    const float iy1 = fminf(a.y1, g.y1), ix1 = fminf(a.x1, g.x1);
    const float ih = fmaxf(iy1 - iy0, 0.0f);
    const float iw = fmaxf(ix1 - ix0, 0.0f);
    const float inter = ih * iw;
    const float uni = (area_a + area_g) - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}

// One workgroup per image.  Sweep s covers anchors [s*2048, (s+1)*2048): thread t holds anchors
// s*2048 + i*256 + t.  Per box the best (overlap, anchor) is reduced across the wave with
// shuffles and across waves and sweeps through LDS, always preferring the lower anchor index on
// equal overlap.
__global__ __launch_bounds__(256) void anchor_assign_kernel(
    const float* __restrict__ anchors, int A, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_classes, int G, float pos_iou, float ignore_iou,
    float* __restrict__ classes, float* __restrict__ weights, float* __restrict__ regs,
    int32_t* __restrict__ match) {
    __shared__ Box gbox[kMaxBoxes];
    __shared__ float garea[kMaxBoxes];
    __shared__ int gcls[kMaxBoxes];
    __shared__ float wave_iou[4][kMaxBoxes];
    __shared__ int wave_a[4][kMaxBoxes];
    __shared__ float claim_iou[kMaxBoxes];
    __shared__ int claim_a[kMaxBoxes];

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < G) {
        const float* s = boxes + ((long)b * G + tid) * 4;
        const Box g{s[0], s[1], s[2], s[3]};
        gbox[tid] = g;
        garea[tid] = (g.y1 - g.y0) * (g.x1 - g.x0);
        gcls[tid] = box_classes[(long)b * G + tid];
        claim_iou[tid] = 0.0f;
        claim_a[tid] = -1;
    }
    __syncthreads();

    float* cl = classes + (long)b * A;
    float* wt = weights + (long)b * A;
    float* rg = regs + (long)b * A * 4;
    int32_t* mt = match + (long)b * A;
    const bool use_pos = pos_iou <= 1.0f;

    constexpr int SWEEP = 256 * kAnchorsPerThread;
    for (int a0 = 0; a0 < A; a0 += SWEEP) {
        Box an[kAnchorsPerThread];
        float area[kAnchorsPerThread], best[kAnchorsPerThread];
        int bestg[kAnchorsPerThread];
#pragma unroll
        for (int i = 0; i < kAnchorsPerThread; ++i) {
            const int a = a0 + i * 256 + tid;
            const float4 v = a < A ? reinterpret_cast<const float4*>(anchors)[a]
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            an[i] = Box{v.x, v.y, v.z, v.w};
            area[i] = (v.z - v.x) * (v.w - v.y);
            best[i] = 0.0f;
            bestg[i] = -1;
        }
        for (int g = 0; g < G; ++g) {
            if (gcls[g] <= 0) continue;  // uniform
            const Box gb = gbox[g];
            const float ga = garea[g];
            float top = 0.0f;
            int topa = -1;
#pragma unroll
            for (int i = 0; i < kAnchorsPerThread; ++i) {
                const int a = a0 + i * 256 + tid;
                const float v = a < A ? box_iou(an[i], area[i], gb, ga) : 0.0f;
                if (v > best[i]) {
                    best[i] = v;
                    bestg[i] = g;
                }
                if (v > top) {  // anchors ascend with i: the first maximum stays
                    top = v;
                    topa = a;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ov = __shfl_xor(top, off);
                const int oa = __shfl_xor(topa, off);
                if (ov > top || (ov == top && oa >= 0 && (topa < 0 || oa < topa))) {
                    top = ov;
                    topa = oa;
                }
            }
            if (lane == 0) {
                wave_iou[wid][g] = top;
                wave_a[wid][g] = topa;
            }
        }
        __syncthreads();
        if (tid < G && gcls[tid] > 0) {
            float top = claim_iou[tid];
            int topa = claim_a[tid];
            for (int w = 0; w < 4; ++w) {  // ascending waves = ascending anchors within a row of 256
                const float ov = wave_iou[w][tid];
                const int oa = wave_a[w][tid];
                if (ov > top || (ov == top && oa >= 0 && (topa < 0 || oa < topa))) {
                    top = ov;
                    topa = oa;
                }
            }
            claim_iou[tid] = top;
            claim_a[tid] = topa;
        }
        // this sweep's anchors: threshold rule (the claims are applied at the end)
#pragma unroll
        for (int i = 0; i < kAnchorsPerThread; ++i) {
            const int a = a0 + i * 256 + tid;
            if (a >= A) continue;
            int m = -1;
            float w = 1.0f;
            if (use_pos && bestg[i] >= 0 && best[i] >= pos_iou) m = bestg[i];
            else if (best[i] >= ignore_iou) w = 0.0f;
            mt[a] = m;
            wt[a] = w;
            cl[a] = m >= 0 ? (float)gcls[m] : 0.0f;
            const Box z{0.f, 0.f, 0.f, 0.f};
            const Box r = m >= 0 ? gbox[m] : z;
            reinterpret_cast<float4*>(rg)[a] = make_float4(r.y0, r.x0, r.y1, r.x1);
        }
        __syncthreads();
    }
    // every labelled box takes the anchor it overlaps most; the lowest box index wins a tie
    if (tid == 0) {
        for (int g = G - 1; g >= 0; --g) {
            const int a = claim_a[g];
            if (gcls[g] <= 0 || a < 0 || claim_iou[g] <= 0.0f) continue;
            mt[a] = g;
            wt[a] = 1.0f;
            cl[a] = (float)gcls[g];
            const Box r = gbox[g];
            reinterpret_cast<float4*>(rg)[a] = make_float4(r.y0, r.x0, r.y1, r.x1);
        }
    }
}

// ---- loss
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    // fixed-shape tree: lanes by shuffle, then the four waves in order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((red[0] + red[1]) + red[2]) + red[3];
}

__global__ __launch_bounds__(256) void box_count_kernel(const float* __restrict__ classes,
                                                        const float* __restrict__ weights, int A,
                                                        int32_t* __restrict__ npos) {
    __shared__ int red[4];
    const int b = blockIdx.x;
    int n = 0;
    for (int a = threadIdx.x; a < A; a += 256)
        n += (classes[(long)b * A + a] > 0.0f && weights[(long)b * A + a] > 0.0f) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) npos[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void box_loss_kernel(
    const float* __restrict__ logit, const float* __restrict__ reg_pred,
    const float* __restrict__ classes, const float* __restrict__ weights,
    const float* __restrict__ regs, int B, int A, float reg_weight,
    const int32_t* __restrict__ npos, float* __restrict__ partial, float* __restrict__ grad_logit,
    float* __restrict__ grad_reg) {
    __shared__ float red[4];
    __shared__ int total;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int i = 0; i < B; ++i) n += npos[i];
        total = n > 1 ? n : 1;
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)total;
    float s_cls = 0.0f, s_reg = 0.0f;
    for (int a = threadIdx.x; a < A; a += 256) {
        const long i = (long)b * A + a;
        const float x = logit[i];
        const float w = weights[i];
        const bool obj = classes[i] > 0.0f;
        const float t = obj ? 1.0f : 0.0f;
        const float e = expf(-fabsf(x));
        const float bce = fmaxf(x, 0.0f) - x * t + log1pf(e);
        s_cls += w * bce;
        // sigmoid from the same exponential: 1/(1+e) for x >= 0, e/(1+e) otherwise
        const float sg = (x >= 0.0f ? 1.0f : e) / (1.0f + e);
        if (grad_logit) grad_logit[i] = w * (sg - t) * inv_n;
        const bool pos = obj && w > 0.0f;
        const float4 pr = reinterpret_cast<const float4*>(reg_pred)[i];
        const float4 tg = reinterpret_cast<const float4*>(regs)[i];
        const float d[4] = {pr.x - tg.x, pr.y - tg.y, pr.z - tg.z, pr.w - tg.w};
        float gq[4];
        float h = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ad = fabsf(d[j]);
            h += ad < 1.0f ? 0.5f * d[j] * d[j] : ad - 0.5f;
            gq[j] = pos ? fminf(fmaxf(d[j], -1.0f), 1.0f) * (reg_weight * inv_n) : 0.0f;
        }
        if (pos) s_reg += h;
        if (grad_reg) reinterpret_cast<float4*>(grad_reg)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
    const float c = block_sum_256(s_cls, red);
    const float r = block_sum_256(s_reg, red);
    if (threadIdx.x == 0) {
        partial[2 * b] = c;
        partial[2 * b + 1] = r;
    }
}

__global__ __launch_bounds__(64) void box_loss_finish_kernel(const float* __restrict__ partial,
                                                             const int32_t* __restrict__ npos, int B,
                                                             float reg_weight,
                                                             float* __restrict__ loss) {
    if (threadIdx.x != 0) return;
    float c = 0.0f, r = 0.0f;
    int n = 0;
    for (int b = 0; b < B; ++b) {  // one image after another: the same bits every run
        c += partial[2 * b];
        r += partial[2 * b + 1];
        n += npos[b];
    }
    const float inv_n = 1.0f / (float)(n > 1 ? n : 1);
    c *= inv_n;
    r *= inv_n;
    loss[0] = c + reg_weight * r;
    loss[1] = c;
    loss[2] = r;
    loss[3] = (float)n;
}

// One person box per image from its labelled keypoints: one thread per image walks the K
// joints in order (K is a few dozen), fixed float32 operation order.
__global__ __launch_bounds__(64) void boxes_from_joints_kernel(const float* __restrict__ joints,
                                                               int B, int K, float H, float W,
                                                               float margin,
                                                               float* __restrict__ boxes,
                                                               int32_t* __restrict__ classes) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* j = joints + (long)b * K * 3;
    float xmin = 0.0f, xmax = 0.0f, ymin = 0.0f, ymax = 0.0f;
    bool any = false;
    for (int k = 0; k < K; ++k) {
        const float x = j[3 * k], y = j[3 * k + 1], v = j[3 * k + 2];
        if (!(v > 0.0f)) continue;
        if (!any) {
            xmin = xmax = x;
            ymin = ymax = y;
            any = true;
        } else {
            xmin = fminf(xmin, x);
            xmax = fmaxf(xmax, x);
            ymin = fminf(ymin, y);
            ymax = fmaxf(ymax, y);
        }
    }
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (any) {
        o[0] = ymin / H - margin;
        o[1] = xmin / W - margin;
        o[2] = ymax / H + margin;
        o[3] = xmax / W + margin;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = fminf(fmaxf(o[i], 0.0f), 1.0f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) boxes[(long)b * 4 + i] = o[i];
    classes[b] = any ? 1 : 0;
}

}  // namespace pe

using namespace pe;

extern "C" int32_t pe_boxes_from_joints_f32(const float* joints, int32_t B, int32_t K, int32_t H,
                                            int32_t W, float margin, float* boxes,
                                            int32_t* box_classes, pe_stream_t stream) {
    const char* what = "pe_boxes_from_joints_f32";
    PE_REQUIRE(B >= 0 && K > 0 && H > 0 && W > 0, PE_E_INVALID,
               "%s: need B >= 0 and K, H, W > 0", what);
    PE_REQUIRE(margin >= 0.0f && margin <= 1.0f, PE_E_INVALID, "%s: margin must be in [0, 1]", what);
    if (B == 0) return PE_OK;
    PE_REQUIRE(joints && boxes && box_classes, PE_E_INVALID, "%s: NULL pointer", what);
    pe::launch(boxes_from_joints_kernel, (unsigned)((B + 63) / 64), 64, 0, as_stream(stream),
               joints, B, K, (float)H, (float)W, margin, boxes, box_classes);
    return check_launch(what);
}

extern "C" int32_t pe_anchor_assign_f32(const float* anchors, int32_t A, const float* boxes,
                                        const int32_t* box_classes, int32_t B, int32_t G,
                                        float pos_iou, float ignore_iou, float* classes,
                                        float* weights, float* regs, int32_t* match,
                                        pe_stream_t stream) {
    const char* what = "pe_anchor_assign_f32";
    PE_REQUIRE(anchors && boxes && box_classes && classes && weights && regs && match, PE_E_INVALID,
               "%s: NULL pointer", what);
    PE_REQUIRE(A > 0 && B > 0 && G > 0, PE_E_INVALID, "%s: A, B, G must be positive", what);
    PE_REQUIRE(G <= kMaxBoxes, PE_E_UNSUPPORTED, "%s: at most %d boxes per image (got %d)", what,
               kMaxBoxes, G);
    PE_REQUIRE(aligned(anchors, 16) && aligned(regs, 16), PE_E_INVALID,
               "%s: anchors and regs must be 16-byte aligned", what);
    PE_REQUIRE(ignore_iou > 0.0f, PE_E_INVALID, "%s: ignore_iou must be positive", what);
    pe::launch(anchor_assign_kernel, (unsigned)B, 256, 0, as_stream(stream), anchors, A, boxes, box_classes, G, pos_iou, ignore_iou, classes, weights, regs, match);
    return check_launch(what);
}

extern "C" size_t pe_box_loss_workspace_bytes(int32_t B) {
    if (B <= 0) return 16;
    return (((size_t)B * 3 * sizeof(float)) + 15) & ~(size_t)15;
}

extern "C" int32_t pe_box_loss_fwd_bwd(const float* logit, const float* reg_pred,
                                       const float* classes, const float* weights,
                                       const float* regs, int32_t B, int32_t A, float reg_weight,
                                       float* loss4, float* grad_logit, float* grad_reg,
                                       void* workspace, pe_stream_t stream) {
    const char* what = "pe_box_loss_fwd_bwd";
    PE_REQUIRE(logit && reg_pred && classes && weights && regs && loss4 && workspace, PE_E_INVALID,
               "%s: NULL pointer", what);
    PE_REQUIRE(A > 0 && B > 0 && B <= 65536, PE_E_INVALID, "%s: need A > 0 and 0 < B <= 65536", what);
    PE_REQUIRE(aligned(reg_pred, 16) && aligned(regs, 16) && (!grad_reg || aligned(grad_reg, 16)) &&
                   aligned(workspace, 16),
               PE_E_INVALID, "%s: box tensors and workspace must be 16-byte aligned", what);
    hipStream_t st = as_stream(stream);
    int32_t* npos = static_cast<int32_t*>(workspace);
    float* partial = reinterpret_cast<float*>(npos + B);
    pe::launch(box_count_kernel, (unsigned)B, 256, 0, st, classes, weights, A, npos);
    int32_t rc = check_launch("pe_box_loss_fwd_bwd(count)");
    if (rc != PE_OK) return rc;
    pe::launch(box_loss_kernel, (unsigned)B, 256, 0, st, logit, reg_pred, classes, weights, regs, B, A,
                                                 reg_weight, npos, partial, grad_logit, grad_reg);
    rc = check_launch("pe_box_loss_fwd_bwd(loss)");
    if (rc != PE_OK) return rc;
    pe::launch(box_loss_finish_kernel, 1, 64, 0, st, partial, npos, B, reg_weight, loss4);
    return check_launch("pe_box_loss_fwd_bwd(finish)");
}
