// Loss / clip / optimizer kernels of the xBD train step:
//   ComboLoss{dice:1, focal:8} per output channel ..... xBD_code/losses.py:24-34, 95-126, 273-288; weights train.py:348-353
//   clip_grad_norm_(parameters, 0.999) BEFORE the step .. xBD_code/train.py:373
//   hand-rolled AdamW (eps added to sqrt(v) before the bias correction) ... xBD_code/adamw.py:37-86
// All of it is HBM-bound streaming work: two passes over the logits for the loss (the dice gradient needs the
// channel totals first), one pass over the gradient arena for the norm, one for the update.  Sums are carried
// in fp64 through a fixed two-stage tree, so results are deterministic.
#include "common.h"

namespace {

constexpr int CL_BLOCKS = 256;       // partial-sum workgroups per channel
constexpr float XEPS = 1e-6f;        // xBD_code/losses.py:12

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// partial[c][blk][4] = { sum s*t, sum s, sum t, sum focal } over the pixels this workgroup visits
__global__ __launch_bounds__(256) void combo_partial_kernel(const float* __restrict__ logits, const float* __restrict__ masks,
                                                            int B, int C, long HW, double* __restrict__ partial) {
    __shared__ double sh[16];
    const int c = blockIdx.y;
    double aI = 0, aS = 0, aT = 0, aF = 0;
    const long n = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i - b * HW;
        const long at = (b * C + c) * HW + p;
        const float s = sigmoidf_(logits[at]);
        const float t = masks[at];
        const float o = fminf(fmaxf(s, XEPS), 1.f - XEPS), tt = fminf(fmaxf(t, XEPS), 1.f - XEPS);
        const float pt = (1.f - tt) * (1.f - o) + tt * o;
        aI += (double)s * t; aS += s; aT += t;
        aF += -(double)((1.f - pt) * (1.f - pt)) * (double)logf(pt);
    }
    double* dst = partial + ((long)c * gridDim.x + blockIdx.x) * 4;
    double r;
    r = dh_block_sum_f64(aI, sh); if (threadIdx.x == 0) dst[0] = r;
    r = dh_block_sum_f64(aS, sh); if (threadIdx.x == 0) dst[1] = r;
    r = dh_block_sum_f64(aT, sh); if (threadIdx.x == 0) dst[2] = r;
    r = dh_block_sum_f64(aF, sh); if (threadIdx.x == 0) dst[3] = r;
}

// sums[c][4] totals, channel_loss[c] = dice + 8 focal, *loss = sum_c w_c channel_loss[c]
__global__ void combo_finalize_kernel(const double* __restrict__ partial, int C, int nblk, double npix,
                                      const float* __restrict__ weights, float dice_w, float focal_w,
                                      float* __restrict__ sums, float* __restrict__ channel_loss, float* __restrict__ loss) {
    __shared__ double tot[16][4];
    const int c = threadIdx.x >> 2, k = threadIdx.x & 3;
    if (c < C) {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += partial[((long)c * nblk + b) * 4 + k];
        tot[c][k] = a;
        sums[c * 4 + k] = (float)a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float total = 0.f;
        for (int cc = 0; cc < C; ++cc) {
            // float arithmetic in the reference's order: 1 - (2 I + eps) / (S + T + eps); focal mean
            const float I = (float)tot[cc][0], S = (float)tot[cc][1], T = (float)tot[cc][2];
            const float dice = 1.f - (2.f * I + XEPS) / (S + T + XEPS);
            const float focal = (float)(tot[cc][3] / npix);
            const float l = dice_w * dice + focal_w * focal;
            channel_loss[cc] = l;
            total += weights[cc] * l;
        }
        *loss = total;
    }
}

// dlogits = upstream * w_c * ( dice_w * d dice/ds + focal_w * d focal/ds ) * s (1 - s)
__global__ __launch_bounds__(256) void combo_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ masks,
                                                        const float* __restrict__ sums, const float* __restrict__ weights,
                                                        const float* __restrict__ upstream, float dice_w, float focal_w,
                                                        int B, int C, long HW, float* __restrict__ dlogits) {
    const long n = (long)B * C * HW;
    const float up = upstream ? *upstream : 1.f;
    const float inv_n = 1.f / (float)((double)B * (double)HW);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        const float I = sums[c * 4 + 0], U = sums[c * 4 + 1] + sums[c * 4 + 2] + XEPS;
        const float s = sigmoidf_(logits[i]);
        const float t = masks[i];
        const float ddice = -(2.f * t * U - (2.f * I + XEPS)) / (U * U);
        float dfocal = 0.f;
        if (s > XEPS && s < 1.f - XEPS) {          // clamp passes the gradient strictly inside only
            const float tt = fminf(fmaxf(t, XEPS), 1.f - XEPS);
            const float pt = (1.f - tt) * (1.f - s) + tt * s;
            const float q = 1.f - pt;
            dfocal = (2.f * q * logf(pt) - q * q / pt) * (2.f * tt - 1.f) * inv_n;
        }
        dlogits[i] = up * weights[c] * (dice_w * ddice + focal_w * dfocal) * s * (1.f - s);
    }
}

// ---- gradient norm + clip coefficient -----------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
    __shared__ double sh[16];
    double a = 0.0;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        a += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; a += (double)v * v; }
    const double r = dh_block_sum_f64(a, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}
// out[0] = total L2 norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ void clip_coef_kernel(const double* __restrict__ partial, int nblk, float max_norm, float* __restrict__ out) {
    __shared__ double sh[16];
    double a = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) a += partial[i];
    const double r = dh_block_sum_f64(a, sh);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(r);
        out[0] = norm;
        out[1] = fminf(max_norm / (norm + 1e-6f), 1.f);
    }
}

// xBD_code/adamw.py:66-84: m, v updates; denom = sqrt(v) + eps; step = lr sqrt(bc2) / bc1; decay w -= wd lr w first
__global__ void adamw_xbd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps, float wd,
                                 float step_size, const float* __restrict__ grad_scale_dev) {
    const float gs = grad_scale_dev ? *grad_scale_dev : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gs;
        const float mm = m[i] * beta1 + (1.f - beta1) * gr;
        const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        float w = p[i];
        if (wd != 0.f) w += (-wd * lr) * w;
        w += -step_size * (mm / (sqrtf(vv) + eps));
        p[i] = w;
    }
}

// HIP-graph form: hyper = [lr, beta1, beta2, eps, weight_decay, -, step_size (out)], step counter on the device
__global__ void adamw_xbd_tick_kernel(float* hyper, int* step) {
    const int s = *step + 1;
    *step = s;
    const double bc1 = 1.0 - pow((double)hyper[1], (double)s), bc2 = 1.0 - pow((double)hyper[2], (double)s);
    hyper[6] = (float)((double)hyper[0] * sqrt(bc2) / bc1);
}
__global__ void adamw_xbd_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, long n, const float* __restrict__ hyper,
                                     const float* __restrict__ grad_scale_dev) {
    const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], step_size = hyper[6];
    const float gs = grad_scale_dev ? *grad_scale_dev : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gs;
        const float mm = m[i] * beta1 + (1.f - beta1) * gr;
        const float vv = v[i] * beta2 + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        float w = p[i];
        if (wd != 0.f) w += (-wd * lr) * w;
        w += -step_size * (mm / (sqrtf(vv) + eps));
        p[i] = w;
    }
}

inline hipStream_t ST(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace

extern "C" long dh_combo_loss_workspace_size(int C) { return (long)C * CL_BLOCKS * 4 * sizeof(double); }

// logits, masks: [B][C][H*W] fp32 (NCHW, the reference's layout).  sums_out [C][4], channel_loss_out [C], loss_out [1].
extern "C" int dh_combo_loss_fwd(const float* logits, const float* masks, int B, int C, long HW, const float* weights_dev,
                                 float dice_weight, float focal_weight, float* sums_out, float* channel_loss_out,
                                 float* loss_out, void* workspace, void* stream) {
    DH_REQUIRE(C >= 1 && C <= 16, "combo_loss: C=%d must be in 1..16", C);
    DH_REQUIRE(B > 0 && HW > 0, "combo_loss: empty input");
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(combo_partial_kernel, dim3(CL_BLOCKS, C), dim3(256), 0, ST(stream), logits, masks, B, C, HW, partial);
    DH_CHECK_LAUNCH("combo_partial");
    hipLaunchKernelGGL(combo_finalize_kernel, dim3(1), dim3(64), 0, ST(stream), partial, C, CL_BLOCKS,
                       (double)B * (double)HW, weights_dev, dice_weight, focal_weight, sums_out, channel_loss_out, loss_out);
    DH_CHECK_LAUNCH("combo_finalize");
    return 0;
}

extern "C" int dh_combo_loss_bwd(const float* logits, const float* masks, const float* sums, const float* weights_dev,
                                 const float* upstream_dev, float dice_weight, float focal_weight, int B, int C, long HW,
                                 float* dlogits, void* stream) {
    const long n = (long)B * C * HW;
    long g = (n + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(combo_bwd_kernel, dim3((int)g), dim3(256), 0, ST(stream), logits, masks, sums, weights_dev,
                       upstream_dev, dice_weight, focal_weight, B, C, HW, dlogits);
    DH_CHECK_LAUNCH("combo_bwd");
    return 0;
}

extern "C" long dh_grad_norm_workspace_size(void) { return 1024 * sizeof(double); }

// out_dev[0] = ||grad||_2 over the flat arena, out_dev[1] = the clip_grad_norm_ coefficient for max_norm
extern "C" int dh_grad_norm_clip_coef(const float* grad, long n, float max_norm, float* out_dev, void* workspace, void* stream) {
    DH_REQUIRE((reinterpret_cast<size_t>(grad) & 15) == 0, "grad_norm: arena must be 16-byte aligned");
    double* partial = reinterpret_cast<double*>(workspace);
    long g = (n / 4 + 255) / 256;
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((int)g), dim3(256), 0, ST(stream), grad, n, partial);
    DH_CHECK_LAUNCH("sumsq_partial");
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, ST(stream), partial, (int)g, max_norm, out_dev);
    DH_CHECK_LAUNCH("clip_coef");
    return 0;
}

extern "C" int dh_adamw_xbd_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step,
                                 const float* grad_scale_dev, void* stream) {
    DH_REQUIRE(step >= 1, "adamw_xbd: step must be >= 1");
    if (n == 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(adamw_xbd_kernel, dim3((int)g), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, step_size, grad_scale_dev);
    DH_CHECK_LAUNCH("adamw_xbd");
    return 0;
}

extern "C" int dh_adamw_xbd_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                       float* hyper_dev, int* step_dev, const float* grad_scale_dev, void* stream) {
    if (n == 0) return 0;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(adamw_xbd_tick_kernel, dim3(1), dim3(1), 0, ST(stream), hyper_dev, step_dev);
    hipLaunchKernelGGL(adamw_xbd_dev_kernel, dim3((int)g), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq, n,
                       hyper_dev, grad_scale_dev);
    DH_CHECK_LAUNCH("adamw_xbd_graph");
    return 0;
}
