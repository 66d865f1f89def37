// Focal loss (forward + gradient in one pass), arg-max mask, fused multi-tensor AdamW, and the
// space-to-depth helpers that turn the 7x7/stride-2 stem into a 4x4/stride-1 MFMA convolution.
//
// Reference semantics restated:
//   focal_loss ..... models/losses.py:106-196 (alpha 0.5, gamma 2, one_hot + 1e-6, mean over B*H*W)
//   mask ........... torch.argmax(dim=1), first maximum wins (models/trainer.py:170, evaluator.py:101)
//   AdamW .......... torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
//                    (models/trainer.py:39-40), decoupled decay, bias-corrected, no amsgrad
//   stem ........... nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (models/resnet.py:150)
#include "common.h"

namespace {

constexpr int MAXC = 8;

// logits NCHW fp32 [B][C][HW]; target int64 [B][HW]; dlogits NCHW fp32 (already scaled by gscale/(B*HW))
// CT > 0: the class count as a compile-time constant (2: the change-detection nets, 5: xBD-style heads) -- the generic form
// walks MAXC = 8 predicated class slots per pixel and divides a 64-bit pixel index by HW (28 us for 2 M pixels, 1.7 TB/s);
// same operations in the same order, so the results are bit-identical.
template <int CT>
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                    int B, int Crt, long HW, float alpha, float gscale,
                                                    float* __restrict__ dlogits, float* __restrict__ partial) {
    __shared__ float red[256];
    const int C = CT > 0 ? CT : Crt;
    constexpr int NC = CT > 0 ? CT : MAXC;
    const long total = (long)B * HW;
    float lsum = 0.f;
    // (b, p) advance by a two-digit counter instead of i / HW, i % HW
    const long stride = (long)gridDim.x * blockDim.x;
    const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long b = i0 / HW, p = i0 - b * HW;
    const long db = stride / HW, dp = stride - db * HW;
    for (long i = i0; i < total; i += stride) {
        float z[NC], pr[NC], lp[NC];
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < C) { z[c] = logits[(b * C + c) * HW + p]; m = fmaxf(m, z[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < C) { pr[c] = expf(z[c] - m); s += pr[c]; }
        const float ls = logf(s), inv = 1.f / s;
        const int t = (int)target[i];
        float loss = 0.f, gsum = 0.f, gp[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < C) {
                pr[c] *= inv;
                lp[c] = z[c] - m - ls;
                const float oh = (c == t ? 1.f : 0.f) + 1e-6f;
                const float om = 1.f - pr[c];
                loss += oh * (-alpha * om * om * lp[c]);
                // d f_c / d p_c * p_c, f_c = -alpha (1-p)^2 log p
                gp[c] = oh * (-alpha) * (-2.f * om * pr[c] * lp[c] + om * om);
                gsum += gp[c];
            }
        lsum += loss;
        if (dlogits) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (c < C) dlogits[(b * C + c) * HW + p] = gscale * (gp[c] - pr[c] * gsum);
        }
        b += db; p += dp;
        if (p >= HW) { p -= HW; ++b; }
    }
    red[threadIdx.x] = lsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- cross entropy of the batch-size-1 branch (models/losses.py:9-26, trainer.py:260-261): class weights [1, 1],
// ignore_index, mean over the non-ignored pixels.  partial[blk] = {sum of -log p_t, count}
__global__ __launch_bounds__(256) void ce_partial_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                         int B, int C, long HW, int ignore, double* __restrict__ partial) {
    __shared__ double sh[16];
    double ls = 0.0, cnt = 0.0;
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)target[i];
        if (t == ignore) continue;
        const long b = i / HW, p = i % HW;
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, logits[(b * C + c) * HW + p]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(logits[(b * C + c) * HW + p] - m);
        ls += (double)(m + logf(s) - logits[(b * C + t) * HW + p]);
        cnt += 1.0;
    }
    double r = dh_block_sum_f64(ls, sh);
    if (threadIdx.x == 0) partial[2 * blockIdx.x] = r;
    r = dh_block_sum_f64(cnt, sh);
    if (threadIdx.x == 0) partial[2 * blockIdx.x + 1] = r;
}
// out[0] = mean loss, out[1] = number of contributing pixels
__global__ void ce_finalize_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ out) {
    __shared__ double sh[16];
    double ls = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) { ls += partial[2 * i]; cnt += partial[2 * i + 1]; }
    const double a = dh_block_sum_f64(ls, sh);
    const double n = dh_block_sum_f64(cnt, sh);
    if (threadIdx.x == 0) { out[0] = (float)(a / n); out[1] = (float)n; }      // 0/0 = nan, as torch
}
__global__ void ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target, int B, int C, long HW,
                              int ignore, const float* __restrict__ fwd_out, const float* __restrict__ upstream,
                              float* __restrict__ dlogits) {
    const long total = (long)B * HW;
    const float gs = (upstream ? *upstream : 1.f) / fwd_out[1];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i % HW;
        const int t = (int)target[i];
        if (t == ignore) {
            for (int c = 0; c < C; ++c) dlogits[(b * C + c) * HW + p] = 0.f;
            continue;
        }
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, logits[(b * C + c) * HW + p]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(logits[(b * C + c) * HW + p] - m);
        const float inv = 1.f / s;
        for (int c = 0; c < C; ++c)
            dlogits[(b * C + c) * HW + p] = gs * (expf(logits[(b * C + c) * HW + p] - m) * inv - (c == t ? 1.f : 0.f));
    }
}

// ---- the gradient-free dice term of trainer.py:256-259: smp DiceLoss(mode='binary') on the ARG-MAX mask
// (from_logits => sigmoid of the mask value, dims (0, 2), smooth 0, eps 1e-7, zeroed when the target is empty).
// partial[blk] = {sum p*t, sum (p + t), sum t}
__global__ __launch_bounds__(256) void dice_argmax_partial_kernel(const float* __restrict__ logits,
                                                                  const long long* __restrict__ target, int B, int C, long HW,
                                                                  double* __restrict__ partial) {
    __shared__ double sh[16];
    double aI = 0.0, aC = 0.0, aT = 0.0;
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i % HW;
        float best = logits[(b * C) * HW + p];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = logits[(b * C + c) * HW + p];
            if (v > best) { best = v; arg = c; }
        }
        const float pr = 1.f / (1.f + expf(-(float)arg));        // logsigmoid(mask).exp()
        const float t = (float)target[i];
        aI += (double)pr * t; aC += (double)pr + t; aT += t;
    }
    double r = dh_block_sum_f64(aI, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x] = r;
    r = dh_block_sum_f64(aC, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x + 1] = r;
    r = dh_block_sum_f64(aT, sh); if (threadIdx.x == 0) partial[3 * blockIdx.x + 2] = r;
}
__global__ void dice_argmax_finalize_kernel(const double* __restrict__ partial, int nblk, float eps, float* __restrict__ out) {
    __shared__ double sh[16];
    double aI = 0.0, aC = 0.0, aT = 0.0;
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) { aI += partial[3 * i]; aC += partial[3 * i + 1]; aT += partial[3 * i + 2]; }
    const double I = dh_block_sum_f64(aI, sh), Cd = dh_block_sum_f64(aC, sh), T = dh_block_sum_f64(aT, sh);
    if (threadIdx.x == 0) {
        const float card = fmaxf((float)Cd, eps);
        *out = T > 0.0 ? 1.f - 2.f * (float)I / card : 0.f;
    }
}

__global__ void argmax_nchw_kernel(const float* __restrict__ logits, long long* __restrict__ mask, int B, int C, long HW) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const long b = i / HW, p = i % HW;
    float best = logits[(b * C) * HW + p];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = logits[(b * C + c) * HW + p];
        if (v > best) { best = v; arg = c; }
    }
    mask[i] = arg;
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                             float wd, float bc1, float bc2_sqrt, float grad_scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * grad_scale;
        float w = p[i];
        w *= 1.f - lr * wd;
        const float mm = m[i] + (1.f - beta1) * (gr - m[i]);
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        w -= (lr / bc1) * (mm / denom);
        p[i] = w;
    }
}

// x NCHW fp32 [N][3][H][W] -> space-to-depth NHWC T [N][H/2][W/2][CP], channel (ry*2+rx)*3 + c.
// One lane writes one 16-byte piece (8 bf16 / 4 fp32 channels); the 12 real channels are gathered from the three
// colour planes, the padding channels are zeros.
template <typename T>
__global__ void stem_s2d_kernel(const float* __restrict__ x, T* __restrict__ y, int N, int H, int W, int CP) {
    constexpr int V = V16<T>::N;
    const int H2 = H / 2, W2 = W / 2, vn = CP / V;
    const long total = (long)N * H2 * W2 * vn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch0 = (int)(i % vn) * V;
        long t = i / vn;
        const int x2 = (int)(t % W2); t /= W2;
        const int y2 = (int)(t % H2);
        const long n = t / H2;
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int ch = ch0 + j;
            v[j] = 0.f;
            if (ch < 12) {
                const int c = ch % 3, r = ch / 3, ry = r >> 1, rx = r & 1;
                v[j] = x[((n * 3 + c) * H + 2 * y2 + ry) * W + 2 * x2 + rx];
            }
        }
        stv(y + i * V, v);
    }
}
// w OIHW fp32 [O][3][7][7] -> packed [16 taps (dy,dx)][O][CP] T, kh = 2*dy + ry - 1, kw = 2*dx + rx - 1
template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ w, const float* __restrict__ oscale, T* __restrict__ out, int O,
                                 int CP) {
    const long total = 16L * O * CP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % CP);
        const int o = (int)((i / CP) % O);
        const int tap = (int)(i / ((long)CP * O));
        float v = 0.f;
        if (ch < 12) {
            const int c = ch % 3, r = ch / 3, ry = r >> 1, rx = r & 1;
            const int kh = 2 * (tap / 4) + ry - 1, kw = 2 * (tap % 4) + rx - 1;
            if (kh >= 0 && kh < 7 && kw >= 0 && kw < 7) v = w[((o * 3 + c) * 7 + kh) * 7 + kw] * (oscale ? oscale[o] : 1.f);
        }
        stf(out + i, v);
    }
}
// dw2 OIHW-of-the-4x4 form [O][CP][4][4] fp32 -> dw [O][3][7][7] (+)=
__global__ void stem_unpack_grad_kernel(const float* __restrict__ dw2, float* __restrict__ dw, int O, int CP,
                                        int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O * 147) return;
    const int kw = i % 7, kh = (i / 7) % 7, c = (i / 49) % 3, o = i / 147;
    const int dy = (kh + 1) / 2, ry = (kh + 1) % 2, dx = (kw + 1) / 2, rx = (kw + 1) % 2;
    const int ch = (ry * 2 + rx) * 3 + c;
    const float v = dw2[((long)(o * CP + ch) * 4 + dy) * 4 + dx];
    if (accumulate) dw[i] += v; else dw[i] = v;
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
extern "C" int dh_reduce_partials(const float* partial, long nt, long n, float scale, float* out, int accumulate,
                                  void* stream);

// workspace: 1024 floats.  loss_out: device scalar.  dlogits may be null (evaluation).
extern "C" int dh_focal_loss(const float* logits_nchw, const long long* target, int B, int C, long HW, float alpha,
                             float grad_scale, float* loss_out, float* dlogits_nchw, void* workspace, void* stream) {
    DH_REQUIRE(C >= 1 && C <= MAXC, "focal_loss: n_class=%d unsupported (max %d)", C, MAXC);
    const long total = (long)B * HW;
    long g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    float* partial = reinterpret_cast<float*>(workspace);
    auto kern = C == 2 ? focal_kernel<2> : (C == 5 ? focal_kernel<5> : focal_kernel<0>);
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(256), 0, ST(stream), logits_nchw, target, B, C, HW, alpha,
                       grad_scale / (float)total, dlogits_nchw, partial);
    DH_CHECK_LAUNCH("focal_loss");
    return dh_reduce_partials(partial, g, 1, 1.0f / (float)total, loss_out, 0, stream);
}

// workspace: 2048 doubles.  out_dev[0] = loss, out_dev[1] = contributing pixel count (needed by the backward)
extern "C" int dh_cross_entropy_fwd(const float* logits_nchw, const long long* target, int B, int C, long HW,
                                    int ignore_index, float* out_dev, void* workspace, void* stream) {
    DH_REQUIRE(C >= 1 && B > 0 && HW > 0, "cross_entropy: empty input");
    const long total = (long)B * HW;
    long g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(ce_partial_kernel, dim3((int)g), dim3(256), 0, ST(stream), logits_nchw, target, B, C, HW,
                       ignore_index, partial);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), partial, (int)g, out_dev);
    DH_CHECK_LAUNCH("cross_entropy_fwd");
    return 0;
}
extern "C" int dh_cross_entropy_bwd(const float* logits_nchw, const long long* target, int B, int C, long HW,
                                    int ignore_index, const float* fwd_out_dev, const float* upstream_dev,
                                    float* dlogits_nchw, void* stream) {
    const long total = (long)B * HW;
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(ce_bwd_kernel, dim3((int)g), dim3(256), 0, ST(stream), logits_nchw, target, B, C, HW, ignore_index,
                       fwd_out_dev, upstream_dev, dlogits_nchw);
    DH_CHECK_LAUNCH("cross_entropy_bwd");
    return 0;
}
// workspace: 3072 doubles
extern "C" int dh_dice_argmax_constant(const float* logits_nchw, const long long* target, int B, int C, long HW, float eps,
                                       float* loss_out, void* workspace, void* stream) {
    const long total = (long)B * HW;
    long g = (total + 255) / 256;
    if (g > 1024) g = 1024;
    double* partial = reinterpret_cast<double*>(workspace);
    hipLaunchKernelGGL(dice_argmax_partial_kernel, dim3((int)g), dim3(256), 0, ST(stream), logits_nchw, target, B, C, HW,
                       partial);
    hipLaunchKernelGGL(dice_argmax_finalize_kernel, dim3(1), dim3(256), 0, ST(stream), partial, (int)g, eps, loss_out);
    DH_CHECK_LAUNCH("dice_argmax_constant");
    return 0;
}

extern "C" int dh_argmax_nchw(const float* logits_nchw, long long* mask, int B, int C, long HW, void* stream) {
    hipLaunchKernelGGL(argmax_nchw_kernel, dim3(dh_cdiv((long)B * HW, 256)), dim3(256), 0, ST(stream), logits_nchw,
                       mask, B, C, HW);
    DH_CHECK_LAUNCH("argmax");
    return 0;
}

// one launch over the flat fp32 arenas (param / grad / exp_avg / exp_avg_sq); step >= 1
extern "C" int dh_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* stream) {
    DH_REQUIRE(step >= 1, "adamw: step must be >= 1");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (n == 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)g), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    DH_CHECK_LAUNCH("adamw");
    return 0;
}

extern "C" int dh_stem_space_to_depth(int dtype, const float* x_nchw, void* y, int N, int H, int W, int CP,
                                      void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(H % 2 == 0 && W % 2 == 0 && CP >= 12 && CP % V == 0, "stem_s2d: H, W must be even and CP >= 12 (multiple of %d)", V);
    const long n = (long)N * (H / 2) * (W / 2) * (CP / V);
    long g = (n + 255) / 256;
    if (g > 8192) g = 8192;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(stem_s2d_kernel<bf16>, dim3((int)g), dim3(256), 0, ST(stream), x_nchw, (bf16*)y, N, H, W, CP);
    else hipLaunchKernelGGL(stem_s2d_kernel<float>, dim3((int)g), dim3(256), 0, ST(stream), x_nchw, (float*)y, N, H, W, CP);
    DH_CHECK_LAUNCH("stem_s2d");
    return 0;
}
extern "C" int dh_stem_pack_weight(int dtype, const float* w_oihw, const float* out_scale, void* packed, int O, int CP, void* stream) {
    const long n = 16L * O * CP;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(stem_pack_kernel<bf16>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), w_oihw, out_scale, (bf16*)packed, O, CP);
    else hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), w_oihw, out_scale, (float*)packed, O, CP);
    DH_CHECK_LAUNCH("stem_pack");
    return 0;
}
extern "C" int dh_stem_unpack_grad(const float* dw2, float* dw_oihw, int O, int CP, int accumulate, void* stream) {
    hipLaunchKernelGGL(stem_unpack_grad_kernel, dim3(dh_cdiv(O * 147, 256)), dim3(256), 0, ST(stream), dw2, dw_oihw, O,
                       CP, accumulate);
    DH_CHECK_LAUNCH("stem_unpack_grad");
    return 0;
}

// ---- graph-capturable AdamW: hyper-parameters and the step counter live in device memory -------------
// hyper: [lr, beta1, beta2, eps, weight_decay, grad_scale, bc1 (out), bc2_sqrt (out)]
__global__ void adamw_tick_kernel(float* hyper, int* step) {
    const int s = *step + 1;
    *step = s;
    hyper[6] = (float)(1.0 - pow((double)hyper[1], (double)s));
    hyper[7] = (float)sqrt(1.0 - pow((double)hyper[2], (double)s));
}
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, const float* __restrict__ hyper) {
    const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], gs = hyper[5];
    const float bc1 = hyper[6], bc2_sqrt = hyper[7];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gs;
        float w = p[i];
        w *= 1.f - lr * wd;
        const float mm = m[i] + (1.f - beta1) * (gr - m[i]);
        const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
        m[i] = mm;
        v[i] = vv;
        w -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        p[i] = w;
    }
}
extern "C" int dh_adamw_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                   float* hyper_dev, int* step_dev, void* stream) {
    if (n == 0) return 0;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, ST(stream), hyper_dev, step_dev);
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((int)g), dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq, n,
                       hyper_dev);
    DH_CHECK_LAUNCH("adamw_graph");
    return 0;
}

// ---- arg-max mask + confusion matrix in one pass (models/trainer.py:163-173, misc/metric_tool.py:141-158) ---------
// counts[gt * C + pred] += 1 over all pixels (int64, atomics on integers: order-independent, exact); mask optional
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                        int B, int C, long HW, long long* __restrict__ mask,
                                                        unsigned long long* __restrict__ counts) {
    __shared__ unsigned int hist[MAXC * MAXC];
    for (int i = threadIdx.x; i < C * C; i += 256) hist[i] = 0;
    __syncthreads();
    const long total = (long)B * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / HW, p = i % HW;
        float best = logits[(b * C) * HW + p];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = logits[(b * C + c) * HW + p];
            if (v > best) { best = v; arg = c; }
        }
        if (mask) mask[i] = arg;
        const int t = (int)target[i];
        if (t >= 0 && t < C) atomicAdd(&hist[t * C + arg], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += 256)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}
extern "C" int dh_confusion_matrix(const float* logits_nchw, const long long* target, int B, int C, long HW,
                                   long long* mask, long long* counts, void* stream) {
    DH_REQUIRE(C >= 1 && C <= MAXC, "confusion_matrix: n_class=%d unsupported (max %d)", C, MAXC);
    long g = ((long)B * HW + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(confusion_kernel, dim3((int)g), dim3(256), 0, ST(stream), logits_nchw, target, B, C, HW, mask,
                       reinterpret_cast<unsigned long long*>(counts));
    DH_CHECK_LAUNCH("confusion_matrix");
    return 0;
}

// dst[i] = src[i] * (*scalar_dev): chains an upstream scalar gradient without a host sync
__global__ void scale_by_scalar_kernel(const float* __restrict__ src, const float* __restrict__ scalar_dev,
                                       float* __restrict__ dst, long n) {
    const float s = *scalar_dev;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i] * s;
}
extern "C" int dh_scale_by_scalar(const float* src, const float* scalar_dev, float* dst, long n, void* stream) {
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (n == 0) return 0;
    hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((int)g), dim3(256), 0, ST(stream), src, scalar_dev, dst, n);
    DH_CHECK_LAUNCH("scale_by_scalar");
    return 0;
}

// ---- error channel ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" void dh_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* dh_last_error(void) { return g_err; }
extern "C" int dh_abi_version(void) { return 1; }
