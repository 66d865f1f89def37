// BatchNorm2d (train / eval, forward / backward) and LayerNorm(32) kernels, NHWC.
//
// BatchNorm follows torch.nn.BatchNorm2d as used by the reference (models/resnet.py:152,
// models/help_funcs.py:11): eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased
// for running_var.  The Siamese trunk is run once on the concatenated [A;B] batch; "groups"
// reproduces the reference's two separate forward_single calls (models/networks.py:360-361):
// statistics are per stream and the running buffers are updated stream A first, then B.
// All kernels are HBM-bound element-wise / reduction passes with 16-byte (fp32 x4 / bf16 x4)
// accesses; statistics are accumulated in fp32 per workgroup and combined in fp64.
#include "common.h"

namespace {

// ---- finalize train-mode statistics --------------------------------------------------------
// partial: [2][CP][ntiles] from the conv epilogue (tile order = image order): one channel's tiles are contiguous.
// One workgroup per channel, one WAVEFRONT per statistics group (the two temporal streams are reduced concurrently,
// not one after the other); thread 0 then applies the running-statistics updates in group order.
constexpr int BN_MAXG = 4;
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int ntiles, int CP, int C, int G,
                                   double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out,
                                   long long* __restrict__ nbt) {
    __shared__ double sm[BN_MAXG], sv[BN_MAXG], ps[4], pq[4];
    const int c = blockIdx.x;
    if (nbt && c == 0 && threadIdx.x == 0) *nbt += G;      // one forward_single per stream (models/networks.py:359-360)
    // 4 wavefronts: 4 / G of them share a group's tiles (G in {1, 2, 4}), two accumulator pairs per lane
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wpg = 4 / G;
    const int g = w / wpg, slice = w % wpg;
    const int tpg = ntiles / G;
    {
        double s = 0.0, q = 0.0, s2 = 0.0, q2 = 0.0;
        const float* p0 = partial + ((size_t)0 * CP + c) * ntiles + (size_t)g * tpg;
        const float* p1 = partial + ((size_t)1 * CP + c) * ntiles + (size_t)g * tpg;
        const int step = 64 * wpg;
        int t = slice * 64 + lane;
#pragma unroll 4
        for (; t + step < tpg; t += 2 * step) {      // (16 loads in flight: classifier.1 has 32 channels x 16 K tiles, 14.5 us at four)
            s += (double)p0[t]; q += (double)p1[t];
            s2 += (double)p0[t + step]; q2 += (double)p1[t + step];
        }
        if (t < tpg) { s += (double)p0[t]; q += (double)p1[t]; }
        s += s2; q += q2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (lane == 0) { ps[w] = s; pq[w] = q; }
        __syncthreads();
        if (lane == 0 && slice == 0) {
            s = 0.0; q = 0.0;
            for (int k = 0; k < wpg; ++k) { s += ps[g * wpg + k]; q += pq[g * wpg + k]; }
            const double mean = s / count;
            double var = q / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float sc = gamma[c] * invstd;
            mean_out[g * C + c] = (float)mean;
            invstd_out[g * C + c] = invstd;
            scale_out[g * C + c] = sc;
            shift_out[g * C + c] = beta[c] - (float)mean * sc;
            sm[g] = mean;
            sv[g] = count > 1.0 ? var * count / (count - 1.0) : var;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && running_mean) {
        float rm = running_mean[c], rv = running_var[c];
        for (int k = 0; k < G; ++k) {
            rm = (1.f - momentum) * rm + momentum * (float)sm[k];
            rv = (1.f - momentum) * rv + momentum * (float)sv[k];
        }
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// ---- y = act(x*scale[g][c] + shift[g][c] (+ res)) -------------------------------------------
// relu_bits (optional): byte i = the ReLU mask of the 16-byte piece i (V = 8 bf16 / 4 fp32 elements), bit j = (y[V i + j] > 0) --
// what the backward of a BatchNorm + residual + ReLU layer needs of `y` (one sixteenth of the tensor instead of the tensor, twice)
template <int V> __device__ __forceinline__ unsigned relu_mask_byte(const float (&v)[V]) {
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) m |= (v[j] > 0.f ? 1u : 0u) << j;
    return m;
}
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                const float* __restrict__ scale, const float* __restrict__ shift, long nvec,
                                int C, long group_vec, int act, unsigned char* __restrict__ relu_bits) {
    constexpr int V = V16<T>::N;        // one 16-byte piece per lane
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * V) % C);
        const int g = (int)(i / group_vec);
        float v[V];
        ldv(x + i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = v[j] * scale[g * C + c + j] + shift[g * C + c + j];
        if (res) {
            float r[V];
            ldv(res + i * V, r);
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] += r[j];
        }
        if (act == DH_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        stv(y + i * V, v);
        if (relu_bits) relu_bits[i] = (unsigned char)relu_mask_byte(v);
    }
}

// Same map when 256*V is a multiple of C (every power-of-two width up to 2048): a thread's channel piece is the same in
// every iteration, so its scale / shift are loaded ONCE (the generic kernel spends 2-7 coefficient loads per data
// load in the vector cache).  grid = (workgroups per group, groups); two independent pieces in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_hoist_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                             T* __restrict__ y, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int C, long group_vec,
                                                             int act, unsigned char* __restrict__ relu_bits) {
    constexpr int V = V16<T>::N;
    const int g = blockIdx.y;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)((i * V) % C);
    float sc[V], sh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = scale[g * C + c + j]; sh[j] = shift[g * C + c + j]; }
    const size_t base = (size_t)g * group_vec * V;
    x += base; y += base;
    if (res) res += base;
    if (relu_bits) relu_bits += (size_t)g * group_vec;
    const bool relu = act == DH_ACT_RELU;
    for (; i + stride < group_vec; i += 2 * stride) {
        float a[V], b[V], ra[V], rb[V];
        ldv(x + i * V, a);
        ldv(x + (i + stride) * V, b);
        if (res) { ldv(res + i * V, ra); ldv(res + (i + stride) * V, rb); }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            a[j] = a[j] * sc[j] + sh[j];
            b[j] = b[j] * sc[j] + sh[j];
            if (res) { a[j] += ra[j]; b[j] += rb[j]; }
            if (relu) { a[j] = fmaxf(a[j], 0.f); b[j] = fmaxf(b[j], 0.f); }
        }
        stv(y + i * V, a);
        stv(y + (i + stride) * V, b);
        if (relu_bits) { relu_bits[i] = (unsigned char)relu_mask_byte(a); relu_bits[i + stride] = (unsigned char)relu_mask_byte(b); }
    }
    if (i < group_vec) {
        float a[V], ra[V];
        ldv(x + i * V, a);
        if (res) ldv(res + i * V, ra);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            a[j] = a[j] * sc[j] + sh[j];
            if (res) a[j] += ra[j];
            if (relu) a[j] = fmaxf(a[j], 0.f);
        }
        stv(y + i * V, a);
        if (relu_bits) relu_bits[i] = (unsigned char)relu_mask_byte(a);
    }
}

// ---- backward, pass 1: per-workgroup partial sums of dy and dy*xhat -------------------------
// dy = dout * (out > 0) when `out` (the post-ReLU activation) is given.
// partial: [2][C][G*bpg]; a workgroup never straddles two groups.
template <typename T, int MASK>      // MASK: 0 none, 1 from `out`, 2 recomputed from x, 3 from the forward's mask BYTES (`out` = relu_bits)
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                            const T* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int C,
                                                            long pix_per_group, int bpg,
                                                            float* __restrict__ partial,
                                                            const float* __restrict__ mscale,
                                                            const float* __restrict__ mshift) {
    constexpr int V = V16<T>::N;               // channels per lane: one 16-byte piece
    __shared__ float red[2 * 256 * V];
    const int g = blockIdx.x / bpg, b = blockIdx.x % bpg;
    const int cvn = C / V;                     // vector columns
    const int cv = threadIdx.x % cvn, r0 = threadIdx.x / cvn, rstep = 256 / cvn;
    const long chunk = (pix_per_group + bpg - 1) / bpg;
    const long p0 = b * chunk, p1 = (p0 + chunk < pix_per_group) ? p0 + chunk : pix_per_group;
    float mu[V], is[V], s1[V], s2[V], ms[V], mh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        mu[j] = mean[g * C + cv * V + j]; is[j] = invstd[g * C + cv * V + j]; s1[j] = 0.f; s2[j] = 0.f;
        if (MASK == 2) { ms[j] = mscale[g * C + cv * V + j]; mh[j] = mshift[g * C + cv * V + j]; }
    }
    // two pixels' loads in flight per thread; the accumulation order (p ascending) is unchanged
    auto fetch = [&](long p, float (&d)[V], float (&xv)[V], float (&o)[V]) {
        const size_t off = ((size_t)g * pix_per_group + p) * C + cv * V;
        ldv(dout + off, d);
        ldv(x + off, xv);
        if (MASK == 1) ldv(out + off, o);
        if (MASK == 3) o[0] = __uint_as_float((unsigned)reinterpret_cast<const unsigned char*>(out)[off / V]);
    };
    auto accum = [&](float (&d)[V], float (&xv)[V], float (&o)[V]) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (MASK == 1) d[j] = o[j] > 0.f ? d[j] : 0.f;
            if (MASK == 3) d[j] = ((__float_as_uint(o[0]) >> j) & 1u) ? d[j] : 0.f;
            // MASK 2: ReLU mask recomputed from the pre-normalisation input (layers without a residual): the same
            // x * scale + shift the forward evaluated, so one tensor read less in each backward pass
            if (MASK == 2) d[j] = (xv[j] * ms[j] + mh[j]) > 0.f ? d[j] : 0.f;
            s1[j] += d[j];
            s2[j] += d[j] * (xv[j] - mu[j]) * is[j];
        }
    };
    long p = p0 + r0;
    for (; p + rstep < p1; p += 2 * rstep) {
        float d0[V], x0[V], o0[V], d1[V], x1[V], o1[V];
        fetch(p, d0, x0, o0);
        fetch(p + rstep, d1, x1, o1);
        accum(d0, x0, o0);
        accum(d1, x1, o1);
    }
    if (p < p1) {
        float d0[V], x0[V], o0[V];
        fetch(p, d0, x0, o0);
        accum(d0, x0, o0);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
        red[(0 * 256 + threadIdx.x) * V + j] = s1[j];
        red[(1 * 256 + threadIdx.x) * V + j] = s2[j];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * C; o += 256) {
        const int which = o / C, c = o % C;
        float t = 0.f;
        for (int r = 0; r < rstep; ++r) t += red[(which * 256 + r * cvn + c / V) * V + (c % V)];
        partial[((size_t)which * C + c) * gridDim.x + blockIdx.x] = t;      // [2][C][G*bpg]
    }
}

// combine partials: per-group sums (for dx) and total dgamma / dbeta (accumulated or assigned)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partial, int bpg, int G, int C,
                                       float* __restrict__ sums /*[G][2][C]*/, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate) {
    __shared__ double t1[BN_MAXG], t2[BN_MAXG];
    const int c = blockIdx.x, lane = threadIdx.x & 63, g = threadIdx.x >> 6;      // blockDim = 64 * G
    double s1 = 0.0, s2 = 0.0;
    const float* p0 = partial + ((size_t)0 * C + c) * G * bpg + (size_t)g * bpg;
    const float* p1 = partial + ((size_t)1 * C + c) * G * bpg + (size_t)g * bpg;
#pragma unroll 8
    for (int t = lane; t < bpg; t += 64) { s1 += (double)p0[t]; s2 += (double)p1[t]; }      // (loads in flight; adds in order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) {
        sums[(g * 2 + 0) * C + c] = (float)s1;
        sums[(g * 2 + 1) * C + c] = (float)s2;
        t1[g] = s1; t2[g] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tb = 0.0, tg = 0.0;
        for (int k = 0; k < G; ++k) { tb += t1[k]; tg += t2[k]; }
        if (accumulate) { dgamma[c] += (float)tg; dbeta[c] += (float)tb; }
        else { dgamma[c] = (float)tg; dbeta[c] = (float)tb; }
    }
}

// ---- backward, pass 2: dx = gamma*invstd*(dy - (s1 + xhat*s2)/M); optional dres = dy --------
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ sums, float inv_m,
                                    long nvec, int C, long group_vec, T* __restrict__ dx, T* __restrict__ dres,
                                    const float* __restrict__ mscale, const float* __restrict__ mshift, int out_is_bits) {
    constexpr int V = V16<T>::N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)((i * V) % C);
        const int g = (int)(i / group_vec);
        float d[V], xv[V], r[V];
        ldv(dout + i * V, d);
        ldv(x + i * V, xv);
        if (out && out_is_bits) {
            const unsigned m = reinterpret_cast<const unsigned char*>(out)[i];
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] = ((m >> j) & 1u) ? d[j] : 0.f;
        } else if (out) {
            float o[V];
            ldv(out + i * V, o);
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
        } else if (mscale) {
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] = (xv[j] * mscale[g * C + c + j] + mshift[g * C + c + j]) > 0.f ? d[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float is = invstd[g * C + c + j];
            const float xh = (xv[j] - mean[g * C + c + j]) * is;
            r[j] = gamma[c + j] * is * (d[j] - (sums[(g * 2 + 0) * C + c + j] + xh * sums[(g * 2 + 1) * C + c + j]) * inv_m);
        }
        stv(dx + i * V, r);
        if (dres) stv(dres + i * V, d);
    }
}

// The same expression with the per-channel operands hoisted (see bn_apply_hoist_kernel); grid = (workgroups per group, groups)
template <typename T, int MASK>     // MASK: 0 none, 1 from `out` (post-ReLU activation), 2 recomputed from x, 3 from mask bytes (`out`)
__global__ __launch_bounds__(256) void bn_bwd_apply_hoist_kernel(const T* __restrict__ dout, const T* __restrict__ out,
                                                                 const T* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ sums, float inv_m, int C,
                                                                 long group_vec, T* __restrict__ dx, T* __restrict__ dres,
                                                                 const float* __restrict__ mscale,
                                                                 const float* __restrict__ mshift) {
    constexpr int V = V16<T>::N;
    const int g = blockIdx.y;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)((i * V) % C);
    float is[V], mu[V], ga[V], s1[V], s2[V], ms[V], mh[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
        is[j] = invstd[g * C + c + j]; mu[j] = mean[g * C + c + j]; ga[j] = gamma[c + j];
        s1[j] = sums[(g * 2 + 0) * C + c + j]; s2[j] = sums[(g * 2 + 1) * C + c + j];
        if (MASK == 2) { ms[j] = mscale[g * C + c + j]; mh[j] = mshift[g * C + c + j]; }
    }
    const size_t base = (size_t)g * group_vec * V;
    dout += base; x += base; dx += base;
    if (MASK == 1) out += base;
    const unsigned char* bits = reinterpret_cast<const unsigned char*>(out) + (size_t)g * group_vec;      // MASK == 3
    if (dres) dres += base;
    auto piece = [&](long k, float (&d)[V], float (&xv)[V], float (&o)[V]) {
        ldv(dout + k * V, d);
        ldv(x + k * V, xv);
        if (MASK == 1) ldv(out + k * V, o);
        if (MASK == 3) o[0] = __uint_as_float((unsigned)bits[k]);
    };
    auto finish = [&](long k, float (&d)[V], float (&xv)[V], float (&o)[V]) {
        float r[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (MASK == 1) d[j] = o[j] > 0.f ? d[j] : 0.f;
            if (MASK == 2) d[j] = (xv[j] * ms[j] + mh[j]) > 0.f ? d[j] : 0.f;
            if (MASK == 3) d[j] = ((__float_as_uint(o[0]) >> j) & 1u) ? d[j] : 0.f;
            const float xh = (xv[j] - mu[j]) * is[j];
            r[j] = ga[j] * is[j] * (d[j] - (s1[j] + xh * s2[j]) * inv_m);
        }
        stv(dx + k * V, r);
        if (dres) stv(dres + k * V, d);
    };
    for (; i + stride < group_vec; i += 2 * stride) {
        float d0[V], x0[V], o0[V], d1[V], x1[V], o1[V];
        piece(i, d0, x0, o0);
        piece(i + stride, d1, x1, o1);
        finish(i, d0, x0, o0);
        finish(i + stride, d1, x1, o1);
    }
    if (i < group_vec) {
        float d0[V], x0[V], o0[V];
        piece(i, d0, x0, o0);
        finish(i, d0, x0, o0);
    }
}

// eval-mode / frozen-statistics backward is not needed: the reference only trains in train mode.

// ---- LayerNorm over 32 channels: 8 lanes per row, 4 channels per lane -----------------------
template <typename T>
__global__ void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                              const float* __restrict__ beta, T* __restrict__ y, float* __restrict__ stats,
                              long rows, float eps) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid >> 3;
    const int q = (int)(gid & 7);
    const bool ok = row < rows;
    float v[4] = {0, 0, 0, 0};
    if (ok) ld4(x + row * 32 + q * 4, v);
    float s = v[0] + v[1] + v[2] + v[3];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.f / 32.f);
    float d[4], qq = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { d[j] = v[j] - mean; qq += d[j] * d[j]; }
    qq += __shfl_xor(qq, 1, 64); qq += __shfl_xor(qq, 2, 64); qq += __shfl_xor(qq, 4, 64);
    const float rstd = rsqrtf(qq * (1.f / 32.f) + eps);
    if (!ok) return;
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = d[j] * rstd * gamma[q * 4 + j] + beta[q * 4 + j];
    st4(y + row * 32 + q * 4, o);
    if (stats && q == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// dx (optionally accumulated into dx_acc) and per-workgroup partial dgamma/dbeta [nblk][2][32]
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, T* __restrict__ dx,
                                                     const T* __restrict__ dx_add, float* __restrict__ partial,
                                                     long rows) {
    __shared__ float red[2][32][33];
    const int q = threadIdx.x & 7, lr = threadIdx.x >> 3;
    float pg[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
    // grid-stride over groups of 32 rows: the number of partial rows stays <= gridDim.x
    for (long base = (long)blockIdx.x * 32; base < rows; base += (long)gridDim.x * 32) {
        const long row = base + lr;
        const bool ok = row < rows;
        float g[4] = {0, 0, 0, 0}, xv[4] = {0, 0, 0, 0};
        float mean = 0.f, rstd = 0.f;
        if (ok) {
            ld4(dy + row * 32 + q * 4, g);
            ld4(x + row * 32 + q * 4, xv);
            mean = stats[row * 2];
            rstd = stats[row * 2 + 1];
        }
        float xh[4], gh[4], a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[j] = (xv[j] - mean) * rstd;
            gh[j] = g[j] * gamma[q * 4 + j];
            a += gh[j];
            b += gh[j] * xh[j];
        }
        a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
        b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64); b += __shfl_xor(b, 4, 64);
        if (ok) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rstd * (gh[j] - (a + xh[j] * b) * (1.f / 32.f));
            if (dx_add) {
                float e[4];
                ld4(dx_add + row * 32 + q * 4, e);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] += e[j];
            }
            st4(dx + row * 32 + q * 4, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pg[j] += g[j] * xh[j]; pb[j] += g[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][lr][q * 4 + j] = pg[j];
        red[1][lr][q * 4 + j] = pb[j];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[which][r][c];
        partial[((size_t)blockIdx.x * 2 + which) * 32 + c] = t;
    }
}

// out[i] (+)= scale * sum_t partial[t][i], t < nt  (deterministic second reduction stage)
// 256 threads = 8 row phases x 32 consecutive outputs (coalesced rows), fp64 accumulation.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, long nt, long n,
                                                              float scale, float* __restrict__ out, int accumulate) {
    __shared__ double red[8][32];
    dh_reduce_partials_body(partial, nt, n, scale, out, accumulate, blockIdx.x, red);
}
inline void launch_reduce(const float* partial, long nt, long n, float scale, float* out, int accumulate,
                          hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, partial, nt, n,
                       scale, out, accumulate);
}

inline int ew_grid(long n, int block) {
    long g = (n + block - 1) / block;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
// workgroups per group of the hoisted element-wise BN kernels: ~2048 in total (8 per CU), >= 2 pieces per thread
inline int hoist_grid(long gvec, int groups) {
    long g = (gvec + 511) / 512, cap = 2048 / groups;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

template <typename T>
void launch_bn_bwd_apply(const void* dout, const void* out_relu, const void* x, const float* mean, const float* invstd,
                         const float* gamma, const float* sums, long ppg, long nvec, int C, int groups, void* dx, void* dres,
                         const float* mask_scale, const float* mask_shift, hipStream_t st, bool out_is_bits = false) {
    constexpr int V = V16<T>::N;
    const long gvec = nvec / groups;
    const float inv_m = 1.0f / (float)ppg;
    if ((256 * V) % C == 0) {
        const dim3 grid(hoist_grid(gvec, groups), groups);
#define DH_BWD_APPLY(M)                                                                                              \
        hipLaunchKernelGGL((bn_bwd_apply_hoist_kernel<T, M>), grid, dim3(256), 0, st, (const T*)dout, (const T*)out_relu, \
                           (const T*)x, mean, invstd, gamma, sums, inv_m, C, gvec, (T*)dx, (T*)dres, mask_scale, mask_shift)
        if (out_relu && out_is_bits) DH_BWD_APPLY(3);
        else if (out_relu) DH_BWD_APPLY(1);
        else if (mask_scale) DH_BWD_APPLY(2);
        else DH_BWD_APPLY(0);
#undef DH_BWD_APPLY
    } else {
        hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_grid(nvec, 256)), dim3(256), 0, st, (const T*)dout,
                           (const T*)out_relu, (const T*)x, mean, invstd, gamma, sums, inv_m, nvec, C, gvec, (T*)dx, (T*)dres,
                           mask_scale, mask_shift, out_is_bits ? 1 : 0);
    }
}

template <typename T>
void launch_bn_bwd_reduce(const void* dout, const void* out_relu, const void* x, const float* mean, const float* invstd, int C,
                          long ppg, int bpg, int groups, float* partial, const float* mask_scale, const float* mask_shift,
                          hipStream_t st, bool out_is_bits = false) {
#define DH_BWD_REDUCE(M)                                                                                                  \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M>), dim3(groups * bpg), dim3(256), 0, st, (const T*)dout, (const T*)out_relu, \
                       (const T*)x, mean, invstd, C, ppg, bpg, partial, mask_scale, mask_shift)
    if (out_relu && out_is_bits) DH_BWD_REDUCE(3);
    else if (out_relu) DH_BWD_REDUCE(1);
    else if (mask_scale) DH_BWD_REDUCE(2);
    else DH_BWD_REDUCE(0);
#undef DH_BWD_REDUCE
}

template <typename T>
void launch_bn_apply(const void* x, const void* residual, void* y, const float* scale, const float* shift, long nvec, int C,
                     int groups, int act, hipStream_t st, unsigned char* relu_bits = nullptr) {
    constexpr int V = V16<T>::N;
    const long gvec = nvec / groups;
    if ((256 * V) % C == 0)
        hipLaunchKernelGGL(bn_apply_hoist_kernel<T>, dim3(hoist_grid(gvec, groups), groups), dim3(256), 0, st, (const T*)x,
                           (const T*)residual, (T*)y, scale, shift, C, gvec, act, relu_bits);
    else
        hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(ew_grid(nvec, 256)), dim3(256), 0, st, (const T*)x, (const T*)residual,
                           (T*)y, scale, shift, nvec, C, gvec, act, relu_bits);
}


// ---- stem tail backward in one pass (bf16): 3x3/2 max-pool backward from the saved arg-max (the gather of
// maxpool_bwd_kernel, pointwise.hip) + the ReLU mask of relu(bn(y)) recomputed from y + the BatchNorm-backward reduction.
// Writes the masked gradient d and per-workgroup partial sums (sum d, sum d * xhat); the separate reduce pass (one more read of
// d and y, 268 MB at the bench size) disappears.  grid = G * bpg; a workgroup stays inside one group's images.
// extra (optional, [N][H][W][C]): a second gradient of the SAME pre-pool activation, added before the mask -- the hierarchical
// model taps the stem's output twice (max-pool -> layer1 and cat([a_128, b_128]) -> conv_layer2_0, models/networks.py:1118-1128,
// 1344): its sum with the pool gradient, the mask and the reduction were add + max-pool backward + two BatchNorm passes.
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const unsigned char* __restrict__ arg, const bf16* __restrict__ dpool,
                                                                 const bf16* __restrict__ extra,
                                                                 const bf16* __restrict__ y, const float* __restrict__ mscale,
                                                                 const float* __restrict__ mshift, bf16* __restrict__ d,
                                                                 float* __restrict__ partial, int npg, int H, int W, int C, int OH,
                                                                 int OW, int bpg) {
    constexpr int V = 8;
    __shared__ float red[2][256][V];
    const int g = blockIdx.x / bpg, b = blockIdx.x % bpg;
    const int vn = C / V, BH = (H + 1) / 2, BW = (W + 1) / 2;
    const int c = (threadIdx.x % vn) * V;            // 256 % vn == 0 and the loop stride is a multiple of vn: fixed per thread
    float ms[V], mh[V], s1[V], s2[V];        // s2 = sum d * y: the finalize turns it into sum d * xhat (in double)
#pragma unroll
    for (int j = 0; j < V; ++j) {
        ms[j] = mscale[g * C + c + j]; mh[j] = mshift[g * C + c + j];
        s1[j] = s2[j] = 0.f;
    }
    const long items = (long)npg * BH * BW * vn;
    for (long i = (long)b * 256 + threadIdx.x; i < items; i += (long)bpg * 256) {
        long t = i / vn;
        const int bx = (int)(t % BW); t /= BW;
        const int by = (int)(t % BH);
        const long n = (long)g * npg + t / BH;
        unsigned long long bits[2][2];
        float dv[2][2][V];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = by + wy, ox = bx + wx;
                bits[wy][wx] = ~0ull;
#pragma unroll
                for (int j = 0; j < V; ++j) dv[wy][wx][j] = 0.f;
                if (oy < OH && ox < OW) {
                    const long o = ((n * OH + oy) * OW + ox) * C + c;
                    bits[wy][wx] = *reinterpret_cast<const unsigned long long*>(arg + o);
                    ldv(dpool + o, dv[wy][wx]);
                }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                const int iy = 2 * by + a, ix = 2 * bx + bb;
                if (iy >= H || ix >= W) continue;
                float gs[V], yv[V];
#pragma unroll
                for (int j = 0; j < V; ++j) gs[j] = 0.f;
#pragma unroll
                for (int wy = 0; wy <= a; ++wy)
#pragma unroll
                    for (int wx = 0; wx <= bb; ++wx) {
                        const unsigned kk = (a + 1 - 2 * wy) * 3 + (bb + 1 - 2 * wx);
#pragma unroll
                        for (int j = 0; j < V; ++j)
                            if (((bits[wy][wx] >> (8 * j)) & 0xff) == kk) gs[j] += dv[wy][wx][j];
                    }
                const long o = ((n * H + iy) * W + ix) * C + c;
                ldv(y + o, yv);
                if (extra) {
                    float ev[V];
                    ldv(extra + o, ev);
#pragma unroll
                    for (int j = 0; j < V; ++j) gs[j] += ev[j];
                }
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float gm = (yv[j] * ms[j] + mh[j]) > 0.f ? gs[j] : 0.f;
                    gs[j] = gm;
                    s1[j] += gm;
                    s2[j] += gm * yv[j];
                }
                stv(d + o, gs);
            }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    if ((int)threadIdx.x < 2 * C) {
        const int which = threadIdx.x / C, ch = threadIdx.x % C, piece = ch / V, j = ch % V;
        float t = 0.f;
        for (int k = piece; k < 256; k += vn) t += red[which][k][j];
        partial[((size_t)which * C + ch) * gridDim.x + blockIdx.x] = t;       // [2][C][G * bpg]
    }
}
// bn_bwd_finalize_kernel + the per-channel coefficients of dx = A * d + B * y + Cc (= gamma*invstd*(d - (s1 + xhat*s2)/M))
// that the stem's weight gradient applies while it loads d and y (conv_wgrad.hip, DYT): coef [G][3][C].  partial[1] holds
// sum d * y (raw), converted here.
__device__ __forceinline__ void bn_bwd_finalize_coef_body(const int c, const float* __restrict__ partial, int bpg, int G, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, float inv_m, float* __restrict__ coef,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    __shared__ double t1[BN_MAXG], t2[BN_MAXG];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;      // wavefronts [0, G) take part (blockDim >= 64 * G)
    const bool act = g < G;
    double s1 = 0.0, s2 = 0.0;
    const float* p0 = partial + ((size_t)0 * C + c) * G * bpg + (size_t)(act ? g : 0) * bpg;
    const float* p1 = partial + ((size_t)1 * C + c) * G * bpg + (size_t)(act ? g : 0) * bpg;
#pragma unroll 8
    for (int t = lane; t < (act ? bpg : 0); t += 64) { s1 += (double)p0[t]; s2 += (double)p1[t]; }      // (loads in flight; adds in order)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0 && act) {
        s2 = (double)invstd[g * C + c] * (s2 - (double)mean[g * C + c] * s1);        // sum d * y -> sum d * xhat
        const float is = invstd[g * C + c], A = gamma[c] * is, B = -A * is * inv_m * (float)s2;
        coef[(g * 3 + 0) * C + c] = A;
        coef[(g * 3 + 1) * C + c] = B;
        coef[(g * 3 + 2) * C + c] = -A * inv_m * (float)s1 - B * mean[g * C + c];
        t1[g] = s1; t2[g] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tb = 0.0, tg = 0.0;
        for (int k = 0; k < G; ++k) { tb += t1[k]; tg += t2[k]; }
        if (accumulate) { dgamma[c] += (float)tg; dbeta[c] += (float)tb; }
        else { dgamma[c] = (float)tg; dbeta[c] = (float)tb; }
    }
}
__global__ void bn_bwd_finalize_coef_kernel(const float* __restrict__ partial, int bpg, int G, int C, const float* __restrict__ mean,
                                            const float* __restrict__ invstd, const float* __restrict__ gamma, float inv_m,
                                            float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                            int accumulate) {
    bn_bwd_finalize_coef_body(blockIdx.x, partial, bpg, G, C, mean, invstd, gamma, inv_m, coef, dgamma, dbeta, accumulate);
}


// ---- class head: data gradient of the 3x3 head convolution + the BatchNorm backward behind it, nothing in between ----------
// The classifier (TwoLayerConv2d, models/help_funcs.py:7-15) is conv0 -> BatchNorm -> ReLU -> conv1 (32 -> n_class).  The
// gradient of relu(BN(y)) with respect to its 32 channels is conv1's data gradient: K = (tap, class) = 18 products per element,
// i.e. ONE v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels from a 16-byte-per-pixel dlogits tensor (the form of
// head_dgrad3x3_mfma_kernel, pointwise.hip).  That is cheaper to RECOMPUTE than to keep: written once and read by the two
// passes of the BatchNorm backward it is 3 x 134 MB at the bench size.  So the two passes form it themselves:
//   pass 1 (APPLY = false): g = mask * (W1^T (*) dlogits); per-workgroup partial sums (sum g, sum g * y) -> [2][32][blocks]
//   (bn_bwd_finalize_coef_kernel: dgamma, dbeta and the coefficients of dx = A g + B y + C)
//   pass 2 (APPLY = true):  the same g again, dx = A g + B y + C written as bf16
// mask = (y * mscale + mshift > 0) recomputed from the pre-BatchNorm activation y, the only tensor either pass reads.
// An MFMA leaves a lane with channels 4 g + j of the 16-channel half s; one v_permlane16_swap per register pairs the halves of
// neighbouring lane groups so that a lane owns EIGHT consecutive channels of its pixel -- one 16-byte load of y, one 16-byte
// store of dx per lane, a wave instruction covers 1 KiB of consecutive addresses.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// dlogits [N][NC][H][W] fp32 (the loss kernel's layout) -> [N][H + 2][W + 2] pixels of NCP bf16 classes (NCP = 2: a pair, one word;
// NCP = 8 for 3 .. 8 classes: one 16-byte piece) inside a border of zeros: what head_bn_bwd_kernel gathers its nine taps from, one
// load per tap and no bounds test.  PL = 2 (the split-product fp32 mode): per pixel the piece of the values' bf16 heads, then
// the piece of their bf16 remainders.
template <int PL, int NCP>
__global__ void head_dlogits_pack_kernel(const float* __restrict__ src, unsigned* __restrict__ dst, int N, int NC, int H, int W) {
    const int Wp = W + 2, Hp = H + 2;
    constexpr int WPP = NCP / 2;            // words per plane and pixel
    const long total = (long)N * Hp * Wp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wp) - 1, yy = (int)((i / Wp) % Hp) - 1;
        const long n = i / ((long)Wp * Hp);
        const bool in = (unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H;
        unsigned hi[WPP], lo[WPP];
#pragma unroll
        for (int k = 0; k < WPP; ++k) {
            const float a = (in && 2 * k < NC) ? src[((n * NC + 2 * k) * H + yy) * (long)W + xx] : 0.f;
            const float b = (in && 2 * k + 1 < NC) ? src[((n * NC + 2 * k + 1) * H + yy) * (long)W + xx] : 0.f;
            hi[k] = f2bf2(a, b);
            lo[k] = f2bf2(a - __uint_as_float(hi[k] << 16), b - __uint_as_float(hi[k] & 0xffff0000u));
        }
        unsigned* o = dst + i * (WPP * PL);
#pragma unroll
        for (int k = 0; k < WPP; ++k) {
            o[k] = hi[k];
            if constexpr (PL == 2) o[WPP + k] = lo[k];
        }
    }
}

// WGRAD (pass 1 only): the head convolution's own weight gradient from the same loads.  dW1[ci][tap][class] = sum over pixels
// of relu(BN(y))[pixel][ci] * dlogits[pixel - tap offset][class] -- for the INPUT pixel a lane holds, the nine dlogits it has
// just gathered ARE those neighbours (k = (tap, class)).  Pixels are the K dimension of that product: both operands go through
// a wave-private LDS tile ([16 px][32] bf16, pitch 96 bytes; [16 px][96] at pitch 224 for NCP = 8) and come back transposed
// (ds_read_b64_tr_b16, four pixels of one channel / one k per lane) for v_mfma_f32_16x16x16_bf16; a wave's LDS operations
// execute in order, no barrier in the loop.  Per-workgroup partials [32 * 9 * NCP + NCP][blocks] (column ci * 9 NCP + tap * NCP +
// class, then the bias gradient = the column sums of dlogits): head_bwd_finalize_kernel.  What this replaces read y a third time
// (conv_wgrad_kernel, 54 us at the bench size).
constexpr int HB_TP = 96;                                // tile pitch (bytes): odd multiple of 32 -- conflict-free transpose reads
constexpr int hb_wcols(int ncp) { return 32 * 9 * ncp + ncp; }
constexpr int HB_WCOLS_MAX = hb_wcols(8);

// MODE 2 (dh_head_relu_bwd): no BatchNorm -- `y` is the OUTPUT of the ReLU in front of the head (classifier(conv_layer2(...)),
// models/networks.py:1351-1355), dx = (y > 0) * g in one pass, with the head's weight gradient (WGRAD) from the same loads.
// T = float (the bf16x3 mode of the fp32 pipeline, dh_set_f32_mma_mode != 0): y / dx are fp32 (32 bytes per lane), every matrix
// product is the three split products hi.hi + hi.lo + lo.hi of bf16 planes (x = hi + lo, ~2^-17: what that mode's backward
// convolutions and weight gradients compute): the weights' planes are formed once, the dlogits' planes come from the pair map
// (PL = 2), the planes of relu(BN(y)) are formed per group for the weight gradient's tiles.
// NCP = 8 (3 .. 8 classes: the five-class heads of the xBD nets, xBD_code/zoo/model_transformer_encoding.py): k = tap * 8 + class,
// three K steps of 32 -- lane group g of step st gathers the 16-byte piece of tap 4 st + g -- instead of one.
template <typename T, int MODE, bool WGRAD, int NCP = 2>
__global__ __launch_bounds__(256) void head_bn_bwd_kernel(const unsigned* __restrict__ dlp, const float* __restrict__ w_oihw, int N, int H,
                                                          int W, int NC, const T* __restrict__ y, const float* __restrict__ mscale,
                                                          const float* __restrict__ mshift, int groups, float* __restrict__ partial,
                                                          const float* __restrict__ coef, T* __restrict__ dx,
                                                          float* __restrict__ wpartial) {
    constexpr bool APPLY = MODE == 1, RELU = MODE == 2, STORE = MODE != 0, F32 = sizeof(T) == 4;
    constexpr int PL = F32 ? 2 : 1;  // operand planes
    constexpr int KST = NCP == 2 ? 1 : 3;                    // K steps (of 32) of the data-gradient product
    constexpr int NKB = 2 * KST;                             // 16-wide k blocks of the neighbourhood tile
    constexpr int TNP = NCP == 2 ? HB_TP : 224;              // ... and its pitch (bytes: an odd multiple of 32)
    constexpr int WCOLS = hb_wcols(NCP);
    constexpr int HB_DEPTH = 2;      // groups in flight per wave (depths 3 and 4 measured: no faster, more registers)
    constexpr int PXB = 32 * (int)sizeof(T);                 // bytes of a pixel of y / dx
    static_assert(!(APPLY && WGRAD), "the weight gradient rides on the reduction pass");
    static_assert(NCP == 2 || NCP == 8, "class pieces of 2 or 8");
    constexpr int TILE = 16 * HB_TP, TILEN = 16 * TNP;
    // (the four waves' partial weight gradients are parked over the tiles once the loop is done)
    constexpr int TBYTES = 4 * PL * (TILE + TILEN), RBYTES = 4 * WCOLS * 4;
    __shared__ __attribute__((aligned(16))) unsigned char hb_tiles[WGRAD ? (TBYTES > RBYTES ? TBYTES : RBYTES) : 16];
    const int lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    unsigned char* tH = hb_tiles + (WGRAD ? (threadIdx.x >> 6) * PL * (TILE + TILEN) : 0);      // [PL] tiles of the head's input
    unsigned char* tN = tH + (WGRAD ? PL * TILE : 0);                                           // [PL] tiles of the dlogits neighbourhood
    f32x4 wacc[2][NKB];
    float bsum[NCP];
#pragma unroll
    for (int c = 0; c < NCP; ++c) bsum[c] = 0.f;
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) wacc[a_][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A fragments (planes): row ci = s * 16 + pl, k = 32 st + 8 g + e -> NCP = 2: tap 4 g + e / 2, class e & 1; NCP = 8: tap 4 st + g, class e
    s16x8 wa[PL][KST][2];
#pragma unroll
    for (int st = 0; st < KST; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int tap = NCP == 2 ? 4 * g + (e >> 1) : 4 * st + g, co = NCP == 2 ? (e & 1) : e;
                v[e] = (tap < 9 && co < NC) ? w_oihw[((size_t)co * 32 + s * 16 + pl) * 9 + tap] : 0.f;
            }
            uint4 wp[PL];
            split_bf16_planes<PL>(v, wp);
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                union { uint4 u; s16x8 h; } pk;
                pk.u = wp[p];
                wa[p][st][s] = pk.h;
            }
        }
    const int bpg = gridDim.x / groups, bg = blockIdx.x / bpg;
    const long gpix = (long)N * H * W / groups, total = (bg + 1) * gpix, ngrp16 = (gpix + 15) / 16;
    // a wave takes CONSECUTIVE 16-pixel groups: its pixel coordinates advance by additions (the grid-stride form spent more
    // on three 64-bit divisions per group than on everything else)
    const long wave = ((long)(blockIdx.x - bg * bpg) * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)bpg * blockDim.x) >> 6;
    const long gpw = (ngrp16 + nwaves - 1) / nwaves, grp0 = wave * gpw;
    const int ngr = (int)(grp0 >= ngrp16 ? 0 : (grp0 + gpw > ngrp16 ? ngrp16 - grp0 : gpw));
    // after the swap: lane group g owns channels cb .. cb + 7 (g = 0: 0, 1: 16, 2: 8, 3: 24)
    const int cb = (g & 1) ? 16 + (g - 1) * 4 : g * 4;
    float ms[8], mh[8], cA[8], cB[8], cC[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = bg * 32 + cb + e;
        if constexpr (!RELU) { ms[e] = mscale[c]; mh[e] = mshift[c]; }
        s1[e] = s2[e] = 0.f;
        if constexpr (APPLY) { cA[e] = coef[(bg * 3 + 0) * 32 + cb + e]; cB[e] = coef[(bg * 3 + 1) * 32 + cb + e]; cC[e] = coef[(bg * 3 + 2) * 32 + cb + e]; }
    }
    // Every access goes through a buffer descriptor with a 32-bit byte offset: a load that must deliver zeros (a tap that does
    // not exist, a group past the wave's share) gets an offset past the end -- the hardware returns 0, no select touches a
    // loaded value before the group is USED, and the loads of group i + 1 stay in flight under group i; a store of a pixel that
    // does not exist is dropped the same way.  (The entry points require the tensors below 2 GiB.)
    const unsigned npx_all = (unsigned)((long)N * H * W);
    const int Wp = W + 2;
    constexpr int PB = 2 * NCP * PL;                        // bytes of a pixel of the class map
    const auto rs_dl = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(dlp), 0, N * (H + 2) * Wp * PB, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(y), 0, (int)(npx_all * (unsigned)PXB), 0x00020000);
    const auto rs_dx = __builtin_amdgcn_make_buffer_rsrc(dx, 0, STORE ? (int)(npx_all * (unsigned)PXB) : 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const long pxl = bg * gpix + grp0 * 16 + pl;            // this lane's first pixel; (cx, cy) its column / row inside its image
    unsigned px = (unsigned)pxl;
    int cx = (int)(pxl % W), cy = (int)((pxl / W) % H);
    // dlogits come as dh_head_dlogits_pack left them: [N][H + 2][W + 2] class pieces (x PL planes) with a border of zeros -- a tap
    // needs no bounds test, only an offset from the lane's own (padded) position pc4 (bytes)
    unsigned pc4 = (unsigned)(((pxl / ((long)W * H)) * (H + 2) + cy + 1) * Wp + cx + 1) * (unsigned)PB;
    constexpr int NLD = NCP == 2 ? 4 : KST;                 // taps this lane gathers per group
    int toff[NLD];                                          // their byte offsets from pc4
    bool thave[NLD];
#pragma unroll
    for (int q = 0; q < NLD; ++q) {
        const int tap = NCP == 2 ? 4 * g + q : 4 * q + g, kh = tap / 3, kw = tap - kh * 3;
        thave[q] = tap < 9;
        toff[q] = ((1 - kh) * Wp + (1 - kw)) * PB;
    }
    struct Grp { u32x4 bfr[PL][KST]; u32x4 yraw[PL]; unsigned at; };      // B fragments of the K steps; (fp32: yraw = the lane's 8 values)
    auto request = [&](Grp& q_, bool valid) {
        const bool inb = valid & (px < (unsigned)total);
        q_.at = inb ? px * (unsigned)PXB + cb * (unsigned)sizeof(T) : OOB;
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
            const unsigned off = (inb & thave[q]) ? pc4 + toff[q] : OOB;
            if constexpr (NCP == 2) {
                if constexpr (F32) {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_dl, off, 0, 0);
                    q_.bfr[0][0][q] = v[0]; q_.bfr[PL - 1][0][q] = v[1];
                } else {
                    q_.bfr[0][0][q] = __builtin_amdgcn_raw_buffer_load_b32(rs_dl, off, 0, 0);
                }
            } else {
                q_.bfr[0][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_dl, off, 0, 0);
                if constexpr (F32) q_.bfr[PL - 1][q] = __builtin_amdgcn_raw_buffer_load_b128(rs_dl, off + 16u, 0, 0);
            }
        }
        q_.yraw[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, q_.at, 0, 0);
        if constexpr (F32) q_.yraw[PL - 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, q_.at + 16u, 0, 0);      // (OOB + 16 stays out of range)
        px += 16; cx += 16; pc4 += 16 * PB;
        while (cx >= W) {                                   // next row: over the two border pixels; next image: over two border rows
            cx -= W; pc4 += 2 * PB;
            if (++cy == H) { cy = 0; pc4 += 2 * Wp * PB; }
        }
    };
    // lane (pl, g): pixels 4 g .. 4 g + 3 of column 16 c + pl of a tile
    auto frag = [&](const unsigned char* t, int pitch, int c) {
        const unsigned char* base = t + (g * 4 + (pl >> 2)) * pitch + (c * 16 + (pl & 3) * 4) * 2;
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    };
    auto use = [&](const Grp& q_) {
        f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < KST; ++st) {
            const s16x8 b0 = __builtin_bit_cast(s16x8, q_.bfr[0][st]);
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][st][0], b0, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][st][1], b0, d1, 0, 0, 0);
            if constexpr (F32) {
                const s16x8 bl = __builtin_bit_cast(s16x8, q_.bfr[PL - 1][st]);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][st][0], bl, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[0][st][1], bl, d1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[PL - 1][st][0], b0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[PL - 1][st][1], b0, d1, 0, 0, 0);
            }
        }
        float r[8], yv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0[j]), __float_as_uint(d1[j]), false, false);
            r[j] = __uint_as_float(sw[0]);
            r[4 + j] = __uint_as_float(sw[1]);
        }
        if constexpr (F32) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { yv[k] = __uint_as_float(q_.yraw[0][k]); yv[4 + k] = __uint_as_float(q_.yraw[PL - 1][k]); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { yv[2 * k] = __uint_as_float(q_.yraw[0][k] << 16); yv[2 * k + 1] = __uint_as_float(q_.yraw[0][k] & 0xffff0000u); }
        }
        float z[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            z[e] = RELU ? yv[e] : __builtin_fmaf(yv[e], ms[e], mh[e]);      // (explicit fma: every instantiation rounds alike)
            r[e] = z[e] > 0.f ? r[e] : 0.f;          // (a group of zeros has r = 0 already)
        }
        if constexpr (WGRAD) {
            if constexpr (RELU && !F32) {            // the head's input is y itself
                *reinterpret_cast<uint4*>(tH + pl * HB_TP + cb * 2) = make_uint4(q_.yraw[0][0], q_.yraw[0][1], q_.yraw[0][2], q_.yraw[0][3]);
            } else {
                float hv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) hv[e] = fmaxf(z[e], 0.f);        // (a group of zeros: its dlogits are zero)
                uint4 hp[PL];
                split_bf16_planes<PL>(hv, hp);
#pragma unroll
                for (int p = 0; p < PL; ++p) *reinterpret_cast<uint4*>(tH + p * TILE + pl * HB_TP + cb * 2) = hp[p];
            }
            // the neighbourhood tile, k-major as the fragments are: NCP = 2: columns 8 g ..; NCP = 8: the piece of tap 4 st + g
#pragma unroll
            for (int p = 0; p < PL; ++p)
#pragma unroll
                for (int st = 0; st < KST; ++st)
                    *reinterpret_cast<uint4*>(tN + p * TILEN + pl * TNP + (NCP == 2 ? g : 4 * st + g) * 16) =
                        make_uint4(q_.bfr[p][st][0], q_.bfr[p][st][1], q_.bfr[p][st][2], q_.bfr[p][st][3]);
            asm volatile("" ::: "memory");
            s16x4 hf[PL][2], nf[PL][NKB];
#pragma unroll
            for (int p = 0; p < PL; ++p) {
#pragma unroll
                for (int c = 0; c < 2; ++c) hf[p][c] = frag(tH + p * TILE, HB_TP, c);
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) nf[p][kb] = frag(tN + p * TILEN, TNP, kb);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    wacc[cs][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hf[0][cs], nf[0][kb], wacc[cs][kb], 0, 0, 0);
                    if constexpr (F32) {
                        wacc[cs][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hf[0][cs], nf[PL - 1][kb], wacc[cs][kb], 0, 0, 0);
                        wacc[cs][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(hf[PL - 1][cs], nf[0][kb], wacc[cs][kb], 0, 0, 0);
                    }
                }
            // the centre tap is the pixel's own dlogits: the bias gradient.  NCP = 2: tap 4 = lane group 1's first load;
            // NCP = 8: tap 4 = step 1 of lane group 0
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                if constexpr (NCP == 2) {
                    bsum[0] += __uint_as_float(q_.bfr[p][0][0] << 16);
                    bsum[1] += __uint_as_float(q_.bfr[p][0][0] & 0xffff0000u);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bsum[2 * k] += __uint_as_float(q_.bfr[p][1][k] << 16);
                        bsum[2 * k + 1] += __uint_as_float(q_.bfr[p][1][k] & 0xffff0000u);
                    }
                }
            }
        }
        auto store8 = [&](const float (&o)[8]) {
            if constexpr (F32) {
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])}, rs_dx, q_.at, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(o[4]), __float_as_uint(o[5]), __float_as_uint(o[6]), __float_as_uint(o[7])}, rs_dx, q_.at + 16u, 0, 0);
            } else {
                const uint4 pk = pack16<bf16>(o);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{pk.x, pk.y, pk.z, pk.w}, rs_dx, q_.at, 0, 0);
            }
        };
        if constexpr (APPLY) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(cA[e], r[e], __builtin_fmaf(cB[e], yv[e], cC[e]));
            store8(o);
        } else if constexpr (RELU) {
            store8(r);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += r[e]; s2[e] = __builtin_fmaf(r[e], yv[e], s2[e]); }
        }
    };
    // HB_DEPTH groups in flight per wave
    Grp gq[HB_DEPTH];
#pragma unroll
    for (int d = 0; d < HB_DEPTH - 1; ++d) request(gq[d], d < ngr);
    for (int i = 0; i < ngr; i += HB_DEPTH) {
#pragma unroll
        for (int d = 0; d < HB_DEPTH; ++d) {
            request(gq[(d + HB_DEPTH - 1) % HB_DEPTH], i + d + HB_DEPTH - 1 < ngr);
            use(gq[d]);
        }
    }
    if constexpr (!APPLY) {
        const int wv = threadIdx.x >> 6;
        if constexpr (!RELU) {
            __shared__ float red[4][2][32];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = row16_sum(s1[e]), b = row16_sum(s2[e]);
                if (pl == 0) { red[wv][0][cb + e] = a; red[wv][1][cb + e] = b; }
            }
            __syncthreads();
            if (threadIdx.x < 64) {
                const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
                partial[((size_t)which * 32 + c) * gridDim.x + blockIdx.x] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
            }
        }
        if constexpr (WGRAD) {
            float (*wred)[WCOLS] = reinterpret_cast<float (*)[WCOLS]>(hb_tiles);
            __syncthreads();         // every wave has read its last fragments
            // D: lane (pl, g) holds rows ci = 16 cs + 4 g + j of column k = 16 kb + pl
#pragma unroll
            for (int cs = 0; cs < 2; ++cs)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = kb * 16 + pl, ci = cs * 16 + 4 * g + j;
                        if (k < 9 * NCP) wred[wv][ci * (9 * NCP) + k] = wacc[cs][kb][j];
                    }
#pragma unroll
            for (int c = 0; c < NCP; ++c) {
                const float b = row16_sum(bsum[c]);
                if (lane == (NCP == 2 ? 16 : 0)) wred[wv][32 * 9 * NCP + c] = b;
            }
            __syncthreads();
            for (int i = threadIdx.x; i < WCOLS; i += 256)
                wpartial[(size_t)i * gridDim.x + blockIdx.x] = wred[0][i] + wred[1][i] + wred[2][i] + wred[3][i];
        }
    }
}

// blocks [0, bn_blocks = 32 or 0): the BatchNorm's channels (bn_bwd_finalize_coef_body); then hb_wcols(NCP) blocks: one column of the head's
// weight-gradient partials each -> dw [n_class][32][3][3] / db [n_class] (+)=
__global__ __launch_bounds__(256) void head_bwd_finalize_kernel(const float* __restrict__ partial, int bpg, int G,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, float inv_m, float* __restrict__ coef,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                                const float* __restrict__ wpartial, int NC, float* __restrict__ dw,
                                                                float* __restrict__ db, int bn_blocks, int NCP) {
    if ((int)blockIdx.x < bn_blocks) {
        bn_bwd_finalize_coef_body(blockIdx.x, partial, bpg, G, 32, mean, invstd, gamma, inv_m, coef, dgamma, dbeta, accumulate);
        return;
    }
    __shared__ float red[4];
    const int col = blockIdx.x - bn_blocks, rows = bpg * G;
    const float* p = wpartial + (size_t)col * rows;
    float a = 0.f;
    for (int t = threadIdx.x; t < rows; t += 256) a += p[t];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = red[0] + red[1] + red[2] + red[3];
        const int kk = 9 * NCP;
        if (col < 32 * kk) {
            const int ci = col / kk, k = col % kk, tap = k / NCP, co = k % NCP;
            if (co < NC) { float* o = dw + ((size_t)co * 32 + ci) * 9 + tap; *o = accumulate ? *o + v : v; }
        } else if (col - 32 * kk < NC) {
            float* o = db + (col - 32 * kk);
            *o = accumulate ? *o + v : v;
        }
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int dh_bn_finalize(const float* partial, int ntiles, int CP, int C, int groups, double count,
                              const float* gamma, const float* beta, float* running_mean, float* running_var,
                              float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                              long long* num_batches_tracked, void* stream) {
    DH_REQUIRE(groups > 0 && ntiles % groups == 0, "bn_finalize: ntiles=%d not divisible by groups=%d", ntiles, groups);
    DH_REQUIRE(groups == 1 || groups == 2 || groups == 4, "bn_finalize: 1, 2 or 4 statistics groups, got %d", groups);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, ST(stream), partial, ntiles, CP, C, groups, count,
                       gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift,
                       num_batches_tracked);
    DH_CHECK_LAUNCH("bn_finalize");
    return 0;
}

extern "C" int dh_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, int C, float* scale, float* shift,
                                 void* stream) {
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3(dh_cdiv(C, 64)), dim3(64), 0, ST(stream), gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    DH_CHECK_LAUNCH("bn_eval_params");
    return 0;
}

static int bn_apply_impl(int dtype, const void* x, const void* residual, void* y, const float* scale, const float* shift, long npix,
                         int C, int groups, int act, unsigned char* relu_bits, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;       // elements of one 16-byte piece
    DH_REQUIRE(C % V == 0 && npix % groups == 0, "bn_apply: C=%d npix=%ld groups=%d", C, npix, groups);
    DH_REQUIRE(!relu_bits || act == DH_ACT_RELU, "bn_apply: the ReLU mask bytes go with act = DH_ACT_RELU");
    const long nvec = npix * C / V, gvec = nvec / groups;
    (void)gvec;
    if (dtype == DH_DTYPE_BF16) launch_bn_apply<bf16>(x, residual, y, scale, shift, nvec, C, groups, act, ST(stream), relu_bits);
    else launch_bn_apply<float>(x, residual, y, scale, shift, nvec, C, groups, act, ST(stream), relu_bits);
    DH_CHECK_LAUNCH("bn_apply");
    return 0;
}
extern "C" int dh_bn_apply(int dtype, const void* x, const void* residual, void* y, const float* scale,
                           const float* shift, long npix, int C, int groups, int act, void* stream) {
    return bn_apply_impl(dtype, x, residual, y, scale, shift, npix, C, groups, act, nullptr, stream);
}
// ... + relu_bits [npix * C / V] bytes (V = 8 bf16 / 4 fp32 elements per 16-byte piece): byte i = the ReLU mask of piece i of y
// (bit j: y[V i + j] > 0), for dh_bn_bwd_bits / dh_bn_bwd_persist_bits (act = DH_ACT_RELU)
extern "C" int dh_bn_apply_bits(int dtype, const void* x, const void* residual, void* y, const float* scale, const float* shift,
                                long npix, int C, int groups, int act, unsigned char* relu_bits, void* stream) {
    DH_REQUIRE(relu_bits, "bn_apply_bits: relu_bits missing");
    return bn_apply_impl(dtype, x, residual, y, scale, shift, npix, C, groups, act, relu_bits, stream);
}

// workspace: partial [groups*bpg][2][C] floats + sums [groups][2][C] floats
// Stem tail backward (see pool_bn_bwd_reduce_kernel): dpool [N][OH][OW][C] + arg-max -> masked gradient d [N][H][W][C] of the
// pre-pool activation relu(y * mask_scale + mask_shift); dgamma / dbeta (+)=; coef [groups][3][C] for dh_stem_wgrad_bn.
// workspace: dh_stem_pool_bn_bwd_workspace_size bytes.
// 768 workgroups = three per CU, what the kernel's ~150 registers per lane keep resident: one full round, no ragged second one
extern "C" long dh_stem_pool_bn_bwd_workspace_size(int C, int groups) { return (long)(768 / groups) * groups * 2 * C * 4; }
extern "C" int dh_stem_pool_bn_bwd_plus(const unsigned char* argmax, const void* dpool, const void* extra, const void* y,
                                        const float* mask_scale, const float* mask_shift, const float* mean, const float* invstd,
                                        const float* gamma, int N, int H, int W, int C, int groups, void* d, float* coef,
                                        float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream);
extern "C" int dh_stem_pool_bn_bwd(const unsigned char* argmax, const void* dpool, const void* y, const float* mask_scale,
                                   const float* mask_shift, const float* mean, const float* invstd, const float* gamma, int N,
                                   int H, int W, int C, int groups, void* d, float* coef, float* dgamma, float* dbeta,
                                   int accumulate, void* workspace, void* stream) {
    return dh_stem_pool_bn_bwd_plus(argmax, dpool, nullptr, y, mask_scale, mask_shift, mean, invstd, gamma, N, H, W, C, groups, d,
                                    coef, dgamma, dbeta, accumulate, workspace, stream);
}
// ... with a second gradient `extra` [N][H][W][C] of the pre-pool activation (or NULL) added before the mask
extern "C" int dh_stem_pool_bn_bwd_plus(const unsigned char* argmax, const void* dpool, const void* extra, const void* y,
                                        const float* mask_scale, const float* mask_shift, const float* mean, const float* invstd,
                                        const float* gamma, int N, int H, int W, int C, int groups, void* d, float* coef,
                                        float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream) {
    DH_REQUIRE(C % 8 == 0 && 256 % (C / 8) == 0 && 2 * C <= 256, "stem_pool_bn_bwd: unsupported C=%d", C);
    DH_REQUIRE(groups >= 1 && groups <= BN_MAXG && N % groups == 0, "stem_pool_bn_bwd: N=%d groups=%d", N, groups);
    const int bpg = 768 / groups, OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    float* partial = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(pool_bn_bwd_reduce_kernel, dim3(groups * bpg), dim3(256), 0, ST(stream), argmax, (const bf16*)dpool,
                       (const bf16*)extra, (const bf16*)y, mask_scale, mask_shift, (bf16*)d, partial, N / groups, H, W, C, OH, OW, bpg);
    hipLaunchKernelGGL(bn_bwd_finalize_coef_kernel, dim3(C), dim3(64 * groups), 0, ST(stream), partial, bpg, groups, C, mean, invstd,
                       gamma, 1.0f / (float)((long)(N / groups) * H * W), coef, dgamma, dbeta, accumulate);
    DH_CHECK_LAUNCH("stem_pool_bn_bwd");
    return 0;
}

extern "C" long dh_bn_bwd_workspace_size(long npix, int C, int groups) {
    const int bpg = 1024 / groups;      // ~1024 workgroups in total (4 per CU)
    return ((long)groups * bpg * 2 * C + (long)groups * 2 * C) * 4;
}

static int bn_bwd_impl(int dtype, const void* dout, const void* out_relu, bool bits, const void* x, const float* mean,
                       const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                       void* dres, float* dgamma, float* dbeta, int accumulate, const float* mask_scale,
                       const float* mask_shift, void* workspace, void* stream) {
    DH_REQUIRE(!(out_relu && mask_scale), "bn_bwd: give the ReLU mask either as out_relu or as mask_scale/shift");
    DH_REQUIRE(!bits || out_relu, "bn_bwd: mask bytes missing");
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;       // elements of one 16-byte piece
    DH_REQUIRE(C % V == 0 && (256 * V) % C == 0, "bn_bwd: unsupported C=%d", C);
    DH_REQUIRE(npix % groups == 0 && groups <= BN_MAXG, "bn_bwd: npix %% groups, at most %d groups", BN_MAXG);
    const int bpg = 1024 / groups;      // ~1024 workgroups in total (4 per CU)
    const long ppg = npix / groups;
    float* partial = reinterpret_cast<float*>(workspace);
    float* sums = partial + (long)groups * bpg * 2 * C;
    const long nvec = npix * C / V;
    if (dtype == DH_DTYPE_BF16) {
        launch_bn_bwd_reduce<bf16>(dout, out_relu, x, mean, invstd, C, ppg, bpg, groups, partial, mask_scale, mask_shift, ST(stream), bits);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * groups), 0, ST(stream), partial, bpg, groups, C, sums,
                           dgamma, dbeta, accumulate);
        launch_bn_bwd_apply<bf16>(dout, out_relu, x, mean, invstd, gamma, sums, ppg, nvec, C, groups, dx, dres, mask_scale,
                                  mask_shift, ST(stream), bits);
    } else {
        launch_bn_bwd_reduce<float>(dout, out_relu, x, mean, invstd, C, ppg, bpg, groups, partial, mask_scale, mask_shift, ST(stream), bits);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * groups), 0, ST(stream), partial, bpg, groups, C, sums,
                           dgamma, dbeta, accumulate);
        launch_bn_bwd_apply<float>(dout, out_relu, x, mean, invstd, gamma, sums, ppg, nvec, C, groups, dx, dres, mask_scale,
                                   mask_shift, ST(stream), bits);
    }
    DH_CHECK_LAUNCH("bn_bwd");
    return 0;
}
extern "C" int dh_bn_bwd(int dtype, const void* dout, const void* out_relu, const void* x, const float* mean,
                         const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                         void* dres, float* dgamma, float* dbeta, int accumulate, const float* mask_scale,
                         const float* mask_shift, void* workspace, void* stream) {
    return bn_bwd_impl(dtype, dout, out_relu, false, x, mean, invstd, gamma, npix, C, groups, dx, dres, dgamma, dbeta, accumulate,
                       mask_scale, mask_shift, workspace, stream);
}
// the same with the ReLU mask given as the mask BYTES dh_bn_apply_bits wrote (bf16): the two passes read npix * C / 8 bytes
// where dh_bn_bwd reads the post-activation tensor (npix * C * 2 bytes), twice
extern "C" int dh_bn_bwd_bits(int dtype, const void* dout, const unsigned char* relu_bits, const void* x, const float* mean,
                              const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                              void* dres, float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream) {
    DH_REQUIRE(relu_bits, "bn_bwd_bits: relu_bits missing");
    return bn_bwd_impl(dtype, dout, relu_bits, true, x, mean, invstd, gamma, npix, C, groups, dx, dres, dgamma, dbeta, accumulate,
                       nullptr, nullptr, workspace, stream);
}

// BN backward when the producer of dout (a gated data-gradient launch of conv_mfma) already applied the ReLU mask and
// left per-tile partials [ntiles][2][C] = (sum g, sum g*xhat): combine them, then dx = gamma*invstd*(g - (s1 + xhat*s2)/M).
// workspace: sums [groups][2][C] floats
extern "C" int dh_bn_bwd_from_partials(int dtype, const void* g, const void* x, const float* partial, int ntiles,
                                       const float* mean, const float* invstd, const float* gamma, long npix, int C,
                                       int groups, void* dx, float* dgamma, float* dbeta, int accumulate,
                                       void* workspace, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0 && groups > 0 && ntiles % groups == 0 && npix % groups == 0,
               "bn_bwd_from_partials: C=%d ntiles=%d npix=%ld groups=%d", C, ntiles, npix, groups);
    float* sums = reinterpret_cast<float*>(workspace);
    const long ppg = npix / groups, nvec = npix * C / V;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64 * groups), 0, ST(stream), partial, ntiles / groups, groups, C, sums,
                       dgamma, dbeta, accumulate);
    if (dtype == DH_DTYPE_BF16)
        launch_bn_bwd_apply<bf16>(g, nullptr, x, mean, invstd, gamma, sums, ppg, nvec, C, groups, dx, nullptr, nullptr, nullptr,
                                  ST(stream));
    else
        launch_bn_bwd_apply<float>(g, nullptr, x, mean, invstd, gamma, sums, ppg, nvec, C, groups, dx, nullptr, nullptr, nullptr,
                                   ST(stream));
    DH_CHECK_LAUNCH("bn_bwd_from_partials");
    return 0;
}

extern "C" int dh_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y,
                                float* stats, long rows, int C, float eps, void* stream) {
    DH_REQUIRE(C == 32, "layernorm: only dim 32 (the reference's transformer width), got %d", C);
    if (rows == 0) return 0;
    const int grid = dh_cdiv(rows * 8, 256);
    if (dtype == DH_DTYPE_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (const bf16*)x, gamma, beta,
                           (bf16*)y, stats, rows, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)x, gamma, beta,
                           (float*)y, stats, rows, eps);
    DH_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

static inline int ln_bwd_grid(long rows) { long g = dh_cdiv(rows * 8, 256); return (int)(g > 512 ? 512 : g); }
extern "C" long dh_layernorm_bwd_workspace_size(long rows) { return ((long)ln_bwd_grid(rows) * 64 + 64) * 4; }

extern "C" int dh_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma,
                                void* dx, const void* dx_add, float* dgamma, float* dbeta, int accumulate,
                                long rows, int C, void* workspace, void* stream) {
    DH_REQUIRE(C == 32, "layernorm_bwd: only dim 32, got %d", C);
    if (rows == 0) return 0;
    const int grid = ln_bwd_grid(rows);
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == DH_DTYPE_BF16)
        hipLaunchKernelGGL(ln_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (const bf16*)dy, (const bf16*)x,
                           stats, gamma, (bf16*)dx, (const bf16*)dx_add, partial, rows);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)dy,
                           (const float*)x, stats, gamma, (float*)dx, (const float*)dx_add, partial, rows);
    // per-block partial rows are [dgamma(32) | dbeta(32)]: reduce into a 64-float temp, then split
    launch_reduce(partial, (long)grid, 64L, 1.0f, partial + (long)grid * 64, 0, ST(stream));
    launch_reduce(partial + (long)grid * 64, 1L, 32L, 1.0f, dgamma, accumulate, ST(stream));
    launch_reduce(partial + (long)grid * 64 + 32, 1L, 32L, 1.0f, dbeta, accumulate, ST(stream));
    DH_CHECK_LAUNCH("layernorm_bwd");
    return 0;
}

extern "C" int dh_reduce_partials(const float* partial, long nt, long n, float scale, float* out, int accumulate,
                                  void* stream) {
    if (n == 0) return 0;
    launch_reduce(partial, nt, n, scale, out, accumulate, ST(stream));
    DH_CHECK_LAUNCH("reduce_partials");
    return 0;
}

// dh_head_dgrad3x3 + dh_bn_bwd in two passes that never materialise the head's data gradient (head_bn_bwd_kernel): bf16,
// n_class <= 2, 32 channels.  dlp [N][H + 2][W + 2] bf16 pairs (dh_head_dlogits_pack), w_oihw [n_class][32][3][3] fp32, y the
// pre-BatchNorm activation [N][H][W][32], mask_scale / mask_shift / mean / invstd [groups][32] as dh_bn_finalize left them;
// dx [N][H][W][32] = the gradient of y; dgamma / dbeta (+)=.  workspace: dh_head_bn_bwd_workspace_size bytes.
// Replaces: the autograd of Conv2d(32, n_class, 3) -> ReLU -> BatchNorm2d(32) in models/help_funcs.py:7-15.
extern "C" int dh_head_bn_bwd_blocks(int N, int H, int W, int groups) {
    if (groups < 1 || N % groups) return 0;
    const long n16 = ((long)N * H * W / groups + 15) / 16;
    long bpg = (n16 * 64 + 255) / 256;
    // pass 1: every workgroup leaves 642 partial sums, one per row of the finalize kernel's columns (512 / 768 / 1024 / 2048
    // workgroups measured: 116 / 123 / 121 / 131 us for the three launches at 32 x 256 x 256 pixels)
    static const long cap1 = [] { const char* e = getenv("DAHITRA_HB_GRID1"); return e ? atol(e) : 512L; }();
    const long cap = cap1 / groups;
    if (bpg > cap) bpg = cap;
    return (int)(bpg * groups);
}
static int head_bn_bwd_apply_blocks(int N, int H, int W, int groups) {
    const long n16 = ((long)N * H * W / groups + 15) / 16;
    long bpg = (n16 * 64 + 255) / 256;
    static const long cap2 = [] { const char* e = getenv("DAHITRA_HB_GRID2"); return e ? atol(e) : 2048L; }();
    const long cap = cap2 / groups;
    if (bpg > cap) bpg = cap;
    return (int)(bpg * groups);
}
extern "C" long dh_head_bn_bwd_workspace_size(int N, int H, int W, int groups) {
    return ((long)dh_head_bn_bwd_blocks(N, H, W, groups) * (2 * 32 + HB_WCOLS_MAX) + (long)groups * 3 * 32) * 4;
}
// dtype = DH_DTYPE_BF16: one plane of class pieces per pixel; DH_DTYPE_F32 (the split-product mode): two (bf16 heads, bf16 remainders).
// A piece holds 2 classes (NC <= 2: one word) or 8 (3 <= NC <= 8: 16 bytes): dlp is [N][H + 2][W + 2][planes][piece].
extern "C" int dh_head_dlogits_pack(int dtype, const float* dlogits_nchw, int N, int NC, int H, int W, void* dlp, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 8 && dlogits_nchw && dlp, "head_dlogits_pack: n_class=%d", NC);
    const long n = (long)N * (H + 2) * (W + 2);
    DH_REQUIRE(n * (NC <= 2 ? 8 : 32) < (1L << 31), "head_dlogits_pack: %d x %d x %d does not fit a 2 GiB buffer descriptor", N, H, W);
    const dim3 grid(ew_grid(n, 256));
    hipStream_t st = ST(stream);
    if (dtype == DH_DTYPE_BF16) {
        if (NC <= 2) hipLaunchKernelGGL((head_dlogits_pack_kernel<1, 2>), grid, dim3(256), 0, st, dlogits_nchw, (unsigned*)dlp, N, NC, H, W);
        else hipLaunchKernelGGL((head_dlogits_pack_kernel<1, 8>), grid, dim3(256), 0, st, dlogits_nchw, (unsigned*)dlp, N, NC, H, W);
    } else {
        if (NC <= 2) hipLaunchKernelGGL((head_dlogits_pack_kernel<2, 2>), grid, dim3(256), 0, st, dlogits_nchw, (unsigned*)dlp, N, NC, H, W);
        else hipLaunchKernelGGL((head_dlogits_pack_kernel<2, 8>), grid, dim3(256), 0, st, dlogits_nchw, (unsigned*)dlp, N, NC, H, W);
    }
    DH_CHECK_LAUNCH("head_dlogits_pack");
    return 0;
}
template <typename T, int NCP>
static int head_bn_bwd_launch(const void* dlp, const float* w_oihw, int NC, const void* y, const float* mask_scale, const float* mask_shift,
                              const float* mean, const float* invstd, const float* gamma, int groups, void* dx, float* dgamma, float* dbeta,
                              float* dw, float* db, int accumulate, int N, int H, int W, void* workspace, hipStream_t st) {
    const int grid = dh_head_bn_bwd_blocks(N, H, W, groups), grid2 = head_bn_bwd_apply_blocks(N, H, W, groups);
    float* partial = reinterpret_cast<float*>(workspace);
    float* coef = partial + (size_t)grid * 2 * 32;
    float* wpartial = coef + (size_t)groups * 3 * 32;
    const float inv_m = (float)(1.0 / ((double)N * H * W / groups));
    if (dw) {
        hipLaunchKernelGGL((head_bn_bwd_kernel<T, 0, true, NCP>), dim3(grid), dim3(256), 0, st, (const unsigned*)dlp, w_oihw, N, H, W, NC,
                           (const T*)y, mask_scale, mask_shift, groups, partial, (const float*)nullptr, (T*)nullptr, wpartial);
        hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(32 + hb_wcols(NCP)), dim3(256), 0, st, partial, grid / groups, groups, mean,
                           invstd, gamma, inv_m, coef, dgamma, dbeta, accumulate, wpartial, NC, dw, db, 32, NCP);
    } else {
        hipLaunchKernelGGL((head_bn_bwd_kernel<T, 0, false, NCP>), dim3(grid), dim3(256), 0, st, (const unsigned*)dlp, w_oihw, N, H, W, NC,
                           (const T*)y, mask_scale, mask_shift, groups, partial, (const float*)nullptr, (T*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(bn_bwd_finalize_coef_kernel, dim3(32), dim3(64 * groups), 0, st, partial, grid / groups, groups, 32, mean,
                           invstd, gamma, inv_m, coef, dgamma, dbeta, accumulate);
    }
    hipLaunchKernelGGL((head_bn_bwd_kernel<T, 1, false, NCP>), dim3(grid2), dim3(256), 0, st, (const unsigned*)dlp, w_oihw, N, H, W, NC,
                       (const T*)y, mask_scale, mask_shift, groups, (float*)nullptr, coef, (T*)dx, (float*)nullptr);
    return 0;
}
// dtype = DH_DTYPE_F32: y / dx fp32, dlp with two planes, every product as three split bf16 products (the bf16x3 mode's
// arithmetic: callers use it only under dh_set_f32_mma_mode != 0).  n_class <= 8 (3 .. 8: the 8-class piece form of dlp).
extern "C" int dh_head_bn_bwd(int dtype, const void* dlp, const float* w_oihw, int NC, const void* y, const float* mask_scale,
                              const float* mask_shift, const float* mean, const float* invstd, const float* gamma, int groups,
                              void* dx, float* dgamma, float* dbeta, float* dw, float* db, int accumulate, int N, int H, int W,
                              void* workspace, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 8 && dlp && y && mask_scale && mask_shift && mean && invstd && gamma && dx && dgamma && dbeta && workspace,
               "head_bn_bwd: bad arguments (n_class=%d)", NC);
    DH_REQUIRE((dw == nullptr) == (db == nullptr), "head_bn_bwd: dw and db come together");
    const int grid = dh_head_bn_bwd_blocks(N, H, W, groups);
    DH_REQUIRE(grid > 0 && groups <= BN_MAXG, "head_bn_bwd: %d images do not split into %d groups", N, groups);
    DH_REQUIRE((long)N * H * W * 128 < (1L << 31), "head_bn_bwd: %d x %d x %d pixels x 128 bytes do not fit a 2 GiB buffer descriptor", N, H, W);
#define HB_GO(T, NCP) head_bn_bwd_launch<T, NCP>(dlp, w_oihw, NC, y, mask_scale, mask_shift, mean, invstd, gamma, groups, dx, dgamma, dbeta, \
                                                 dw, db, accumulate, N, H, W, workspace, ST(stream))
    if (dtype == DH_DTYPE_BF16) { if (NC <= 2) HB_GO(bf16, 2); else HB_GO(bf16, 8); }
    else { if (NC <= 2) HB_GO(float, 2); else HB_GO(float, 8); }
#undef HB_GO
    DH_CHECK_LAUNCH("head_bn_bwd");
    return 0;
}

// The class head behind a ReLU (classifier(conv_layer2(...)), models/networks.py:1351-1355; n_class <= 8): dx = (relu_out
// > 0) * (W^T (*) dlogits) [N][H][W][32] -- dh_head_dgrad3x3_relu from the zero-bordered class map dlp (dh_head_dlogits_pack) --
// AND the head's own weight / bias gradient dw [n_class][32][3][3] / db [n_class] ((+)= when accumulate) from the same loads
// of relu_out, which is the head's input (head_bn_bwd_kernel<T, 2, true, .>).  workspace: dh_head_bn_bwd_workspace_size(N, H, W, 1).
extern "C" int dh_head_relu_bwd(int dtype, const void* dlp, const float* w_oihw, int NC, const void* relu_out, void* dx, float* dw,
                                float* db, int accumulate, int N, int H, int W, void* workspace, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 8 && dlp && relu_out && dx && dw && db && workspace, "head_relu_bwd: bad arguments (n_class=%d)", NC);
    DH_REQUIRE((long)N * H * W * 128 < (1L << 31), "head_relu_bwd: %d x %d x %d pixels x 128 bytes do not fit a 2 GiB buffer descriptor", N, H, W);
    const int grid = dh_head_bn_bwd_blocks(N, H, W, 1);
    float* wpartial = reinterpret_cast<float*>(workspace);
#define HR_GO(T, NCP) hipLaunchKernelGGL((head_bn_bwd_kernel<T, 2, true, NCP>), dim3(grid), dim3(256), 0, ST(stream), (const unsigned*)dlp, w_oihw, \
                                         N, H, W, NC, (const T*)relu_out, (const float*)nullptr, (const float*)nullptr, 1, (float*)nullptr,          \
                                         (const float*)nullptr, (T*)dx, wpartial)
    if (dtype == DH_DTYPE_BF16) { if (NC <= 2) HR_GO(bf16, 2); else HR_GO(bf16, 8); }
    else { if (NC <= 2) HR_GO(float, 2); else HR_GO(float, 8); }
#undef HR_GO
    const int ncp = NC <= 2 ? 2 : 8;
    hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(hb_wcols(ncp)), dim3(256), 0, ST(stream), (const float*)nullptr, grid, 1,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, accumulate, wpartial, NC, dw, db, 0, ncp);
    DH_CHECK_LAUNCH("head_relu_bwd");
    return 0;
}
