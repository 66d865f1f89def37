// Fused token encoder (models/networks.py:457-512 == help_funcs.py:117-167: Residual(PreNorm(Attention)) +
// Residual(PreNorm(FeedForward)) on the 2L <= 8 semantic tokens of an image pair), fp32, for gfx950.
//
// The encoder is a few MFLOP per image on an [8][32] activation: as separate LayerNorm / linear / attention /
// GELU kernels it is ~22 launches of 5-25 us per layer, pure launch latency.  Here: ONE workgroup per image for the
// forward, ONE per image for the data gradient (recomputing the forward from the saved layer input), and ONE
// launch for every weight / bias / LayerNorm gradient (reductions over the B*n token rows).  All math is fp32 on
// the vector ALUs -- tokens stay fp32 in every compute mode (see tokens.hip) and the products are far too small
// for the matrix cores to matter.  Parameters are read in their torch layouts straight from the flat arena;
// consecutive layers of a stack lie `pstride` floats apart.
//
// A first version with one thread per output and 500-1500-step serial dot products was SLOWER than the launch chain
// (32 workgroups, each a long dependent-load loop).  The long reductions (K = inner for the out-projection,
// K = 3*inner for the qkv data gradient, K = B*n rows for the weight gradients) are therefore split over 8 (4) thread
// groups that each keep all token rows in registers, and combined through LDS in a fixed order; the attention dot
// products use 4 lanes per (head, query).
#include "common.h"

namespace {

constexpr int D = 32;
// threads per (one-image) workgroup of the forward / data-gradient kernels: only B workgroups exist, so the strided phases
// (projections, attention products, copies) run on two waves per SIMD; the fixed 8 x 32 / 4-lanes-per-(head, query) phases
// use the first 256 threads
constexpr int ENC_THREADS = 512;
constexpr int MAXN = 8;          // tokens per image (2 * token_len)

struct EncArgs {
    const float* x;              // [B][n][32] input tokens
    float* y;                    // fwd: [B][n][32] output;  bwd: dX
    float* xs;                   // layer inputs [depth][B][n][32] (fwd writes them when non-null; bwd reads them)
    const float* dy;             // bwd: gradient of the output
    const float *ln1_g, *ln1_b, *wqkv, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;     // layer 0
    long pstride;                // floats between consecutive layers' parameters
    int depth, B, n, heads, dh, mlp;
    float scale, eps;
    // bwd: per-row operands of the weight-gradient launch, each [depth][B*n][width]
    float *r_xn, *r_dqkv, *r_o, *r_dx1, *r_x1n, *r_dz, *r_h, *r_dx2;
    float* r_ln;                 // [depth][B][4][32]: per-image (dg1, db1, dg2, db2)
};

struct Lds {
    float *x, *xn, *qkv, *p, *o, *x1, *x1n, *z, *h, *xh1, *xh2, *st, *red;
};
// red: [8][n][32] scratch of the split reductions
__device__ __forceinline__ Lds carve(float* sm, int n, int inner, int heads, int mlp) {
    Lds l;
    l.x = sm; sm += n * D;
    l.xn = sm; sm += n * D;
    l.qkv = sm; sm += n * 3 * inner;
    l.p = sm; sm += heads * n * n;
    l.o = sm; sm += n * inner;
    l.x1 = sm; sm += n * D;
    l.x1n = sm; sm += n * D;
    l.z = sm; sm += n * mlp;
    l.h = sm; sm += n * mlp;
    l.xh1 = sm; sm += n * D;
    l.xh2 = sm; sm += n * D;
    l.st = sm; sm += 4 * n;       // [2][n][2] (mean, rstd) of LN1 / LN2
    l.red = sm;                   // [8][n][32]
    return l;
}
// floats of one saved (layer, image) record: the forward's LDS image up to (not including) the reduction scratch --
// layer input, both LayerNorm outputs / normalised values / statistics, qkv, probabilities, attention output,
// MLP pre-activations and activations
__host__ __device__ static inline size_t saved_img_floats(int n, int inner, int heads, int mlp) {
    return (size_t)n * D * 6 + (size_t)n * 3 * inner + (size_t)heads * n * n + (size_t)n * inner + 2 * (size_t)n * mlp + 4 * n;
}
static inline size_t fwd_lds_floats(int n, int inner, int heads, int mlp) {
    return (size_t)n * D * 6 + (size_t)n * 3 * inner + (size_t)heads * n * n + (size_t)n * inner + 2 * (size_t)n * mlp + 4 * n +
           (size_t)8 * n * D;
}

// rows t < n of `src` -> xhat (dst_hat) and affine (dst), 32 lanes per row; stats st[t] = (mean, rstd)
__device__ __forceinline__ void layer_norm_rows(const float* src, const float* g, const float* b, float eps, int n,
                                                float* dst, float* dst_hat, float* st) {
    const int t = threadIdx.x >> 5, c = threadIdx.x & 31;
    if (t < n) {
        const float v = src[t * D + c];
        float s = v;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mu = s * (1.f / D);
        float q = (v - mu) * (v - mu);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rstd = rsqrtf(q * (1.f / D) + eps);
        const float xh = (v - mu) * rstd;
        dst_hat[t * D + c] = xh;
        dst[t * D + c] = xh * g[c] + b[c];
        if (c == 0) { st[t * 2] = mu; st[t * 2 + 1] = rstd; }
    }
}

// out[t][c] (c < 32) = sum_{j < K} in[t][j] * W[j*sj + c*sc]  for all rows t < n, K split over the 8 groups of 32
// threads (thread = (kg, c)); every thread keeps the n row sums in registers, `red` [8][n][32] combines them.
// The caller reads the result from red[0 .. n*32) after the trailing barrier.
__device__ __forceinline__ void ksplit_rows(const float* in, int in_pitch, const float* __restrict__ W, long sj, long sc, int K,
                                            int n, float* red) {
    // 16 K-groups: wave w (ENC_THREADS / 64 = 8) x lane half h; the two halves of a wave meet by one cross-half shuffle, so `red`
    // still combines 8 partial sums.  (The first form ran on 8 groups of 32 lanes -- half of the workgroup idle -- with 8 loads in
    // flight per lane: 24 rounds of L2 latency for K = 1536, 21.9 of the data-gradient kernel's 60 us; tools/enc_timeline.py.)
    const int w = threadIdx.x >> 6, h = (threadIdx.x >> 5) & 1, c = threadIdx.x & 31, kg = w * 2 + h;
    const int per = (((K + 15) >> 4) + 3) & ~3, j0 = min(K, kg * per), j1 = min(K, j0 + per);     // multiples of 4: 16-byte LDS reads
    float acc[MAXN];
#pragma unroll
    for (int t = 0; t < MAXN; ++t) acc[t] = 0.f;
    int j = j0;
    for (; j + 15 < j1; j += 16) {            // sixteen weight loads in flight per step (the loop is load-latency-bound)
        float wv[16];
        if (sj == 1) {
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                const float4 v = *reinterpret_cast<const float4*>(W + (long)c * sc + j + u);
                wv[u] = v.x; wv[u + 1] = v.y; wv[u + 2] = v.z; wv[u + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = W[(long)(j + u) * sj + (long)c * sc];
        }
#pragma unroll
        for (int t = 0; t < MAXN; ++t)
            if (t < n) {
                const float* r = in + t * in_pitch + j;
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(r + u);
                    acc[t] += x.x * wv[u] + x.y * wv[u + 1] + x.z * wv[u + 2] + x.w * wv[u + 3];
                }
            }
    }
    for (; j + 3 < j1; j += 4) {
        const float w0 = W[(long)j * sj + (long)c * sc], w1 = W[(long)(j + 1) * sj + (long)c * sc];
        const float w2 = W[(long)(j + 2) * sj + (long)c * sc], w3 = W[(long)(j + 3) * sj + (long)c * sc];
#pragma unroll
        for (int t = 0; t < MAXN; ++t)
            if (t < n) {
                const float4 x = *reinterpret_cast<const float4*>(in + t * in_pitch + j);
                acc[t] += x.x * w0 + x.y * w1 + x.z * w2 + x.w * w3;
            }
    }
    for (; j < j1; ++j) {
        const float wj = W[(long)j * sj + (long)c * sc];
#pragma unroll
        for (int t = 0; t < MAXN; ++t)
            if (t < n) acc[t] += in[t * in_pitch + j] * wj;
    }
#pragma unroll
    for (int t = 0; t < MAXN; ++t) {
        acc[t] += __shfl_xor(acc[t], 32, 64);
        if (t < n && h == 0) red[(w * n + t) * D + c] = acc[t];
    }
    __syncthreads();
    if (threadIdx.x < n * D) {
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) sum += red[g * n * D + threadIdx.x];
        red[threadIdx.x] = sum;             // slot [0][t][c]: only this thread reads/writes it
    }
    __syncthreads();
}

// -DENC_TIMING (tools/enc_timeline.py; never in the product build): wall-clock stamps of thread 0 after every phase of the LAST layer
#ifdef ENC_TIMING
__device__ long long g_enct[1024 * 16];
#define ENC_T(k) do { if (threadIdx.x == 0) g_enct[b * 16 + (k)] = (long long)wall_clock64(); } while (0)
#define ENC_TF(k) do { if (threadIdx.x == 0) g_enct[512 * 16 + blockIdx.x * 16 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define ENC_T(k) do { } while (0)
#define ENC_TF(k) do { } while (0)
#endif
// one layer forward on the tokens in l.x; leaves every intermediate in LDS; the layer output goes to `out`
// (LDS [n][32], may alias l.x)
__device__ __forceinline__ void layer_forward(const EncArgs& a, const Lds& l, int ly, float* out) {
    const int tid = threadIdx.x, n = a.n, inner = a.heads * a.dh, mlp = a.mlp;
    const long ps = (long)ly * a.pstride;
    ENC_TF(0);
    layer_norm_rows(l.x, a.ln1_g + ps, a.ln1_b + ps, a.eps, n, l.xn, l.xh1, l.st);
    __syncthreads();
    ENC_TF(1);
    // qkv[t][j] = sum_c xn[t][c] * Wqkv[j][c]: a thread owns column j (its 128-byte weight row stays in registers)
    const float* wqkv = a.wqkv + ps;
    // (all of a lane's weight rows requested before the first use -- one L2 round trip instead of three at inner = 512 -- was
    // measured and lost: 205 registers, 12.2 instead of 7.4 us for this phase)
    for (int j = tid; j < 3 * inner; j += blockDim.x) {
        float w[D];
#pragma unroll
        for (int c = 0; c < D; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(wqkv + (size_t)j * D + c);
            w[c] = v.x; w[c + 1] = v.y; w[c + 2] = v.z; w[c + 3] = v.w;
        }
        for (int t = 0; t < n; ++t) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D; c += 4) {          // (same order of the 32 products; one ds_read_b128 per four)
                const float4 x = *reinterpret_cast<const float4*>(l.xn + t * D + c);
                s += x.x * w[c]; s += x.y * w[c + 1]; s += x.z * w[c + 2]; s += x.w * w[c + 3];
            }
            l.qkv[t * 3 * inner + j] = s;
        }
    }
    __syncthreads();
    ENC_TF(2);
    // attention probabilities p[h][t][s] = softmax_s(scale * <q_t, k_s>): 4 lanes per (h, t) split the dh-long dots
    {
        const int pair = tid >> 2, sub = tid & 3, npair = a.heads * n;
        const int h = pair / n, t = pair % n;
        const int e0 = sub * (a.dh >> 2), e1 = e0 + (a.dh >> 2);
        float sc[MAXN];
        if (pair < npair) {
            // 16-byte LDS reads (dh / 4 is a multiple of 4): with scalar reads this phase was 256 dependent-latency
            // ds_read_b32 per thread, 11.3 of the forward's 39 us (tools/build_timing_barriers.py + barrier_timeline.py)
            const float* q = l.qkv + t * 3 * inner + h * a.dh;
#pragma unroll
            for (int s = 0; s < MAXN; ++s) sc[s] = 0.f;
            for (int e = e0; e < e1; e += 4) {
                const float4 q4 = *reinterpret_cast<const float4*>(q + e);
#pragma unroll
                for (int s = 0; s < MAXN; ++s)
                    if (s < n) {
                        const float4 k4 = *reinterpret_cast<const float4*>(l.qkv + s * 3 * inner + inner + h * a.dh + e);
                        sc[s] += q4.x * k4.x + q4.y * k4.y + q4.z * k4.z + q4.w * k4.w;
                    }
            }
        }
#pragma unroll
        for (int s = 0; s < MAXN; ++s) {
            float d = (pair < npair && s < n) ? sc[s] : 0.f;
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            sc[s] = d * a.scale;
        }
        if (pair < npair && sub == 0) {
            float m = -INFINITY, sum = 0.f;
            for (int s = 0; s < n; ++s) m = fmaxf(m, sc[s]);
            for (int s = 0; s < n; ++s) { sc[s] = expf(sc[s] - m); sum += sc[s]; }
            const float inv = 1.f / sum;
            for (int s = 0; s < n; ++s) l.p[(h * n + t) * n + s] = sc[s] * inv;
        }
    }
    __syncthreads();
    ENC_TF(3);
    // o[t][h*dh+e] = sum_s p[h][t][s] * v[s][h*dh+e]
    if (n == MAXN) {
        // a thread owns column j for all 8 tokens: v of the column once, then row t of p by two 16-byte reads (the same address for
        // the 64 lanes of a head) -- 24 LDS reads per lane instead of 128; same order of the sums
        for (int j = tid; j < inner; j += blockDim.x) {
            const int h = j / a.dh;
            float vv[MAXN];
#pragma unroll
            for (int u = 0; u < MAXN; ++u) vv[u] = l.qkv[u * 3 * inner + 2 * inner + j];
#pragma unroll
            for (int t = 0; t < MAXN; ++t) {
                const float* pr = l.p + (h * MAXN + t) * MAXN;
                const float4 p0 = *reinterpret_cast<const float4*>(pr), p1 = *reinterpret_cast<const float4*>(pr + 4);
                float s = 0.f;
                s += p0.x * vv[0]; s += p0.y * vv[1]; s += p0.z * vv[2]; s += p0.w * vv[3];
                s += p1.x * vv[4]; s += p1.y * vv[5]; s += p1.z * vv[6]; s += p1.w * vv[7];
                l.o[t * inner + j] = s;
            }
        }
    } else {
        for (int i = tid; i < n * inner; i += blockDim.x) {
            const int t = i / inner, j = i % inner, h = j / a.dh;
            float s = 0.f;
            for (int u = 0; u < n; ++u) s += l.p[(h * n + t) * n + u] * l.qkv[u * 3 * inner + 2 * inner + j];
            l.o[i] = s;
        }
    }
    __syncthreads();
    ENC_TF(4);
    // x1 = o Wo^T + bo + x      (K = inner, split)
    ksplit_rows(l.o, inner, a.wo + ps, 1, inner, inner, n, l.red);
    if (tid < n * D) l.x1[tid] = l.red[tid] + a.bo[ps + (tid & 31)] + l.x[tid];
    __syncthreads();
    ENC_TF(5);
    layer_norm_rows(l.x1, a.ln2_g + ps, a.ln2_b + ps, a.eps, n, l.x1n, l.xh2, l.st + 2 * n);
    __syncthreads();
    ENC_TF(6);
    // z = x1n W1^T + b1, h = gelu(z)
    const float* w1 = a.w1 + ps;
    for (int i = tid; i < n * mlp; i += blockDim.x) {
        const int t = i / mlp, m = i % mlp;
        float s = a.b1[ps + m];
#pragma unroll
        for (int c = 0; c < D; ++c) s += l.x1n[t * D + c] * w1[(size_t)m * D + c];
        l.z[i] = s;
        l.h[i] = gelu_erf(s);
    }
    __syncthreads();
    ENC_TF(7);
    // out = h W2^T + b2 + x1    (K = mlp <= 64)
    const float* w2 = a.w2 + ps;
    float r = 0.f;
    if (tid < n * D) {
        const int t = tid / D, c = tid % D;
        r = a.b2[ps + c] + l.x1[tid];
#pragma unroll 8
        for (int m = 0; m < mlp; ++m) r += l.h[t * mlp + m] * w2[(size_t)c * mlp + m];
    }
    __syncthreads();
    if (tid < n * D) out[tid] = r;
    __syncthreads();
    ENC_TF(8);
}

__device__ __forceinline__ void enc_fwd_body(const EncArgs& a, const int b, float* sm) {
    const int tid = threadIdx.x, n = a.n;
    const Lds l = carve(sm, n, a.heads * a.dh, a.heads, a.mlp);
    if (tid < n * D) l.x[tid] = a.x[(size_t)b * n * D + tid];
    __syncthreads();
    const int img = (int)saved_img_floats(n, a.heads * a.dh, a.heads, a.mlp);       // a multiple of 4
    for (int ly = 0; ly < a.depth; ++ly) {
        // the layer output goes to the (then idle) reduction scratch, so that the image with the layer INPUT in it
        // can be saved for the backward pass, which then has nothing to recompute (that recompute was 45 of its 96 us)
        layer_forward(a, l, ly, l.red);
        if (a.xs) {
            float4* dst = reinterpret_cast<float4*>(a.xs + ((size_t)ly * a.B + b) * img);
            const float4* src = reinterpret_cast<const float4*>(sm);
            for (int i = tid; i < img / 4; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        if (tid < n * D) l.x[tid] = l.red[tid];
        __syncthreads();
    }
    if (tid < n * D) a.y[(size_t)b * n * D + tid] = l.x[tid];
}
__global__ __launch_bounds__(ENC_THREADS) void encoder_fwd_kernel(EncArgs a) {
    extern __shared__ float sm[];
    enc_fwd_body(a, blockIdx.x, sm);
}
// Several encoder stacks in one launch (dh_encoder_batch_*): a stack occupies B workgroups of one image each -- 32 of the chip's
// 256 CUs at the bench batch -- for 26 - 60 us of dependent-latency work; DAHiTra's three levels are independent of each other,
// so their stacks share a launch (arguments by value, workgroups [first[j], first[j + 1]) run stack j).
constexpr int ENC_MAXJ = 4;
struct EncMulti {
    int n;
    int first[ENC_MAXJ + 1];
    EncArgs a[ENC_MAXJ];
};
__global__ __launch_bounds__(ENC_THREADS) void encoder_fwd_multi_kernel(EncMulti m) {
    extern __shared__ float sm[];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const EncArgs a = m.a[j];
    enc_fwd_body(a, (int)blockIdx.x - m.first[j], sm);
}

// data gradient: per image, layers in reverse, from each layer's saved forward image
__device__ __forceinline__ void enc_bwd_body(const EncArgs& a, const int b, float* sm) {
    const int tid = threadIdx.x, n = a.n, inner = a.heads * a.dh, mlp = a.mlp;
    const Lds l = carve(sm, n, inner, a.heads, mlp);
    float* g = l.red + 8 * n * D;            // gradient scratch after the forward buffers
    float* dx2 = g; g += n * D;              // gradient of the layer output
    float* dz = g; g += n * mlp;
    float* dx1n = g; g += n * D;
    float* dx1 = g; g += n * D;
    float* d_o = g; g += n * inner;
    float* dqkv = g; g += n * 3 * inner;
    float* dxn = g; g += n * D;
    float* dp = g;                           // [heads][n][n] dS
    ENC_T(10);
    if (tid < n * D) dx2[tid] = a.dy[(size_t)b * n * D + tid];
    for (int ly = a.depth - 1; ly >= 0; --ly) {
        const long ps = (long)ly * a.pstride;
        const size_t row0 = ((size_t)ly * a.B + b) * n;
        __syncthreads();
        {
            const int img = (int)saved_img_floats(n, inner, a.heads, mlp);
            const float4* src = reinterpret_cast<const float4*>(a.xs + ((size_t)ly * a.B + b) * img);
            float4* dst = reinterpret_cast<float4*>(sm);
#pragma unroll 8
            for (int i = tid; i < img / 4; i += blockDim.x) dst[i] = src[i];      // the forward's LDS image of this layer
        }
        ENC_T(0);
        __syncthreads();
        ENC_T(1);
        // ---- feed-forward backward ----
        const float* w2 = a.w2 + ps;
        for (int i = tid; i < n * mlp; i += blockDim.x) {
            const int t = i / mlp, m = i % mlp;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) s += dx2[t * D + c] * w2[(size_t)c * mlp + m];
            dz[i] = s * gelu_erf_grad(l.z[i]);
        }
        __syncthreads();
        ENC_T(2);
        const float* w1 = a.w1 + ps;
        if (tid < n * D) {
            const int t = tid / D, c = tid % D;
            float s = 0.f;
#pragma unroll 32
            for (int m = 0; m < mlp; ++m) s += dz[t * mlp + m] * w1[(size_t)m * D + c];      // (32 weight loads in flight: was 8, 5.5 us)
            dx1n[tid] = s;
        }
        __syncthreads();
        ENC_T(3);
        // LayerNorm-2 backward + residual: dx1 = rstd * (gh - mean(gh) - xh * mean(gh * xh)) + dx2
        if (tid < n * D) {
            const int t = tid / D, c = tid % D;
            const float gh = dx1n[tid] * a.ln2_g[ps + c], xh = l.xh2[tid];
            float sa = gh, sb = gh * xh;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
            dx1[tid] = l.st[2 * n + t * 2 + 1] * (gh - (sa + xh * sb) * (1.f / D)) + dx2[tid];
        }
        __syncthreads();
        ENC_T(4);
        // ---- attention backward ----
        // do[t][j] = sum_c dx1[t][c] * Wo[c][j]: a thread owns column j (coalesced reads of Wo rows)
        const float* wo = a.wo + ps;
        for (int j = tid; j < inner; j += blockDim.x) {
            float w[D];
#pragma unroll
            for (int c = 0; c < D; ++c) w[c] = wo[(size_t)c * inner + j];
            for (int t = 0; t < n; ++t) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < D; c += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(dx1 + t * D + c);
                    s += x.x * w[c]; s += x.y * w[c + 1]; s += x.z * w[c + 2]; s += x.w * w[c + 3];
                }
                d_o[t * inner + j] = s;
            }
        }
        __syncthreads();
        ENC_T(5);
        // dS[h][t][s] = p * (dP - sum_s' dP p),  dP[t][s] = <do_t, v_s>: 4 lanes per (h, t)
        {
            const int pair = tid >> 2, sub = tid & 3, npair = a.heads * n;
            const int h = pair / n, t = pair % n;
            const int e0 = sub * (a.dh >> 2), e1 = e0 + (a.dh >> 2);
            float dpv[MAXN];
#pragma unroll
            for (int s = 0; s < MAXN; ++s) dpv[s] = 0.f;
            if (pair < npair) {
                // 16-byte LDS reads, as in the forward's dots (scalar: 256 dependent-latency reads per lane, 10.5 us)
                const float* dor = d_o + t * inner + h * a.dh;
                for (int e = e0; e < e1; e += 4) {
                    const float4 d4 = *reinterpret_cast<const float4*>(dor + e);
#pragma unroll
                    for (int s = 0; s < MAXN; ++s)
                        if (s < n) {
                            const float4 v4 = *reinterpret_cast<const float4*>(l.qkv + s * 3 * inner + 2 * inner + h * a.dh + e);
                            dpv[s] += d4.x * v4.x + d4.y * v4.y + d4.z * v4.z + d4.w * v4.w;
                        }
                }
            }
#pragma unroll
            for (int s = 0; s < MAXN; ++s) {
                float d = (pair < npair && s < n) ? dpv[s] : 0.f;
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                dpv[s] = d;
            }
            if (pair < npair && sub == 0) {
                float dot = 0.f;
                for (int s = 0; s < n; ++s) dot += dpv[s] * l.p[(h * n + t) * n + s];
                for (int s = 0; s < n; ++s) dp[(h * n + t) * n + s] = l.p[(h * n + t) * n + s] * (dpv[s] - dot);
            }
        }
        __syncthreads();
        ENC_T(6);
        if (n == MAXN) {
            // a thread owns column j for all 8 tokens: q, k, do of the column once (24 reads), then row r of dS and of p (two 16-byte
            // reads each, the same address for the 64 lanes of a head): dq[r] from the row, dk / dv accumulate over r in the same
            // order as the per-element form below (384 scalar reads per lane, 8 us)
            for (int j = tid; j < inner; j += blockDim.x) {
                const int h = j / a.dh;
                float qv[MAXN], kv[MAXN], dov[MAXN], dk[MAXN], dv[MAXN];
#pragma unroll
                for (int u = 0; u < MAXN; ++u) {
                    qv[u] = l.qkv[u * 3 * inner + j]; kv[u] = l.qkv[u * 3 * inner + inner + j]; dov[u] = d_o[u * inner + j];
                    dk[u] = 0.f; dv[u] = 0.f;
                }
#pragma unroll
                for (int r = 0; r < MAXN; ++r) {
                    float ds[MAXN], pr[MAXN];
                    const float* dsr = dp + (h * MAXN + r) * MAXN;
                    const float* prr = l.p + (h * MAXN + r) * MAXN;
#pragma unroll
                    for (int u = 0; u < MAXN; u += 4) {
                        const float4 x = *reinterpret_cast<const float4*>(dsr + u), y = *reinterpret_cast<const float4*>(prr + u);
                        ds[u] = x.x; ds[u + 1] = x.y; ds[u + 2] = x.z; ds[u + 3] = x.w;
                        pr[u] = y.x; pr[u + 1] = y.y; pr[u + 2] = y.z; pr[u + 3] = y.w;
                    }
                    float dq = 0.f;
#pragma unroll
                    for (int u = 0; u < MAXN; ++u) {
                        dq += ds[u] * kv[u];              // dS[r][u] k_u
                        dk[u] += ds[u] * qv[r];           // dS[r][u] q_r   -> dk[u], summed over r
                        dv[u] += pr[u] * dov[r];          // p[r][u] do_r   -> dv[u]
                    }
                    dqkv[r * 3 * inner + j] = dq * a.scale;
                }
#pragma unroll
                for (int u = 0; u < MAXN; ++u) {
                    dqkv[u * 3 * inner + inner + j] = dk[u] * a.scale;
                    dqkv[u * 3 * inner + 2 * inner + j] = dv[u];
                }
            }
        } else {
            for (int i = tid; i < n * inner; i += blockDim.x) {
                const int t = i / inner, j = i % inner, h = j / a.dh;
                float dq = 0.f, dk = 0.f, dv = 0.f;
                for (int s = 0; s < n; ++s) {
                    dq += dp[(h * n + t) * n + s] * l.qkv[s * 3 * inner + inner + j];      // dS[t][s] k_s
                    dk += dp[(h * n + s) * n + t] * l.qkv[s * 3 * inner + j];              // dS[s][t] q_s
                    dv += l.p[(h * n + s) * n + t] * d_o[s * inner + j];                   // p[s][t] do_s
                }
                dqkv[t * 3 * inner + j] = dq * a.scale;
                dqkv[t * 3 * inner + inner + j] = dk * a.scale;
                dqkv[t * 3 * inner + 2 * inner + j] = dv;
            }
        }
        __syncthreads();
        ENC_T(7);
        // dxn = dqkv Wqkv   (K = 3*inner, split)
        ksplit_rows(dqkv, 3 * inner, a.wqkv + ps, D, 1, 3 * inner, n, l.red);
        if (tid < n * D) dxn[tid] = l.red[tid];
        __syncthreads();
        ENC_T(8);
        // LayerNorm-1 backward + residual -> gradient of the layer input (next iteration's dx2)
        float dxin = 0.f;
        if (tid < n * D) {
            const int t = tid / D, c = tid % D;
            const float gh = dxn[tid] * a.ln1_g[ps + c], xh = l.xh1[tid];
            float sa = gh, sb = gh * xh;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
            dxin = l.st[t * 2 + 1] * (gh - (sa + xh * sb) * (1.f / D)) + dx1[tid];
        }
        // ---- operands of the weight-gradient launch + per-image LayerNorm parameter gradients ----
        for (int i = tid; i < n * D; i += blockDim.x) {
            a.r_xn[row0 * D + i] = l.xn[i];
            a.r_dx1[row0 * D + i] = dx1[i];
            a.r_x1n[row0 * D + i] = l.x1n[i];
            a.r_dx2[row0 * D + i] = dx2[i];
        }
        for (int i = tid; i < n * 3 * inner; i += blockDim.x) a.r_dqkv[row0 * 3 * inner + i] = dqkv[i];
        for (int i = tid; i < n * inner; i += blockDim.x) a.r_o[row0 * inner + i] = l.o[i];
        for (int i = tid; i < n * mlp; i += blockDim.x) { a.r_dz[row0 * mlp + i] = dz[i]; a.r_h[row0 * mlp + i] = l.h[i]; }
        if (tid < 4 * D) {
            const int which = tid / D, c = tid % D;
            float s = 0.f;
            for (int t = 0; t < n; ++t) {
                if (which == 0) s += dxn[t * D + c] * l.xh1[t * D + c];
                else if (which == 1) s += dxn[t * D + c];
                else if (which == 2) s += dx1n[t * D + c] * l.xh2[t * D + c];
                else s += dx1n[t * D + c];
            }
            a.r_ln[(((size_t)ly * a.B + b) * 4 + which) * D + c] = s;
        }
        __syncthreads();
        ENC_T(9);
        if (tid < n * D) dx2[tid] = dxin;
    }
    __syncthreads();
    if (tid < n * D) a.y[(size_t)b * n * D + tid] = dx2[tid];
    ENC_T(11);
}
__global__ __launch_bounds__(ENC_THREADS) void encoder_bwd_kernel(EncArgs a) {
    extern __shared__ float sm[];
    enc_bwd_body(a, blockIdx.x, sm);
}
__global__ __launch_bounds__(ENC_THREADS) void encoder_bwd_multi_kernel(EncMulti m) {
    extern __shared__ float sm[];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const EncArgs a = m.a[j];
    enc_bwd_body(a, (int)blockIdx.x - m.first[j], sm);
}

// every parameter gradient of the stack in one launch: blockIdx.y = layer, a workgroup = 64 outputs x 4 row groups of
//   dWqkv[j][c] = sum_r dqkv[r][j] xn[r][c]     dWo[c][j] = sum_r dx1[r][c] o[r][j]      dbo[c] = sum_r dx1[r][c]
//   dW1[m][c]  = sum_r dz[r][m] x1n[r][c]       dW2[c][m] = sum_r dx2[r][c] h[r][m]      db1, db2 likewise
//   dln1_g/b, dln2_g/b = sum_b r_ln[b]
// (r over the R = B*n token rows), accumulated into the gradient arena.
struct EncWgArgs {
    const float *r_xn, *r_dqkv, *r_o, *r_dx1, *r_x1n, *r_dz, *r_h, *r_dx2, *r_ln;
    float *dln1_g, *dln1_b, *dwqkv, *dwo, *dbo, *dln2_g, *dln2_b, *dw1, *db1, *dw2, *db2;     // layer 0 of the grad arena
    long pstride;
    int B, n, inner, mlp;
};
__device__ __forceinline__ void enc_wgrad_body(const EncWgArgs& a, const int bx, const int ly, float (*red)[64]) {
    const int R = a.B * a.n, inner = a.inner, mlp = a.mlp;
    const long ps = (long)ly * a.pstride;
    const long nq = 3L * inner * D, no = (long)D * inner, n1 = (long)mlp * D, n2 = (long)D * mlp;
    const long total = nq + no + n1 + n2 + D + mlp + D + 4 * D;
    const int ol = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const long i = (long)bx * 64 + ol;
    const float *A = nullptr, *Bm = nullptr;
    int wa = 0, wb = 0, ia = 0, ib = 0, mode = 0;      // mode 0: nothing, 1: product, 2: column sum, 3: LayerNorm partials
    float* dst = nullptr;
    long e = i;
    if (i < total) {
        if (e < nq) { mode = 1; A = a.r_dqkv; wa = 3 * inner; ia = (int)(e / D); Bm = a.r_xn; wb = D; ib = (int)(e % D); dst = a.dwqkv + ps + e; }
        else if ((e -= nq) < no) { mode = 1; A = a.r_dx1; wa = D; ia = (int)(e / inner); Bm = a.r_o; wb = inner; ib = (int)(e % inner); dst = a.dwo + ps + e; }
        else if ((e -= no) < n1) { mode = 1; A = a.r_dz; wa = mlp; ia = (int)(e / D); Bm = a.r_x1n; wb = D; ib = (int)(e % D); dst = a.dw1 + ps + e; }
        else if ((e -= n1) < n2) { mode = 1; A = a.r_dx2; wa = D; ia = (int)(e / mlp); Bm = a.r_h; wb = mlp; ib = (int)(e % mlp); dst = a.dw2 + ps + e; }
        else if ((e -= n2) < D) { mode = 2; A = a.r_dx1; wa = D; ia = (int)e; dst = a.dbo + ps + e; }
        else if ((e -= D) < mlp) { mode = 2; A = a.r_dz; wa = mlp; ia = (int)e; dst = a.db1 + ps + e; }
        else if ((e -= mlp) < D) { mode = 2; A = a.r_dx2; wa = D; ia = (int)e; dst = a.db2 + ps + e; }
        else {
            e -= D;
            mode = 3;
            const int which = (int)(e / D), c = (int)(e % D);
            ia = which; ib = c;
            float* out = which == 0 ? a.dln1_g : (which == 1 ? a.dln1_b : (which == 2 ? a.dln2_g : a.dln2_b));
            dst = out + ps + c;
        }
    }
    float s = 0.f;
    if (mode == 1) {
        A += (size_t)ly * R * wa;
        Bm += (size_t)ly * R * wb;
        // (16 rows' loads in flight: the plain loop is 64 rounds of dependent L2 latency, 22 us for a 12 MFLOP launch)
#pragma unroll 16
        for (int r = rg; r < R; r += 4) s += A[(size_t)r * wa + ia] * Bm[(size_t)r * wb + ib];
    } else if (mode == 2) {
        A += (size_t)ly * R * wa;
#pragma unroll 16
        for (int r = rg; r < R; r += 4) s += A[(size_t)r * wa + ia];
    } else if (mode == 3) {
        for (int b = rg; b < a.B; b += 4) s += a.r_ln[(((size_t)ly * a.B + b) * 4 + ia) * D + ib];
    }
    red[rg][ol] = s;
    __syncthreads();
    if (rg == 0 && mode != 0) *dst += red[0][ol] + red[1][ol] + red[2][ol] + red[3][ol];
}
__global__ __launch_bounds__(256) void encoder_wgrad_kernel(EncWgArgs a) {
    __shared__ float red[4][64];
    enc_wgrad_body(a, blockIdx.x, blockIdx.y, red);
}
struct EncWgMulti {
    int depth[ENC_MAXJ], nbx[ENC_MAXJ];
    EncWgArgs a[ENC_MAXJ];
};
__global__ __launch_bounds__(256) void encoder_wgrad_multi_kernel(EncWgMulti m) {      // blockIdx.z = stack
    __shared__ float red[4][64];
    const int j = blockIdx.z;
    if ((int)blockIdx.y >= m.depth[j] || (int)blockIdx.x >= m.nbx[j]) return;      // (uniform per workgroup)
    const EncWgArgs a = m.a[j];
    enc_wgrad_body(a, blockIdx.x, blockIdx.y, red);
}

inline hipStream_t ST(void* s) { return reinterpret_cast<hipStream_t>(s); }

int set_lds(const void* kern, size_t lds, bool& done) {
    if (lds > 64 * 1024 && !done) {
        done = true;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("encoder_fused: cannot raise dynamic LDS to %zu", lds);
        }
    }
    return 0;
}

// ---- batched launches ----
struct EncBatch {
    bool on = false;
    int nf = 0, nb = 0;
    size_t flds = 0, blds = 0;
    EncMulti f, b;
    EncWgMulti g;
};
thread_local EncBatch g_eb;
int enc_batch_flush(hipStream_t st) {
    EncBatch& e = g_eb;
    static bool done_f = false, done_b = false;
    if (e.nf) {
        e.f.n = e.nf;
        if (set_lds(reinterpret_cast<const void*>(encoder_fwd_multi_kernel), 160 * 1024, done_f)) return 1;
        hipLaunchKernelGGL(encoder_fwd_multi_kernel, dim3(e.f.first[e.nf]), dim3(ENC_THREADS), e.flds, st, e.f);
        e.nf = 0; e.flds = 0;
        DH_CHECK_LAUNCH("encoder_fwd_multi");
    }
    if (e.nb) {
        e.b.n = e.nb;
        if (set_lds(reinterpret_cast<const void*>(encoder_bwd_multi_kernel), 160 * 1024, done_b)) return 1;
        hipLaunchKernelGGL(encoder_bwd_multi_kernel, dim3(e.b.first[e.nb]), dim3(ENC_THREADS), e.blds, st, e.b);
        int mx = 0, my = 0;
        for (int j = 0; j < e.nb; ++j) { if (e.g.nbx[j] > mx) mx = e.g.nbx[j]; if (e.g.depth[j] > my) my = e.g.depth[j]; }
        hipLaunchKernelGGL(encoder_wgrad_multi_kernel, dim3(mx, my, e.nb), dim3(256), 0, st, e.g);
        e.nb = 0; e.blds = 0;
        DH_CHECK_LAUNCH("encoder_bwd_multi");
    }
    return 0;
}

}  // namespace

// Batched encoder stacks: between dh_encoder_batch_begin() and _end(), dh_encoder_fwd / dh_encoder_bwd only RECORD their
// launches (up to 4 of each direction; a fifth issues the first four); dh_encoder_batch_launch(stream) issues what has been
// recorded as one forward launch and / or one backward + one parameter-gradient launch.  Every buffer of a recorded call
// (including its own workspace) must stay alive and unchanged until then.  Per host thread.
extern "C" int dh_encoder_batch_begin() { g_eb.on = true; g_eb.nf = g_eb.nb = 0; g_eb.flds = g_eb.blds = 0; return 0; }
extern "C" int dh_encoder_batch_pending() { return g_eb.nf + g_eb.nb; }
extern "C" int dh_encoder_batch_launch(void* stream) { return enc_batch_flush(ST(stream)); }
extern "C" int dh_encoder_batch_end(void* stream) { const int rc = enc_batch_flush(ST(stream)); g_eb.on = false; return rc; }
extern "C" int dh_encoder_batch_abort() { g_eb.on = false; g_eb.nf = g_eb.nb = 0; g_eb.flds = g_eb.blds = 0; return 0; }

// largest inner width whose backward working set (forward buffers + gradient scratch) fits the 160 KB LDS
extern "C" int dh_encoder_supported(int n, int heads, int dim_head, int mlp) {
    if (n < 1 || n > MAXN || mlp > 64 || mlp < 1 || heads < 1 || dim_head % 16) return 0;      // (16-byte LDS reads of a quarter head)
    const int inner = heads * dim_head;
    const size_t bwd = (fwd_lds_floats(n, inner, heads, mlp) + (size_t)n * D * 4 + (size_t)n * mlp + (size_t)n * inner +
                        (size_t)n * 3 * inner + (size_t)heads * n * n) * 4;
    return bwd <= 160 * 1024 && heads * n * 4 <= 256 ? 1 : 0;
}

// x, y: [B][n][32] fp32 tokens (n = 2 * token_len <= 8); parameters: layer 0 pointers in torch layouts (to_qkv
// [3*inner][32], to_out [32][inner], net.0 [mlp][32], net.3 [32][mlp]), consecutive layers param_stride floats apart.
// saved_inputs (optional, needed for the backward): [depth][B][n][32].
extern "C" int dh_encoder_fwd(const float* x, float* y, float* saved_inputs, int B, int n, int depth, int heads,
                              int dim_head, int mlp, float scale, float eps, long param_stride, const float* ln1_g,
                              const float* ln1_b, const float* wqkv, const float* wo, const float* bo,
                              const float* ln2_g, const float* ln2_b, const float* w1, const float* b1,
                              const float* w2, const float* b2, void* stream) {
    DH_REQUIRE(dh_encoder_supported(n, heads, dim_head, mlp) && depth >= 1 && B >= 1,
               "encoder_fwd: n=%d depth=%d mlp=%d heads=%d dim_head=%d unsupported", n, depth, mlp, heads, dim_head);
    EncArgs a = {};
    a.x = x; a.y = y; a.xs = saved_inputs; a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.wqkv = wqkv; a.wo = wo; a.bo = bo;
    a.ln2_g = ln2_g; a.ln2_b = ln2_b; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.pstride = param_stride;
    a.depth = depth; a.B = B; a.n = n; a.heads = heads; a.dh = dim_head; a.mlp = mlp; a.scale = scale; a.eps = eps;
    const size_t lds = fwd_lds_floats(n, heads * dim_head, heads, mlp) * 4;
    if (g_eb.on) {
        EncBatch& e = g_eb;
        if (e.nf == ENC_MAXJ && enc_batch_flush(ST(stream))) return 1;
        if (e.nf == 0) e.f.first[0] = 0;
        e.f.a[e.nf] = a;
        e.f.first[e.nf + 1] = e.f.first[e.nf] + B;
        if (lds > e.flds) e.flds = lds;
        ++e.nf;
        return 0;
    }
    static bool done = false;
    if (set_lds(reinterpret_cast<const void*>(encoder_fwd_kernel), lds, done)) return 1;
    hipLaunchKernelGGL(encoder_fwd_kernel, dim3(B), dim3(ENC_THREADS), lds, ST(stream), a);
    DH_CHECK_LAUNCH("encoder_fwd");
    return 0;
}

// floats of the saved_inputs buffer dh_encoder_fwd fills for dh_encoder_bwd: [depth][B][one forward image]
extern "C" long dh_encoder_saved_floats(int B, int n, int depth, int heads, int dim_head, int mlp) {
    return (long)depth * B * (long)saved_img_floats(n, heads * dim_head, heads, mlp);
}

// workspace floats: depth * B*n * (4*32 + 3*inner + inner + 2*mlp) + depth * B * 128
extern "C" long dh_encoder_bwd_workspace_size(int B, int n, int depth, int heads, int dim_head, int mlp) {
    const long R = (long)B * n, inner = (long)heads * dim_head;
    return ((long)depth * R * (4 * D + 3 * inner + inner + 2 * mlp) + (long)depth * B * 4 * D) * 4;
}

// dy, dx: [B][n][32]; saved_inputs from dh_encoder_fwd; gradients (layer 0 pointers into the gradient arena, same
// param_stride) are ACCUMULATED.
extern "C" int dh_encoder_bwd(const float* dy, float* dx, const float* saved_inputs, int B, int n, int depth, int heads,
                              int dim_head, int mlp, float scale, float eps, long param_stride, const float* ln1_g,
                              const float* ln1_b, const float* wqkv, const float* wo, const float* bo,
                              const float* ln2_g, const float* ln2_b, const float* w1, const float* b1,
                              const float* w2, const float* b2, float* dln1_g, float* dln1_b, float* dwqkv, float* dwo,
                              float* dbo, float* dln2_g, float* dln2_b, float* dw1, float* db1, float* dw2, float* db2,
                              void* workspace, void* stream) {
    DH_REQUIRE(dh_encoder_supported(n, heads, dim_head, mlp) && depth >= 1 && B >= 1,
               "encoder_bwd: n=%d depth=%d mlp=%d heads=%d dim_head=%d unsupported", n, depth, mlp, heads, dim_head);
    const int inner = heads * dim_head;
    const long R = (long)B * n;
    EncArgs a = {};
    a.dy = dy; a.y = dx; a.xs = const_cast<float*>(saved_inputs);
    a.ln1_g = ln1_g; a.ln1_b = ln1_b; a.wqkv = wqkv; a.wo = wo; a.bo = bo;
    a.ln2_g = ln2_g; a.ln2_b = ln2_b; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.pstride = param_stride;
    a.depth = depth; a.B = B; a.n = n; a.heads = heads; a.dh = dim_head; a.mlp = mlp; a.scale = scale; a.eps = eps;
    float* w = reinterpret_cast<float*>(workspace);
    a.r_xn = w; w += depth * R * D;
    a.r_dx1 = w; w += depth * R * D;
    a.r_x1n = w; w += depth * R * D;
    a.r_dx2 = w; w += depth * R * D;
    a.r_dqkv = w; w += depth * R * 3 * inner;
    a.r_o = w; w += depth * R * inner;
    a.r_dz = w; w += depth * R * mlp;
    a.r_h = w; w += depth * R * mlp;
    a.r_ln = w;
    // forward buffers + gradient scratch: dx2, dx1n, dx1, dxn, dz, do, dqkv, dS
    const size_t lds = (fwd_lds_floats(n, inner, heads, mlp) + (size_t)n * D * 4 + (size_t)n * mlp + (size_t)n * inner +
                        (size_t)n * 3 * inner + (size_t)heads * n * n) * 4;
    static bool done = false;
    if (!g_eb.on) {
        if (set_lds(reinterpret_cast<const void*>(encoder_bwd_kernel), lds, done)) return 1;
        hipLaunchKernelGGL(encoder_bwd_kernel, dim3(B), dim3(ENC_THREADS), lds, ST(stream), a);
    }
    EncWgArgs g = {};
    g.r_xn = a.r_xn; g.r_dqkv = a.r_dqkv; g.r_o = a.r_o; g.r_dx1 = a.r_dx1; g.r_x1n = a.r_x1n; g.r_dz = a.r_dz;
    g.r_h = a.r_h; g.r_dx2 = a.r_dx2; g.r_ln = a.r_ln;
    g.dln1_g = dln1_g; g.dln1_b = dln1_b; g.dwqkv = dwqkv; g.dwo = dwo; g.dbo = dbo; g.dln2_g = dln2_g; g.dln2_b = dln2_b;
    g.dw1 = dw1; g.db1 = db1; g.dw2 = dw2; g.db2 = db2; g.pstride = param_stride; g.B = B; g.n = n; g.inner = inner; g.mlp = mlp;
    const long total = 3L * inner * D + (long)D * inner + 2L * mlp * D + D + mlp + D + 4 * D;
    if (g_eb.on) {
        EncBatch& e = g_eb;
        if (e.nb == ENC_MAXJ && enc_batch_flush(ST(stream))) return 1;
        if (e.nb == 0) e.b.first[0] = 0;
        e.b.a[e.nb] = a;
        e.b.first[e.nb + 1] = e.b.first[e.nb] + B;
        e.g.a[e.nb] = g; e.g.depth[e.nb] = depth; e.g.nbx[e.nb] = dh_cdiv(total, 64);
        if (lds > e.blds) e.blds = lds;
        ++e.nb;
        return 0;
    }
    hipLaunchKernelGGL(encoder_wgrad_kernel, dim3(dh_cdiv(total, 64), depth), dim3(256), 0, ST(stream), g);
    DH_CHECK_LAUNCH("encoder_bwd");
    return 0;
}

#ifdef ENC_TIMING
extern "C" int dh_debug_enct(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_enct), (size_t)n * 8); }
#endif
