// HBM-bound NHWC pointwise / resampling / layout kernels (gfx950).
// Reference ops restated: MaxPool2d(3,2,1) models/resnet.py:154; nn.Upsample(x2 nearest) and
// nn.Upsample(x4 bilinear, align_corners=False) models/networks.py:199-200; torch.abs(x1-x2)
// models/networks.py:384; GELU / ReLU derivatives; channel concat (torch.cat) models/networks.py:1312,1348.
// Every kernel is a grid-stride loop over 4-channel vectors (16 B fp32 / 8 B bf16 per lane).
#include "common.h"

namespace {

inline int ew_grid(long n, int block) {
    long g = (n + block - 1) / block;
    return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
#define GSL(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---- NCHW fp32 <-> NHWC T -------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, long HW, int CP) {
    const long total = (long)N * HW * CP;      // destination channels zero-padded to CP >= C
    GSL(i, total) {
        const int c = (int)(i % CP);
        const long p = (i / CP) % HW;
        const long n = i / (CP * HW);
        stf(dst + i, c < C ? src[(n * C + c) * HW + p] : 0.f);
    }
}
// the class-gradient case (C = n_class real channels into ONE 16-byte piece per pixel): a thread per pixel reads its C
// planes (each coalesced along the pixels) and stores one piece -- the generic form above stored 2-byte elements
template <typename T>
__global__ void nchw_to_nhwc_piece_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, long HW) {
    constexpr int V = V16<T>::N;
    const long total = (long)N * HW;
    GSL(i, total) {
        const long p = i % HW, n = i / HW;
        float v[V];
#pragma unroll
        for (int c = 0; c < V; ++c) v[c] = c < C ? src[(n * C + c) * HW + p] : 0.f;
        stv(dst + i * V, v);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, long HW) {
    const long total = (long)N * HW * C;
    GSL(i, total) {   // i indexes the NCHW destination (coalesced stores)
        const long p = i % HW;
        const int c = (int)((i / HW) % C);
        const long n = i / (HW * C);
        dst[i] = ldf(src + (n * HW + p) * C + c);
    }
}

// ---- data gradient of the class head: 3x3 / stride 1 / pad 1 convolution 32 -> NC channels (NC = n_class <= 8) ----
//   dX[n][y][x][ci] = sum_{kh,kw} sum_{co<NC} dY[n][y+1-kh][x+1-kw][co] * W[co][ci][kh][kw]
// (models/help_funcs.py:13-14 / networks.py:1247 backward).  With 2..5 real output channels the reduction is 18..45
// long: as an MFMA convolution it would run on a dY zero-padded to a 32-channel K-chunk (16x the bytes and the
// FLOPs); here dY stays [pixel][CP] with CP = one 16-byte piece, one lane owns a pixel and all 32 input channels,
// the weights are LDS broadcasts.  HBM-bound: 16 B read (x9 from L1/L2) + 64/128 B written per pixel.
template <typename T, int CP>
__global__ __launch_bounds__(256) void head_dgrad3x3_kernel(const T* __restrict__ dy, const float* __restrict__ w_oihw,
                                                            T* __restrict__ dx, int N, int H, int W, int NC) {
    __shared__ __attribute__((aligned(16))) float sw[9 * 8 * 32];          // [tap][co][ci]
    for (int i = threadIdx.x; i < 9 * NC * 32; i += blockDim.x) {
        const int ci = i & 31, co = (i >> 5) % NC, tap = i / (32 * NC);
        sw[i] = w_oihw[((size_t)co * 32 + ci) * 9 + tap];
    }
    __syncthreads();
    const long total = (long)N * H * W;
    for (long px = (long)blockIdx.x * blockDim.x + threadIdx.x; px < total; px += (long)gridDim.x * blockDim.x) {
        const int x = (int)(px % W), y = (int)((px / W) % H);
        const long n = px / ((long)W * H);
        float acc[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int yy = y + 1 - kh, xx = x + 1 - kw;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                float v[CP];
                const T* src = dy + ((n * H + yy) * W + xx) * CP;
#pragma unroll
                for (int q = 0; q < CP; q += 4) {
                    float t4[4];
                    ld4(src + q, t4);
                    v[q] = t4[0]; v[q + 1] = t4[1]; v[q + 2] = t4[2]; v[q + 3] = t4[3];
                }
                for (int co = 0; co < NC; ++co) {
                    const float* wr = sw + ((kh * 3 + kw) * NC + co) * 32;
                    const float d = v[co];
#pragma unroll
                    for (int c = 0; c < 32; c += 4) {
                        const float4 ww = *reinterpret_cast<const float4*>(wr + c);
                        acc[c] += d * ww.x; acc[c + 1] += d * ww.y; acc[c + 2] += d * ww.z; acc[c + 3] += d * ww.w;
                    }
                }
            }
        T* dst = dx + px * 32;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            float r4[4] = {acc[c], acc[c + 1], acc[c + 2], acc[c + 3]};
            st4(dst + c, r4);
        }
    }
}

// n_class <= 2 (the change-detection heads): the 9 x NC x (channels of one piece) weights fit the registers of a lane that
// owns ONE 16-byte output piece of a pixel -- no LDS (the form above issues 144 ds_read_b128 per pixel, which bounds it),
// 16-byte coalesced stores (it stored 8-byte elements at a 64-byte lane stride).  HBM-bound: 16 B read, 64 / 128 B
// written per pixel.
template <typename T, int CP, int NC>
__global__ __launch_bounds__(256) void head_dgrad3x3_reg_kernel(const T* __restrict__ dy, const float* __restrict__ w_oihw,
                                                                T* __restrict__ dx, int N, int H, int W) {
    constexpr int V = V16<T>::N;           // channels of one output piece
    constexpr int LPP = 32 / V;            // lanes per pixel
    const int cg = threadIdx.x % LPP;
    // weights: OIHW global -> LDS [tap][co][ci] once per workgroup (coalesced), then 9 * NC * V registers per lane
    // (every lane fetching its 144 values from global cost more than the whole rest of the kernel: 141 us)
    __shared__ __attribute__((aligned(16))) float sw[9 * NC * 32];
    for (int i = threadIdx.x; i < 9 * NC * 32; i += blockDim.x) {
        const int ci = i & 31, co = (i >> 5) % NC, tap = i / (32 * NC);
        sw[i] = w_oihw[((size_t)co * 32 + ci) * 9 + tap];
    }
    __syncthreads();
    float w[9][NC][V];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int co = 0; co < NC; ++co)
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                const float4 q = *reinterpret_cast<const float4*>(sw + (t * NC + co) * 32 + cg * V + j);
                w[t][co][j] = q.x; w[t][co][j + 1] = q.y; w[t][co][j + 2] = q.z; w[t][co][j + 3] = q.w;
            }
    const long total = (long)N * H * W * LPP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long px = i / LPP;
        const int x = (int)(px % W), y = (int)((px / W) % H);
        const long n = px / ((long)W * H);
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int yy = y + 1 - kh, xx = x + 1 - kw;
                const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
                float v[4];
                ld4(dy + ((n * H + (ok ? yy : y)) * W + (ok ? xx : x)) * CP, v);      // the first 4 channels hold the NC real ones
#pragma unroll
                for (int co = 0; co < NC; ++co) {
                    const float d = ok ? v[co] : 0.f;
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += d * w[kh * 3 + kw][co][j];
                }
            }
        stv(dx + i * V, acc);
    }
}

// ---- input pipeline on the device: crop + flips + normalise of pre-decoded uint8 pairs --------------------------------
// What CDDataAugmentation.transform (datasets/data_utils.py:55-111) does per sample on the host with PIL -- crop window,
// horizontal / vertical flip, ToTensor + Normalize(0.5, 0.5) -- for a whole batch in one pass: a PIL loader delivers a few
// hundred pairs/s per worker, the train step consumes ~7 000.  src images [S][H][W][3] uint8 (A and B), labels [S][H][W]
// uint8; sample n takes source pair idx[n] with params[n] = {x0, y0, hflip, vflip}; outputs A / B fp32 [N][3][h][w] in
// [-1, 1] and L uint8 [N][1][h][w].  (The reference's random Gaussian blur has no counterpart here.)
__global__ void augment_pairs_u8_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                        const unsigned char* __restrict__ l, const int* __restrict__ idx,
                                        const int* __restrict__ params, int N, int H, int W, int h, int w,
                                        float* __restrict__ oa, float* __restrict__ ob, unsigned char* __restrict__ ol) {
    const long total = (long)N * h * w;
    GSL(i, total) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const int n = (int)(i / ((long)w * h));
        const int* pr = params + n * 4;
        const int sx = pr[0] + (pr[2] ? w - 1 - x : x), sy = pr[1] + (pr[3] ? h - 1 - y : y);
        const long sp = ((long)idx[n] * H + sy) * W + sx;
        const long plane = (long)h * w, o = (long)n * 3 * plane + (long)y * w + x;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            oa[o + c * plane] = ((float)a[sp * 3 + c] / 255.f - 0.5f) / 0.5f;       // ToTensor then Normalize: same roundings
            ob[o + c * plane] = ((float)b[sp * 3 + c] / 255.f - 0.5f) / 0.5f;
        }
        if (ol) ol[(long)n * plane + (long)y * w + x] = l[sp];
    }
}

// bf16, n_class <= 2: the same data gradient on the matrix cores WITHOUT padding dY in memory.  K = (tap, co) = 18 -> one
// v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 input channels: the weights are the A operand (two fragments, loaded once
// per wave), the B operand (k = 8 g + e -> tap 4 g + e / 2, co = e & 1) is gathered straight from dY with one 4-byte
// load per tap (lane group g = 3 and three quarters of g = 2 hold zeros).  A lane ends with 4 consecutive input channels of
// one pixel, twice: 8-byte stores, 64 B per pixel.  HBM-bound: 16 B read (x9 from L1 / L2) + 64 B written per pixel.
// GATE: the 32 output channels feed relu(BatchNorm(y)) (the classifier's TwoLayerConv2d, models/help_funcs.py:7-15): the ReLU
// mask is recomputed from y (y * mscale + mshift > 0), the MASKED gradient is stored and the per-workgroup partial sums
// (sum g, sum g * xhat) of that BatchNorm's backward are emitted, [2][32][gridDim.x] -- the separate reduction pass over (g, y)
// (268 MB at the bench size) disappears (dh_bn_bwd_from_partials does the rest).  A workgroup stays inside one statistics group.
// RELUREF (without GATE): the 32 channels are the output of a ReLU whose result `gy` is at hand (conv_layer2 of the
// hierarchical model, models/networks.py:1351-1355): the gradient is stored already masked (gy > 0), and the separate
// activation-backward pass over (gradient, output) -- 402 MB at the bench size -- disappears.
template <bool GATE, bool RELUREF = false>
__global__ __launch_bounds__(256) void head_dgrad3x3_mfma_kernel(const bf16* __restrict__ dy, const float* __restrict__ w_oihw,
                                                                 bf16* __restrict__ dx, int N, int H, int W, int NC,
                                                                 const bf16* __restrict__ gy, const float* __restrict__ mscale,
                                                                 const float* __restrict__ mshift, const float* __restrict__ gmean,
                                                                 const float* __restrict__ ginvstd, int groups,
                                                                 float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, pl = lane & 15, g = lane >> 4;
    // A fragments: row ci = s * 16 + pl, k = 8 g + e
    s16x8 wa[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int tap = 4 * g + (e >> 1), co = e & 1;
            v[e] = (tap < 9 && co < NC) ? w_oihw[((size_t)co * 32 + s * 16 + pl) * 9 + tap] : 0.f;
        }
        union { uint4 u; s16x8 h; } pk;
        pk.u = make_uint4(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]), f2bf2(v[4], v[5]), f2bf2(v[6], v[7]));
        wa[s] = pk.h;
    }
    // workgroups [bg * gridDim.x / groups, ...) walk the pixels of statistics group bg (groups = 1 without GATE)
    const int ng = GATE ? groups : 1, bpg = gridDim.x / ng, bg = blockIdx.x / bpg;
    const long gpix = (long)N * H * W / ng, total = (bg + 1) * gpix, ngrp16 = (gpix + 15) / 16;
    const long wave0 = ((long)(blockIdx.x - bg * bpg) * blockDim.x + threadIdx.x) >> 6, nwaves = ((long)bpg * blockDim.x) >> 6;
    float cs[2][4], ch_[2][4], cm[2][4], ci[2][4], s1[2][4], s2[2][4];
    if constexpr (GATE) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = bg * 32 + s * 16 + g * 4 + j;
                cs[s][j] = mscale[c]; ch_[s][j] = mshift[c]; cm[s][j] = gmean[c]; ci[s][j] = ginvstd[c];
                s1[s][j] = s2[s][j] = 0.f;
            }
    }
    for (long grp = wave0; grp < ngrp16; grp += nwaves) {
        const long px = bg * gpix + grp * 16 + pl;
        const bool inb = px < total;
        const int x = (int)(px % W), y = (int)((px / W) % H);
        const long n = px / ((long)W * H);
        unsigned bk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int tap = 4 * g + q, kh = tap / 3, kw = tap - kh * 3;
            const int yy = y + 1 - kh, xx = x + 1 - kw;
            const bool ok = inb && tap < 9 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const unsigned v = *reinterpret_cast<const unsigned*>(dy + ((n * H + (ok ? yy : y)) * W + (ok ? xx : x)) * 8 * (inb ? 1 : 0));
            bk[q] = ok ? v : 0u;
        }
        union { uint4 u; s16x8 h; } b;
        b.u = make_uint4(bk[0], bk[1], bk[2], bk[3]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[s], b.h, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            if (inb) {
                float r[4] = {d[0], d[1], d[2], d[3]};
                if constexpr (GATE) {
                    float yv[4];
                    ld4(gy + px * 32 + s * 16 + g * 4, yv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        r[j] = (yv[j] * cs[s][j] + ch_[s][j]) > 0.f ? r[j] : 0.f;
                        s1[s][j] += r[j];
                        s2[s][j] += r[j] * ((yv[j] - cm[s][j]) * ci[s][j]);
                    }
                }
                if constexpr (RELUREF && !GATE) {
                    float yv[4];
                    ld4(gy + px * 32 + s * 16 + g * 4, yv);
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = yv[j] > 0.f ? r[j] : 0.f;
                }
                st4(dx + px * 32 + s * 16 + g * 4, r);
            }
        }
    }
    if constexpr (GATE) {
        __shared__ float red[4][2][32];
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row16_sum(s1[s][j]), b = row16_sum(s2[s][j]);
                if (pl == 0) { red[wv][0][s * 16 + g * 4 + j] = a; red[wv][1][s * 16 + g * 4 + j] = b; }
            }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
            partial[((size_t)which * 32 + c) * gridDim.x + blockIdx.x] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
        }
    }
}

// ---- channel-slice copy: dst[p, dc0 + c] = src[p, sc0 + c], c < Cn -------------------------
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int Cs, int sc0, T* __restrict__ dst, int Cd,
                                     int dc0, int Cn, long P) {
    const int vn = Cn / 4;
    GSL(i, P * vn) {
        const long p = i / vn;
        const int c = (int)(i % vn) * 4;
        float v[4];
        ld4(src + p * Cs + sc0 + c, v);
        st4(dst + p * Cd + dc0 + c, v);
    }
}
// the same with one 16-byte piece per lane (every channel count and offset a multiple of the piece)
template <typename T>
__global__ void copy_channels16_kernel(const T* __restrict__ src, int Cs, int sc0, T* __restrict__ dst, int Cd,
                                       int dc0, int Cn, long P) {
    constexpr int V = V16<T>::N;
    const int vn = Cn / V;
    GSL(i, P * vn) {
        const long p = i / vn;
        const int c = (int)(i % vn) * V;
        *reinterpret_cast<uint4*>(dst + p * Cd + dc0 + c) = *reinterpret_cast<const uint4*>(src + p * Cs + sc0 + c);
    }
}

// both halves of torch.cat([t[:P], t[P:]], channel) in one launch: cat[p, h * C + c] = t[h * P + p, c] (INVERSE: the other
// way round, the two batch halves of the concatenation's gradient); one 16-byte piece per lane
#define GSLB(i, n, bid, nb) for (long i = (long)(bid) * blockDim.x + threadIdx.x; i < (n); i += (long)(nb) * blockDim.x)
template <typename T, bool INVERSE>
__device__ __forceinline__ void cat_halves_body(T* __restrict__ t, T* __restrict__ cat, int C, long P, int bid, int nb) {
    constexpr int V = V16<T>::N;
    const int vn = 2 * C / V;
    GSLB(i, P * vn, bid, nb) {
        const long p = i / vn;
        const int c2 = (int)(i % vn) * V, h = c2 >= C, c = c2 - h * C;
        uint4* a = reinterpret_cast<uint4*>(t + (h * P + p) * C + c);
        uint4* b = reinterpret_cast<uint4*>(cat + p * 2 * C + c2);
        if (INVERSE) *a = *b;
        else *b = *a;
    }
}
template <typename T, bool INVERSE>
__global__ void cat_halves_kernel(T* __restrict__ t, T* __restrict__ cat, int C, long P) {
    cat_halves_body<T, INVERSE>(t, cat, C, P, blockIdx.x, gridDim.x);
}

// ---- y = a + b ------------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long nvec) {
    GSL(i, nvec) {
        float u[4], v[4];
        ld4(a + i * 4, u);
        ld4(b + i * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] += v[j];
        st4(y + i * 4, u);
    }
}
template <typename T>
__global__ void add16_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long npiece) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    GSL(i, npiece) {
        float u[V], v[V];
        ldv(a + i * V, u);
        ldv(b + i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) u[j] += v[j];
        stv(y + i * V, u);
    }
}

// y[n,p,c] = x[n,p,c] + pos[c,p]  (pos is an NCHW fp32 parameter [1,C,h,w]; networks.py:1291)
template <typename T>
__global__ void add_pos_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y, int N,
                               long HW, int C) {
    GSL(i, (long)N * HW * C) {
        const int c = (int)(i % C);
        const long p = (i / C) % HW;
        stf(y + i, ldf(x + i) + pos[c * HW + p]);
    }
}
template <typename T>
__device__ __forceinline__ void add_pos16_body(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y, int N,
                                               long HW, int C, int bid, int nb) {
    constexpr int V = V16<T>::N;      // one 16-byte piece of x per lane; its V embedding values are L2-resident gathers
    const int vn = C / V;
    GSLB(i, (long)N * HW * vn, bid, nb) {
        const int c = (int)(i % vn) * V;
        const long p = (i / vn) % HW;
        float v[V];
        ldv(x + i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += pos[(long)(c + j) * HW + p];
        stv(y + i * V, v);
    }
}
template <typename T>
__global__ void add_pos16_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y, int N,
                                 long HW, int C) {
    add_pos16_body<T>(x, pos, y, N, HW, C, blockIdx.x, gridDim.x);
}
// dpos[c,p] (+)= sum_n dy[n,p,c]
template <typename T>
__global__ void add_pos_bwd_kernel(const T* __restrict__ dy, float* __restrict__ dpos, int N, long HW, int C,
                                   int accumulate) {
    GSL(i, HW * C) {   // i indexes dpos (c-major)
        const long p = i % HW;
        const int c = (int)(i / HW);
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += ldf(dy + ((long)n * HW + p) * C + c);
        if (accumulate) dpos[i] += s; else dpos[i] = s;
    }
}
// the same for bf16 rows of C = 32 channels: a lane owns one 16-byte piece (8 channels) of a pixel and a quarter of the images
// (4 loads in flight), the four image groups of a workgroup meet in LDS; 16 pixels per 256-thread workgroup.  (One lane per
// (channel, pixel) with 2-byte loads 64 bytes apart and N dependent iterations: 17.6 us for 16.8 MB.)
__device__ __forceinline__ void add_pos_bwd32_body(const bf16* __restrict__ dy, float* __restrict__ dpos, int N, long HW,
                                                   int accumulate, int bid) {
    constexpr int C = 32;
    __shared__ float red[4][64][8];
    const int tid = threadIdx.x, q = tid & 3, px = (tid >> 2) & 15, ng = tid >> 6;
    const long p = (long)bid * 16 + px;
    float s[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[u][j] = 0.f;
    if (p < HW) {
        const int per = (N + 3) / 4, n0 = ng * per, n1 = n0 + per < N ? n0 + per : N;
        auto row = [&](int n, float (&acc)[8]) {
            float v[8];
            unpack16(*reinterpret_cast<const uint4*>(dy + ((long)n * HW + p) * C + q * 8), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        };
        int n = n0;
        for (; n + 3 < n1; n += 4) { row(n, s[0]); row(n + 1, s[1]); row(n + 2, s[2]); row(n + 3, s[3]); }
        for (int u = 0; n < n1; ++n, ++u) row(n, s[u & 3]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ng][tid & 63][j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
    __syncthreads();
    if (ng == 0 && p < HW) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = ((red[0][tid][j] + red[1][tid][j]) + red[2][tid][j]) + red[3][tid][j];
            float* o = dpos + (long)(q * 8 + j) * HW + p;
            if (accumulate) *o += t; else *o = t;
        }
    }
}
__global__ __launch_bounds__(256) void add_pos_bwd32_kernel(const bf16* __restrict__ dy, float* __restrict__ dpos, int N, long HW,
                                                            int accumulate) {
    add_pos_bwd32_body(dy, dpos, N, HW, accumulate, blockIdx.x);
}

// ---- activation derivative: dx = dy * f'(.) ---------------------------------------------------
// mode RELU: ref = post-activation output (mask ref > 0); mode GELU: ref = pre-activation.
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ dx, long nvec,
                               int act) {
    GSL(i, nvec) {
        float g[4], r[4];
        ld4(dy + i * 4, g);
        ld4(ref + i * 4, r);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = act == DH_ACT_RELU ? (r[j] > 0.f ? g[j] : 0.f) : g[j] * gelu_erf_grad(r[j]);
        st4(dx + i * 4, g);
    }
}
template <typename T>
__global__ void act_bwd16_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ dx, long npiece,
                                 int act) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    GSL(i, npiece) {
        float g[V], r[V];
        ldv(dy + i * V, g);
        ldv(ref + i * V, r);
#pragma unroll
        for (int j = 0; j < V; ++j) g[j] = act == DH_ACT_RELU ? (r[j] > 0.f ? g[j] : 0.f) : g[j] * gelu_erf_grad(r[j]);
        stv(dx + i * V, g);
    }
}

// ---- MaxPool 3x3 stride 2 pad 1 -------------------------------------------------------------
// V packed arg-max codes (one byte per channel) as one 4- or 8-byte access
template <int V> __device__ __forceinline__ unsigned long long ld_bytes(const unsigned char* p) {
    if constexpr (V == 8) return *reinterpret_cast<const unsigned long long*>(p);
    else return *reinterpret_cast<const unsigned*>(p);
}
template <int V> __device__ __forceinline__ void st_bytes(unsigned char* p, unsigned long long v) {
    if constexpr (V == 8) *reinterpret_cast<unsigned long long*>(p) = v;
    else *reinterpret_cast<unsigned*>(p) = (unsigned)v;
}

// bn_scale / bn_shift ([groups][C], optional): the input is a pre-normalisation BatchNorm tensor and
// relu(x * scale + shift) is applied on load -- the stem's BN+ReLU output is then never written (its only consumer
// is this pool; rounded to T exactly as the separate bn_apply pass would have stored it)
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, unsigned char* __restrict__ arg, int N,
                                   int H, int W, int C, int OH, int OW, const float* __restrict__ bn_scale,
                                   const float* __restrict__ bn_shift, int groups) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    const int vn = C / V;
    GSL(i, (long)N * OH * OW * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const long n = t / OH;
        float sc[V], sh[V];
        if (bn_scale) {
            const int g = (int)(n / (N / groups));
#pragma unroll
            for (int j = 0; j < V; ++j) { sc[j] = bn_scale[g * C + c + j]; sh[j] = bn_shift[g * C + c + j]; }
        }
        float m[V];
        int k[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { m[j] = -INFINITY; k[j] = -1; }
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float v[V];
                ldv(x + ((n * H + iy) * W + ix) * C + c, v);
                if (bn_scale) {
                    T r[V];
#pragma unroll
                    for (int j = 0; j < V; ++j) { stf(&r[j], fmaxf(v[j] * sc[j] + sh[j], 0.f)); v[j] = ldf(&r[j]); }   // T rounding
                }
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (v[j] > m[j] || k[j] < 0) { m[j] = v[j]; k[j] = ky * 3 + kx; }     // first maximum wins
            }
        }
        stv(y + i * V, m);
        if (arg) {
            unsigned long long bits = 0;
#pragma unroll
            for (int j = 0; j < V; ++j) bits |= (unsigned long long)(k[j] & 0xff) << (8 * j);
            st_bytes<V>(arg + i * V, bits);
        }
    }
}
// gather form with the saved window arg-max (row-major scan, strict >, as ATen's CPU kernel): an input pixel
// receives dy of each of the <= 4 windows covering it whose arg-max it is.  Deterministic, no atomics.
// One thread owns the 2x2 input pixels (2*oy + a, 2*ox + b) of a 16-byte channel piece: the four windows
// (oy..oy+1, ox..ox+1) cover them all, so every window's (arg, dy) piece is loaded once per 4 outputs instead of
// once per output; contributions are added in the same order as the per-pixel gather (window row, then column).
template <typename T>
__global__ void maxpool_bwd_kernel(const unsigned char* __restrict__ arg, const T* __restrict__ dy, T* __restrict__ dx,
                                   int N, int H, int W, int C, int OH, int OW) {
    constexpr int V = V16<T>::N;
    const int vn = C / V, BH = (H + 1) / 2, BW = (W + 1) / 2;      // 2x2 input blocks
    GSL(i, (long)N * BH * BW * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int bx = (int)(t % BW); t /= BW;
        const int by = (int)(t % BH);
        const long n = t / BH;
        unsigned long long bits[2][2];
        float d[2][2][V];
#pragma unroll
        for (int wy = 0; wy < 2; ++wy)
#pragma unroll
            for (int wx = 0; wx < 2; ++wx) {
                const int oy = by + wy, ox = bx + wx;
                bits[wy][wx] = ~0ull;                        // no window: matches no tap index
#pragma unroll
                for (int j = 0; j < V; ++j) d[wy][wx][j] = 0.f;
                if (oy < OH && ox < OW) {
                    const long o = ((n * OH + oy) * OW + ox) * C + c;
                    bits[wy][wx] = ld_bytes<V>(arg + o);
                    ldv(dy + o, d[wy][wx]);
                }
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int iy = 2 * by + a, ix = 2 * bx + b;
                if (iy >= H || ix >= W) continue;
                float g[V];
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] = 0.f;
                // windows oy in {iy/2, (iy+1)/2} = {by, by + a}, ox likewise: tap index of this pixel inside each
#pragma unroll
                for (int wy = 0; wy <= a; ++wy)
#pragma unroll
                    for (int wx = 0; wx <= b; ++wx) {
                        const unsigned kk = (a + 1 - 2 * wy) * 3 + (b + 1 - 2 * wx);
#pragma unroll
                        for (int j = 0; j < V; ++j)
                            if (((bits[wy][wx] >> (8 * j)) & 0xff) == kk) g[j] += d[wy][wx][j];
                    }
                stv(dx + (((n * H + iy) * W + ix) * C + c), g);
            }
    }
}

// ---- nearest x2 -----------------------------------------------------------------------------
template <typename T>
__global__ void up2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    const int vn = C / V;
    GSL(i, (long)N * 4 * H * W * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ox = (int)(t % (2 * W)); t /= 2 * W;
        const int oy = (int)(t % (2 * H));
        const long n = t / (2 * H);
        float v[V];
        ldv(x + ((n * H + oy / 2) * W + ox / 2) * C + c, v);
        stv(y + i * V, v);
    }
}
template <typename T>
__global__ void up2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    const int vn = C / V;
    GSL(i, (long)N * H * W * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        float s[V];
#pragma unroll
        for (int j_ = 0; j_ < V; ++j_) s[j_] = 0.f;
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                float v[V];
                ldv(dy + ((n * 2 * H + 2 * iy + dyy) * 2 * W + 2 * ix + dxx) * C + c, v);
#pragma unroll
                for (int j = 0; j < V; ++j) s[j] += v[j];
            }
        stv(dx + i * V, s);
    }
}

// ---- |a - b| then bilinear x4 (align_corners=False) ------------------------------------------
__device__ __forceinline__ void bil_src(int d, int in, int& i0, int& i1, float& l) {
    float s = ((float)d + 0.5f) * 0.25f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = s - (float)i0;
}
// One thread per SOURCE pixel piece: |a - b| of its 3x3 (border-clamped) neighbourhood is formed once and the 4x4
// destination pixels it governs are written from it -- 18 loads per 16 output pieces instead of 8 per piece, and the
// index arithmetic once per 16.  Same terms in the same order as the per-destination form (bit-identical): destination
// row 4*iy+dy interpolates source rows (iy-1, iy) for dy < 2 and (iy, iy+1) for dy >= 2; where bil_src clamps at a
// border its weight on the substituted row is exactly 0.
template <typename T>
__global__ void absdiff_up4_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, int N,
                                       int H, int W, int C) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    const int vn = C / V, OW = 4 * W;
    GSL(i, (long)N * H * W * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        const int ys[3] = {iy > 0 ? iy - 1 : 0, iy, iy < H - 1 ? iy + 1 : H - 1};
        const int xs[3] = {ix > 0 ? ix - 1 : 0, ix, ix < W - 1 ? ix + 1 : W - 1};
        float d[3][3][V];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float u[V], v[V];
                const long off = ((n * H + ys[r]) * W + xs[q]) * C + c;
                ldv(a + off, u);
                ldv(b + off, v);
#pragma unroll
                for (int j = 0; j < V; ++j) d[r][q][j] = fabsf(u[j] - v[j]);
            }
        float wy[4][2], wx[4][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int i0_, i1_; float l;
            bil_src(4 * iy + k, H, i0_, i1_, l);
            wy[k][0] = 1.f - l; wy[k][1] = l;
            bil_src(4 * ix + k, W, i0_, i1_, l);
            wx[k][0] = 1.f - l; wx[k][1] = l;
        }
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            T* yrow = y + (((n * 4 * H + 4 * iy + dy) * OW + 4 * ix) * C + c);
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                float acc[V];
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const int r = (dy < 2 ? 0 : 1) + pp, q = (dx < 2 ? 0 : 1) + qq;
#pragma unroll
                        for (int j = 0; j < V; ++j) acc[j] += wy[dy][pp] * wx[dx][qq] * d[r][q][j];
                    }
                stv(yrow + dx * C, acc);
            }
        }
    }
}
// gather backward: source pixel (iy, ix) collects from destination rows 4*iy-2 .. 4*iy+5
template <typename T>
__global__ void absdiff_up4_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ dy,
                                       T* __restrict__ da, T* __restrict__ db, int N, int H, int W, int C) {
    constexpr int V = V16<T>::N;      // one 16-byte piece per lane
    const int vn = C / V, OH = 4 * H, OW = 4 * W;
    GSL(i, (long)N * H * W * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        float g[V];
#pragma unroll
        for (int j_ = 0; j_ < V; ++j_) g[j_] = 0.f;
        for (int oy = 4 * iy - 2; oy <= 4 * iy + 5; ++oy) {
            if (oy < 0 || oy >= OH) continue;
            int y0, y1; float ly;
            bil_src(oy, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = 4 * ix - 2; ox <= 4 * ix + 5; ++ox) {
                if (ox < 0 || ox >= OW) continue;
                int x0, x1; float lx;
                bil_src(ox, W, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                float d[V];
                ldv(dy + ((n * OH + oy) * OW + ox) * C + c, d);
#pragma unroll
                for (int j = 0; j < V; ++j) g[j] += wy * wx * d[j];
            }
        }
        float u[V], v[V], ga[V], gb[V];
        ldv(a + i * V, u);
        ldv(b + i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float df = u[j] - v[j];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            ga[j] = sg * g[j];
            gb[j] = -ga[j];
        }
        stv(da + i * V, ga);
        stv(db + i * V, gb);
    }
}

// Second half of the fused backward of |a - b| -> bilinear x4 -> conv3x3 (dh_conv3x3_dgrad_up4 left per-tile coarse partials
// [tile][4][6][32] fp32, tile = (n, ty, tx) of 8x16 fine pixels covering coarse rows 2 ty - 1 .. 2 ty + 2, columns
// 4 tx - 1 .. 4 tx + 4): one thread per coarse pixel piece sums the <= 4 tiles that touch it, in a fixed order, and applies
// the sign of a - b.
__global__ void absdiff_up4_combine_kernel(const float* __restrict__ partial, const bf16* __restrict__ a, const bf16* __restrict__ b,
                                           bf16* __restrict__ da, bf16* __restrict__ db, int N, int H, int W) {
    const int tilesY = H / 2, tilesX = W / 4;            // fine grid 4H x 4W in 8 x 16 tiles
    GSL(i, (long)N * H * W * 4) {
        const int c = (int)(i & 3) * 8;
        long t = i >> 2;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long n = t / H;
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = 0.f;
        for (int ty = (iy - 1) >> 1; ty <= (iy + 1) >> 1; ++ty) {          // tiles with 2 ty - 1 <= iy <= 2 ty + 2
            if (ty < 0 || ty >= tilesY) continue;
            const int lr = iy - (2 * ty - 1);
            if (lr < 0 || lr > 3) continue;
            for (int tx = (ix - 1) >> 2; tx <= (ix + 1) >> 2; ++tx) {      // tiles with 4 tx - 1 <= ix <= 4 tx + 4
                if (tx < 0 || tx >= tilesX) continue;
                const int lc = ix - (4 * tx - 1);
                if (lc < 0 || lc > 5) continue;
                const float* src = partial + ((((n * tilesY + ty) * tilesX + tx) * 4 + lr) * 6 + lc) * 32 + c;
                const float4 u = *reinterpret_cast<const float4*>(src), v = *reinterpret_cast<const float4*>(src + 4);
                g[0] += u.x; g[1] += u.y; g[2] += u.z; g[3] += u.w;
                g[4] += v.x; g[5] += v.y; g[6] += v.z; g[7] += v.w;
            }
        }
        float u[8], v[8], ga[8], gb[8];
        ldv(a + i * 8, u);
        ldv(b + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float df = u[j] - v[j];
            const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            ga[j] = sg * g[j];
            gb[j] = -ga[j];
        }
        stv(da + i * 8, ga);
        stv(db + i * 8, gb);
    }
}

// |a-b| on small fp32/T row tensors (token differences, networks.py:1311) and its derivative
template <typename T>
__global__ void absdiff_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long n) {
    GSL(i, n) stf(y + i, fabsf(ldf(a + i) - ldf(b + i)));
}
template <typename T>
__global__ void absdiff_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ dy,
                                   T* __restrict__ da, T* __restrict__ db, long n, int accumulate) {
    GSL(i, n) {
        const float df = ldf(a + i) - ldf(b + i);
        const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
        const float g = sg * ldf(dy + i);
        if (accumulate) { stf(da + i, ldf(da + i) + g); stf(db + i, ldf(db + i) - g); }
        else { stf(da + i, g); stf(db + i, -g); }
    }
}

// ---- zero insertion for stride-2 data gradients: z[n, 2y, 2x, c] = dy[n, y, x, c] ------------
template <typename T>
__global__ void zero_insert2_kernel(const T* __restrict__ dy, T* __restrict__ z, int N, int OH, int OW, int H, int W,
                                    int C) {
    const int vn = C / 4;
    GSL(i, (long)N * H * W * vn) {
        const int c = (int)(i % vn) * 4;
        long t = i / vn;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const long n = t / H;
        float v[4] = {0, 0, 0, 0};
        if (!(x & 1) && !(y & 1) && y / 2 < OH && x / 2 < OW) ld4(dy + ((n * OH + y / 2) * OW + x / 2) * C + c, v);
        st4(z + i * 4, v);
    }
}

// ---- the data gradient of a 1x1 STRIDE-2 convolution, added where it lives: x[n, 2y, 2x, c] += coarse[n, y, x, c] ---------------
// (zero insertion + a 1x1 convolution on the fine grid wrote and read a 4x larger tensor of 3/4 zeros and ran the GEMM on it:
// the Bottleneck shortcut of a ResNet-50 layer2 (models/resnet.py:106-118) took 304 + 443 us at the 1024 x 1024 bench size)
template <typename T>
__global__ void add_coarse_kernel(T* __restrict__ x, const T* __restrict__ coarse, int N, int OH, int OW, int H, int W, int C) {
    constexpr int V = V16<T>::N;
    const int vn = C / V;
    GSL(i, (long)N * OH * OW * vn) {
        const int c = (int)(i % vn) * V;
        long t = i / vn;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const long n = t / OH;
        if (2 * oy >= H || 2 * ox >= W) continue;
        float a[V], b[V];
        T* dst = x + ((n * H + 2 * oy) * W + 2 * ox) * C + c;
        ldv(dst, a);
        ldv(coarse + i * V, b);
#pragma unroll
        for (int j = 0; j < V; ++j) a[j] += b[j];
        stv(dst, a);
    }
}

// ---- weight packing: OIHW fp32 -> [tap][OPad][I] T  and  [tap'][IPad][O] T (flipped) ---------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ oscale, int O, int I, int KS,
                                   int OPad, T* __restrict__ fwd, int IPad, int OK, T* __restrict__ dgrad) {
    const int taps = KS * KS;
    if (fwd) {
        GSL(i, (long)taps * OPad * I) {
            const int ci = (int)(i % I);
            const int o = (int)((i / I) % OPad);
            const int tap = (int)(i / ((long)I * OPad));
            stf(fwd + i, o < O ? w[((long)o * I + ci) * taps + tap] * (oscale ? oscale[o] : 1.f) : 0.f);
        }
    }
    if (dgrad) {
        GSL(i, (long)taps * IPad * OK) {     // reduction dim (output channels) zero-padded to OK
            const int o = (int)(i % OK);
            const int ci = (int)((i / OK) % IPad);
            const int tap = (int)(i / ((long)OK * IPad));
            stf(dgrad + i, (ci < I && o < O) ? w[((long)o * I + ci) * taps + (taps - 1 - tap)] : 0.f);
        }
    }
}

// All weights of a net in ONE launch: the per-step re-pack was ~120 four-microsecond launches for newUNetTrans.
// jobs[] (device memory, built once by the host: sources live in the flat arena, destinations are persistent)
// are sorted by first_block; a workgroup finds its job by bisection.
struct PackJob {
    const float* w;
    void* fwd;          // [taps][OPad][I] or null
    void* dgrad;        // [taps][IPad][OK] (flipped taps) or null
    int O, I, KS, OPad, IPad, OK;
    int dtype;          // DH_DTYPE_* | 0x200: FRAGMENT-ORDER destinations (see below)
    int first_block, nblocks;
};
// Fragment order (bf16, 3x3): [rows / 16][K / 32][taps][64 lanes][8] -- the 1 KiB a wavefront of csrc/conv_wreg.hip loads as ONE
// MFMA A fragment (lane (pl = lane & 15, g = lane >> 4): row 16 r + pl, reduction channels 32 c + 8 g .. + 7) is contiguous, so
// the once-per-workgroup load of the register-resident weights reads whole cache lines (measured: 10.2 -> 5.0 us for the
// 295 KB of a 64 x 256-channel block).  rows = output channels (forward) / input channels (data gradient, flipped taps).
// eight consecutive destination elements from eight sources `stride` floats apart (ok[e] false: zero): the index arithmetic of a
// destination element (three divisions by run-time values) is paid once per 16-byte store instead of once per element -- the
// kernel was bound by exactly that (37 us for 12 M elements, ~100 integer instructions each)
template <typename T>
__device__ __forceinline__ void pack_store8(T* dst, const float* __restrict__ src, unsigned stride, unsigned okmask) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ((okmask >> e) & 1) ? src[(size_t)e * stride] : 0.f;
    if constexpr (sizeof(T) == 2) {
        uint4 u;
        u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]); u.z = f2bf2(v[4], v[5]); u.w = f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(dst) = u;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}
template <typename T> __device__ __forceinline__ void pack_put8(T* dst, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        uint4 u;
        u.x = f2bf2(v[0], v[1]); u.y = f2bf2(v[2], v[3]); u.z = f2bf2(v[4], v[5]); u.w = f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(dst) = u;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}
// 3x3 layers with O, I multiples of 32 and no padding (all but the class head and the stem: 95 % of the weights): a workgroup takes
// a 32 (o) x 32 (ci) x 9 tile -- 32 runs of 288 CONSECUTIVE floats of the OIHW master, 16-byte loads -- through LDS and writes
// every destination in 16-byte pieces.  The element-wise form reads the master at a 36-byte stride: one cache-line transaction
// per 4-byte element, 40 us per step for 3 M weights whatever the index arithmetic costs (8 elements per lane and store: 39.7).
constexpr int PK_T = 32, PK_ROW = PK_T * 9, PK_PITCH = PK_ROW + 1;
__host__ __device__ inline bool pack_tiled(const PackJob& j) {
    // (the master is a view at any float offset of the parameter arena: the 16-byte loads need it aligned)
    return j.KS == 3 && j.O % PK_T == 0 && j.I % PK_T == 0 && j.OPad == j.O && j.IPad == j.I && j.OK == j.O &&
           (reinterpret_cast<unsigned long long>(j.w) & 15) == 0;
}
template <typename T>
__device__ __forceinline__ void pack_job_tiled(const PackJob& j, int lb, float* t /*[32][PK_PITCH]*/) {
    const int tid = threadIdx.x, nti = j.I / PK_T, ntiles = (j.O / PK_T) * nti;
    T* fwd = reinterpret_cast<T*>(j.fwd);
    T* dgrad = reinterpret_cast<T*>(j.dgrad);
    const bool frag = (j.dtype & 0x200) != 0;
    for (int tile = lb; tile < ntiles; tile += j.nblocks) {
        const int o0 = (tile / nti) * PK_T, c0 = (tile % nti) * PK_T;
        __syncthreads();                                  // the previous tile's readers are done
        for (int q = tid; q < PK_T * (PK_ROW / 4); q += 256) {
            const int r = q / (PK_ROW / 4), k = (q % (PK_ROW / 4)) * 4;
            const float4 v = *reinterpret_cast<const float4*>(j.w + ((size_t)(o0 + r) * j.I + c0) * 9 + k);
            float* d = t + r * PK_PITCH + k;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        // 1152 pieces of 8 elements per destination
        for (int q = tid; q < 9 * PK_T * 4; q += 256) {
            float v[8];
            if (frag) {
                const int lane = q & 63, tap = (q >> 6) % 9, blk = q / (64 * 9);             // blk: 16-row block of the tile
                const int rr = blk * 16 + (lane & 15), kk = (lane >> 4) * 8;
                if (fwd) {                                // rows = o, reduction = ci
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = t[rr * PK_PITCH + (kk + e) * 9 + tap];
                    pack_put8(fwd + ((((size_t)(o0 / 16 + blk) * (j.I / 32) + c0 / 32) * 9 + tap) * 64 + lane) * 8, v);
                }
                if (dgrad) {                              // rows = ci, reduction = o, taps flipped
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = t[(kk + e) * PK_PITCH + rr * 9 + (8 - tap)];
                    pack_put8(dgrad + ((((size_t)(c0 / 16 + blk) * (j.OK / 32) + o0 / 32) * 9 + tap) * 64 + lane) * 8, v);
                }
            } else {
                const int pc = q & 3, r = (q >> 2) & 31, tap = q >> 7;                          // piece, row, tap
                if (fwd) {                                // [tap][o][ci]
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = t[r * PK_PITCH + (pc * 8 + e) * 9 + tap];
                    pack_put8(fwd + ((size_t)tap * j.OPad + o0 + r) * j.I + c0 + pc * 8, v);
                }
                if (dgrad) {                              // [tap][ci][o], taps flipped
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = t[(pc * 8 + e) * PK_PITCH + r * 9 + (8 - tap)];
                    pack_put8(dgrad + ((size_t)tap * j.IPad + c0 + r) * j.OK + o0 + pc * 8, v);
                }
            }
        }
    }
}
template <typename T>
__device__ __forceinline__ void pack_job(const PackJob& j, int lb) {
    const int taps = j.KS * j.KS;
    // 32-bit index arithmetic: a layer's packed forms hold < 2^31 elements (ops.PackPlan.add refuses anything
    // larger; 3x3 x 2048 x 2048 is 3.8e7), and 64-bit divisions by run-time values cost ~100 instructions each
    const unsigned step = (unsigned)j.nblocks * blockDim.x;
    T* fwd = reinterpret_cast<T*>(j.fwd);
    T* dgrad = reinterpret_cast<T*>(j.dgrad);
    const unsigned t0 = (unsigned)lb * blockDim.x + threadIdx.x;
    if (j.dtype & 0x200) {
        // (fragment order: the 8 elements of a lane are 8 consecutive reduction channels -- I % 32 == OK % 32 == 0)
        const int nch_f = j.I / 32, nch_d = j.OK / 32;
        if (fwd)
            for (unsigned i8 = t0; i8 < (unsigned)(taps * j.OPad * j.I) / 8; i8 += step) {
                const unsigned i = i8 * 8;
                const int lane = (int)((i >> 3) & 63), tap = (int)((i >> 9) % (unsigned)taps);
                const int c = (int)((i / (512u * (unsigned)taps)) % (unsigned)nch_f), r16 = (int)(i / (512u * (unsigned)taps * (unsigned)nch_f));
                const int o = r16 * 16 + (lane & 15), ci = c * 32 + (lane >> 4) * 8;
                pack_store8(fwd + i, j.w + (unsigned)((o * j.I + ci) * taps + tap), (unsigned)taps, o < j.O ? 0xffu : 0u);
            }
        if (dgrad)
            for (unsigned i8 = t0; i8 < (unsigned)(taps * j.IPad * j.OK) / 8; i8 += step) {
                const unsigned i = i8 * 8;
                const int lane = (int)((i >> 3) & 63), tap = (int)((i >> 9) % (unsigned)taps);
                const int c = (int)((i / (512u * (unsigned)taps)) % (unsigned)nch_d), r16 = (int)(i / (512u * (unsigned)taps * (unsigned)nch_d));
                const int ci = r16 * 16 + (lane & 15), o = c * 32 + (lane >> 4) * 8;
                unsigned ok = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) ok |= (unsigned)(ci < j.I && o + e < j.O) << e;
                pack_store8(dgrad + i, j.w + (unsigned)((o * j.I + ci) * taps + (taps - 1 - tap)), (unsigned)(j.I * taps), ok);
            }
        return;
    }
    if (fwd) {
        if (j.I % 8 == 0) {
            for (unsigned i8 = t0; i8 < (unsigned)(taps * j.OPad * j.I) / 8; i8 += step) {
                const unsigned i = i8 * 8;
                const int ci = (int)(i % (unsigned)j.I);
                const int o = (int)((i / (unsigned)j.I) % (unsigned)j.OPad);
                const int tap = (int)(i / (unsigned)(j.I * j.OPad));
                pack_store8(fwd + i, j.w + (unsigned)((o * j.I + ci) * taps + tap), (unsigned)taps, o < j.O ? 0xffu : 0u);
            }
        } else {
            for (unsigned i = t0; i < (unsigned)(taps * j.OPad * j.I); i += step) {
                const int ci = (int)(i % (unsigned)j.I);
                const int o = (int)((i / (unsigned)j.I) % (unsigned)j.OPad);
                const int tap = (int)(i / (unsigned)(j.I * j.OPad));
                stf(fwd + i, o < j.O ? j.w[(unsigned)((o * j.I + ci) * taps + tap)] : 0.f);
            }
        }
    }
    if (dgrad) {
        if (j.OK % 8 == 0) {
            for (unsigned i8 = t0; i8 < (unsigned)(taps * j.IPad * j.OK) / 8; i8 += step) {
                const unsigned i = i8 * 8;
                const int o = (int)(i % (unsigned)j.OK);
                const int ci = (int)((i / (unsigned)j.OK) % (unsigned)j.IPad);
                const int tap = (int)(i / (unsigned)(j.OK * j.IPad));
                unsigned ok = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) ok |= (unsigned)(ci < j.I && o + e < j.O) << e;
                pack_store8(dgrad + i, j.w + (unsigned)((o * j.I + ci) * taps + (taps - 1 - tap)), (unsigned)(j.I * taps), ok);
            }
        } else {
            for (unsigned i = t0; i < (unsigned)(taps * j.IPad * j.OK); i += step) {
                const int o = (int)(i % (unsigned)j.OK);
                const int ci = (int)((i / (unsigned)j.OK) % (unsigned)j.IPad);
                const int tap = (int)(i / (unsigned)(j.OK * j.IPad));
                stf(dgrad + i, (ci < j.I && o < j.O) ? j.w[(unsigned)((o * j.I + ci) * taps + (taps - 1 - tap))] : 0.f);
            }
        }
    }
}
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackJob* __restrict__ jobs, int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                               // last job whose first_block <= blockIdx.x
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PackJob j = jobs[lo];
    const int lb = blockIdx.x - j.first_block;
    __shared__ __attribute__((aligned(16))) float tile[PK_T * PK_PITCH];
    if (pack_tiled(j)) {
        if ((j.dtype & 0xff) == DH_DTYPE_BF16) pack_job_tiled<bf16>(j, lb, tile); else pack_job_tiled<float>(j, lb, tile);
        return;
    }
    if ((j.dtype & 0xff) == DH_DTYPE_BF16) pack_job<bf16>(j, lb); else pack_job<float>(j, lb);
}

// ---- column sums: out[c] (+)= sum_p x[p, c]  (bias gradients) --------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long P, int C,
                                                             float* __restrict__ partial) {
    // block b sums rows b, b+grid, ...; thread t covers channel t % C, row phase t / C
    __shared__ float red[256 * 8];
    constexpr int V = sizeof(T) == 2 ? 8 : 4;              // channels per lane: one 16-byte piece
    if (C % V == 0 && C <= 1024 && 256 % (C / V) == 0) {
        // one 16-byte piece per lane and row, 256/(C/V) row phases per workgroup, FOUR independent row streams per lane (the
        // loop is otherwise a chain of dependent-latency loads; with 8-byte loads and two streams: 1.9 TB/s on 33.5 MB)
        const int cvn = C / V, cv = threadIdx.x % cvn, ph = threadIdx.x / cvn, nph = 256 / cvn;
        float s[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j) s[u][j] = 0.f;
        const long step = (long)gridDim.x * nph;
        long p = (long)blockIdx.x * nph + ph;
        auto row = [&](long r, float (&acc)[V]) {
            float v[V];
            if constexpr (V == 8) unpack16(*reinterpret_cast<const uint4*>(x + r * C + cv * V), v);
            else ld4(x + r * C + cv * V, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        };
        for (; p + 3 * step < P; p += 4 * step) {
            row(p, s[0]); row(p + step, s[1]); row(p + 2 * step, s[2]); row(p + 3 * step, s[3]);
        }
        for (int u = 0; p < P; p += step, ++u) row(p, s[u & 3]);
        float* mine = red + threadIdx.x * V;               // red: [256 lanes][V] floats
#pragma unroll
        for (int j = 0; j < V; ++j) mine[j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float t = 0.f;
            for (int r = 0; r < nph; ++r) t += red[(r * cvn + c / V) * V + (c % V)];
            partial[(long)blockIdx.x * C + c] = t;
        }
        return;
    }
    const int c = threadIdx.x % C, ph = threadIdx.x / C, nph = 256 / C;
    float s = 0.f;
    if (ph < nph)
        for (long p = (long)blockIdx.x * nph + ph; p < P; p += (long)gridDim.x * nph) s += ldf(x + p * C + c);
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int r = 0; r < nph; ++r) t += red[r * C + threadIdx.x];
        partial[(long)blockIdx.x * C + threadIdx.x] = t;
    }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
    GSL(i, n) stf(dst + i, src[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, long n, int accumulate) {
    GSL(i, n) { if (accumulate) dst[i] += ldf(src + i); else dst[i] = ldf(src + i); }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
extern "C" int dh_reduce_partials(const float* partial, long nt, long n, float scale, float* out, int accumulate,
                                  void* stream);

extern "C" int dh_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, long HW, int CP, void* stream) {
    if (CP < C) CP = C;
    const long n = (long)N * CP * HW;
    if (CP * (dtype == DH_DTYPE_BF16 ? 2 : 4) == 16) {
        const long np = (long)N * HW;
        if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(nchw_to_nhwc_piece_kernel<bf16>, dim3(ew_grid(np, 256)), dim3(256), 0, ST(stream), src, (bf16*)dst, N, C, HW);
        else hipLaunchKernelGGL(nchw_to_nhwc_piece_kernel<float>, dim3(ew_grid(np, 256)), dim3(256), 0, ST(stream), src, (float*)dst, N, C, HW);
        DH_CHECK_LAUNCH("nchw_to_nhwc");
        return 0;
    }
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), src, (bf16*)dst, N, C, HW, CP);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), src, (float*)dst, N, C, HW, CP);
    DH_CHECK_LAUNCH("nchw_to_nhwc");
    return 0;
}
// dy: [N][H][W][CP] T with CP = 8 (bf16) or 4 / 8 (fp32) channels per pixel, the first NC real; w_oihw: [NC][32][3][3]
// fp32 master; dx: [N][H][W][32] T
extern "C" int dh_head_dgrad3x3(int dtype, const void* dy, int CP, const float* w_oihw, int NC, void* dx, int N, int H,
                                int W, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 8 && NC <= CP, "head_dgrad3x3: n_class=%d with CP=%d", NC, CP);
    const long n = (long)N * H * W;
    const int grid = ew_grid(n, 256);
    if (NC <= 2 && dtype == DH_DTYPE_BF16 && CP == 8 && !getenv("DAHITRA_HEAD_DGRAD_VALU")) {      // matrix cores, K = 18
        const long groups = (n + 15) / 16;
        long g = (groups * 64 + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(head_dgrad3x3_mfma_kernel<false>, dim3((int)g), dim3(256), 0, ST(stream), (const bf16*)dy, w_oihw, (bf16*)dx,
                           N, H, W, NC, (const bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, 1, (float*)nullptr);
        DH_CHECK_LAUNCH("head_dgrad3x3");
        return 0;
    }
    if (NC <= 2 && (dtype == DH_DTYPE_BF16 ? CP == 8 : (CP == 4 || CP == 8))) {      // weights in registers
#define DH_HEAD_REG(T, CPV, NCV, LPP)                                                                                       \
        hipLaunchKernelGGL((head_dgrad3x3_reg_kernel<T, CPV, NCV>), dim3(ew_grid(n * LPP, 256 * 8)), dim3(256), 0, ST(stream), \
                           (const T*)dy, w_oihw, (T*)dx, N, H, W)
        if (dtype == DH_DTYPE_BF16) { if (NC == 2) DH_HEAD_REG(bf16, 8, 2, 4); else DH_HEAD_REG(bf16, 8, 1, 4); }
        else if (CP == 4) { if (NC == 2) DH_HEAD_REG(float, 4, 2, 8); else DH_HEAD_REG(float, 4, 1, 8); }
        else { if (NC == 2) DH_HEAD_REG(float, 8, 2, 8); else DH_HEAD_REG(float, 8, 1, 8); }
#undef DH_HEAD_REG
        DH_CHECK_LAUNCH("head_dgrad3x3");
        return 0;
    }
    if (dtype == DH_DTYPE_BF16) {
        DH_REQUIRE(CP == 8, "head_dgrad3x3: bf16 needs 8 channels per pixel (one 16-byte piece), got %d", CP);
        hipLaunchKernelGGL((head_dgrad3x3_kernel<bf16, 8>), dim3(grid), dim3(256), 0, ST(stream), (const bf16*)dy, w_oihw, (bf16*)dx, N, H, W, NC);
    } else if (CP == 4) {
        hipLaunchKernelGGL((head_dgrad3x3_kernel<float, 4>), dim3(grid), dim3(256), 0, ST(stream), (const float*)dy, w_oihw, (float*)dx, N, H, W, NC);
    } else {
        DH_REQUIRE(CP == 8, "head_dgrad3x3: fp32 needs 4 or 8 channels per pixel, got %d", CP);
        hipLaunchKernelGGL((head_dgrad3x3_kernel<float, 8>), dim3(grid), dim3(256), 0, ST(stream), (const float*)dy, w_oihw, (float*)dx, N, H, W, NC);
    }
    DH_CHECK_LAUNCH("head_dgrad3x3");
    return 0;
}
// dh_head_dgrad3x3 for a head that sits behind a ReLU (bf16, n_class <= 2, dy one 16-byte piece per pixel): relu_out
// [N][H][W][32] is that ReLU's output, dx = (relu_out > 0) * gradient -- the activation's backward folded into this kernel.
extern "C" int dh_head_dgrad3x3_relu(const void* dy, const float* w_oihw, int NC, const void* relu_out, void* dx, int N, int H,
                                     int W, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 2 && dy && relu_out && dx, "head_dgrad3x3_relu: bad arguments (n_class=%d)", NC);
    const long groups = ((long)N * H * W + 15) / 16;
    long g = (groups * 64 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL((head_dgrad3x3_mfma_kernel<false, true>), dim3((int)g), dim3(256), 0, ST(stream), (const bf16*)dy, w_oihw,
                       (bf16*)dx, N, H, W, NC, (const bf16*)relu_out, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, 1, (float*)nullptr);
    DH_CHECK_LAUNCH("head_dgrad3x3_relu");
    return 0;
}
// The gated form (see head_dgrad3x3_mfma_kernel<true>): bf16, n_class <= 2, dy one 16-byte piece per pixel.  g: masked
// gradient [N][H][W][32]; partial: [2][32][dh_head_dgrad3x3_bn_blocks] for dh_bn_bwd_from_partials (ntiles = blocks).
extern "C" int dh_head_dgrad3x3_bn_blocks(int N, int H, int W, int groups) {
    if (groups < 1 || N % groups) return 0;
    const long n16 = ((long)N * H * W / groups + 15) / 16;
    long bpg = (n16 * 64 + 255) / 256;
    const long cap = 2048 / groups;
    if (bpg > cap) bpg = cap;
    return (int)(bpg * groups);
}
extern "C" int dh_head_dgrad3x3_bn(const void* dy, const float* w_oihw, int NC, const void* y, const float* mask_scale,
                                   const float* mask_shift, const float* mean, const float* invstd, int groups, void* g,
                                   float* partial, int N, int H, int W, void* stream) {
    DH_REQUIRE(NC >= 1 && NC <= 2 && y && mask_scale && mask_shift && mean && invstd && g && partial, "head_dgrad3x3_bn: bad arguments");
    const int grid = dh_head_dgrad3x3_bn_blocks(N, H, W, groups);
    DH_REQUIRE(grid > 0, "head_dgrad3x3_bn: %d images do not split into %d groups", N, groups);
    hipLaunchKernelGGL(head_dgrad3x3_mfma_kernel<true>, dim3(grid), dim3(256), 0, ST(stream), (const bf16*)dy, w_oihw, (bf16*)g, N, H, W,
                       NC, (const bf16*)y, mask_scale, mask_shift, mean, invstd, groups, partial);
    DH_CHECK_LAUNCH("head_dgrad3x3_bn");
    return 0;
}
extern "C" int dh_augment_pairs_u8(const unsigned char* a, const unsigned char* b, const unsigned char* l, const int* idx,
                                   const int* params, int N, int H, int W, int h, int w, float* out_a, float* out_b,
                                   unsigned char* out_l, void* stream) {
    DH_REQUIRE(N > 0 && h > 0 && w > 0 && h <= H && w <= W, "augment_pairs_u8: bad sizes N=%d %dx%d -> %dx%d", N, H, W, h, w);
    const long n = (long)N * h * w;
    hipLaunchKernelGGL(augment_pairs_u8_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), a, b, l, idx, params, N, H, W,
                       h, w, out_a, out_b, out_l);
    DH_CHECK_LAUNCH("augment_pairs_u8");
    return 0;
}
extern "C" int dh_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, long HW, void* stream) {
    const long n = (long)N * C * HW;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)src, dst, N, C, HW);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)src, dst, N, C, HW);
    DH_CHECK_LAUNCH("nhwc_to_nchw");
    return 0;
}
extern "C" int dh_copy_channels(int dtype, const void* src, int Cs, int sc0, void* dst, int Cd, int dc0, int Cn,
                                long P, void* stream) {
    DH_REQUIRE(Cn % 4 == 0 && sc0 % 4 == 0 && dc0 % 4 == 0 && Cs % 4 == 0 && Cd % 4 == 0, "copy_channels: channel counts must be multiples of 4");
    const long n = P * (Cn / 4);
    if (dtype == DH_DTYPE_BF16 && !((Cs | sc0 | Cd | dc0 | Cn) & 7))
        hipLaunchKernelGGL(copy_channels16_kernel<bf16>, dim3(ew_grid(P * (Cn / 8), 256)), dim3(256), 0, ST(stream), (const bf16*)src, Cs, sc0, (bf16*)dst, Cd, dc0, Cn, P);
    else if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(copy_channels_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)src, Cs, sc0, (bf16*)dst, Cd, dc0, Cn, P);
    else hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)src, Cs, sc0, (float*)dst, Cd, dc0, Cn, P);
    DH_CHECK_LAUNCH("copy_channels");
    return 0;
}
// t: [2 P][C], cat: [P][2 C] (models/networks.py:1309, 1344: cat([x1, x2], 1) of the two temporal streams, which are the two
// batch halves here).  inverse = 0: cat <- t; 1: t <- cat (the gradient's way back)
extern "C" int dh_cat_halves(int dtype, void* t, void* cat, int C, long P, int inverse, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "cat_halves: C=%d must be a multiple of %d", C, V);
    const long n = P * (2 * C / V);
    if (dtype == DH_DTYPE_BF16) {
        if (inverse) hipLaunchKernelGGL((cat_halves_kernel<bf16, true>), dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (bf16*)t, (bf16*)cat, C, P);
        else hipLaunchKernelGGL((cat_halves_kernel<bf16, false>), dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (bf16*)t, (bf16*)cat, C, P);
    } else {
        if (inverse) hipLaunchKernelGGL((cat_halves_kernel<float, true>), dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (float*)t, (float*)cat, C, P);
        else hipLaunchKernelGGL((cat_halves_kernel<float, false>), dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (float*)t, (float*)cat, C, P);
    }
    DH_CHECK_LAUNCH("cat_halves");
    return 0;
}
extern "C" int dh_add(int dtype, const void* a, const void* b, void* y, long n, void* stream) {
    DH_REQUIRE(n % 4 == 0, "add: n must be a multiple of 4");
    const long nv = n / 4;
    if (dtype == DH_DTYPE_BF16 && n % 8 == 0) hipLaunchKernelGGL(add16_kernel<bf16>, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, n / 8);
    else if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(add_kernel<bf16>, dim3(ew_grid(nv, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, nv);
    else hipLaunchKernelGGL(add_kernel<float>, dim3(ew_grid(nv, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)y, nv);
    DH_CHECK_LAUNCH("add");
    return 0;
}
extern "C" int dh_add_pos(int dtype, const void* x, const float* pos, void* y, int N, long HW, int C, void* stream) {
    const long n = (long)N * HW * C;
    if (dtype == DH_DTYPE_BF16 && C % 8 == 0) hipLaunchKernelGGL(add_pos16_kernel<bf16>, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST(stream), (const bf16*)x, pos, (bf16*)y, N, HW, C);
    else if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(add_pos_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)x, pos, (bf16*)y, N, HW, C);
    else hipLaunchKernelGGL(add_pos_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)x, pos, (float*)y, N, HW, C);
    DH_CHECK_LAUNCH("add_pos");
    return 0;
}
extern "C" int dh_add_pos_bwd(int dtype, const void* dy, float* dpos, int N, long HW, int C, int accumulate, void* stream) {
    const long n = HW * C;
    if (dtype == DH_DTYPE_BF16 && C == 32) hipLaunchKernelGGL(add_pos_bwd32_kernel, dim3((unsigned)((HW + 15) / 16)), dim3(256), 0, ST(stream), (const bf16*)dy, dpos, N, HW, accumulate);
    else if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(add_pos_bwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)dy, dpos, N, HW, C, accumulate);
    else hipLaunchKernelGGL(add_pos_bwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)dy, dpos, N, HW, C, accumulate);
    DH_CHECK_LAUNCH("add_pos_bwd");
    return 0;
}
extern "C" int dh_act_bwd(int dtype, const void* dy, const void* ref, void* dx, long n, int act, void* stream) {
    DH_REQUIRE(n % 4 == 0 && (act == DH_ACT_RELU || act == DH_ACT_GELU), "act_bwd: bad arguments");
    const long nv = n / 4;
    if (dtype == DH_DTYPE_BF16 && n % 8 == 0) hipLaunchKernelGGL(act_bwd16_kernel<bf16>, dim3(ew_grid(n / 8, 256)), dim3(256), 0, ST(stream), (const bf16*)dy, (const bf16*)ref, (bf16*)dx, n / 8, act);
    else if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16>, dim3(ew_grid(nv, 256)), dim3(256), 0, ST(stream), (const bf16*)dy, (const bf16*)ref, (bf16*)dx, nv, act);
    else hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(ew_grid(nv, 256)), dim3(256), 0, ST(stream), (const float*)dy, (const float*)ref, (float*)dx, nv, act);
    DH_CHECK_LAUNCH("act_bwd");
    return 0;
}
extern "C" int dh_maxpool3x3s2_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int N, int H, int W, int C,
                                   const float* bn_scale, const float* bn_shift, int groups, void* stream) {
    DH_REQUIRE(!bn_scale || (bn_shift && groups > 0 && N % groups == 0), "maxpool: bn_scale needs bn_shift and groups | N");
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "maxpool: C=%d must be a multiple of %d", C, V);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long n = (long)N * OH * OW * (C / V);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, argmax, N, H, W, C, OH, OW, bn_scale, bn_shift, groups);
    else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, argmax, N, H, W, C, OH, OW, bn_scale, bn_shift, groups);
    DH_CHECK_LAUNCH("maxpool_fwd");
    return 0;
}
extern "C" int dh_maxpool3x3s2_bwd(int dtype, const unsigned char* argmax, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "maxpool: C=%d must be a multiple of %d", C, V);
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long n = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / V);      // one thread per 2x2 input block piece
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), argmax, (const bf16*)dy, (bf16*)dx, N, H, W, C, OH, OW);
    else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), argmax, (const float*)dy, (float*)dx, N, H, W, C, OH, OW);
    DH_CHECK_LAUNCH("maxpool_bwd");
    return 0;
}
extern "C" int dh_upsample2_nearest_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "upsample2: C=%d must be a multiple of %d", C, V);
    const long n = (long)N * 4 * H * W * (C / V);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(up2_fwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, N, H, W, C);
    else hipLaunchKernelGGL(up2_fwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, N, H, W, C);
    DH_CHECK_LAUNCH("up2_fwd");
    return 0;
}
extern "C" int dh_upsample2_nearest_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "upsample2: C=%d must be a multiple of %d", C, V);
    const long n = (long)N * H * W * (C / V);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(up2_bwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)dy, (bf16*)dx, N, H, W, C);
    else hipLaunchKernelGGL(up2_bwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)dy, (float*)dx, N, H, W, C);
    DH_CHECK_LAUNCH("up2_bwd");
    return 0;
}
extern "C" int dh_absdiff_upsample4_fwd(int dtype, const void* a, const void* b, void* y, int N, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "absdiff_upsample4: C=%d must be a multiple of %d", C, V);
    const long n = (long)N * H * W * (C / V);          // one thread per source pixel piece (4x4 destination pixels)
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_up4_fwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, N, H, W, C);
    else hipLaunchKernelGGL(absdiff_up4_fwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)y, N, H, W, C);
    DH_CHECK_LAUNCH("absdiff_up4_fwd");
    return 0;
}
extern "C" int dh_absdiff_upsample4_bwd(int dtype, const void* a, const void* b, const void* dy, void* da, void* db, int N, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0, "absdiff_upsample4: C=%d must be a multiple of %d", C, V);
    const long n = (long)N * H * W * (C / V);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_up4_bwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (const bf16*)dy, (bf16*)da, (bf16*)db, N, H, W, C);
    else hipLaunchKernelGGL(absdiff_up4_bwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (const float*)dy, (float*)da, (float*)db, N, H, W, C);
    DH_CHECK_LAUNCH("absdiff_up4_bwd");
    return 0;
}
// a, b: [N][H][W][32] bf16 (the COARSE maps); partial from dh_conv3x3_dgrad_up4 on the 4H x 4W grid; da, db like a
extern "C" int dh_absdiff_up4_combine(const float* partial, const void* a, const void* b, void* da, void* db, int N, int H, int W,
                                      void* stream) {
    DH_REQUIRE(H % 2 == 0 && W % 4 == 0, "absdiff_up4_combine: coarse map %dx%d must tile into 8x16 fine tiles", H, W);
    const long n = (long)N * H * W * 4;
    hipLaunchKernelGGL(absdiff_up4_combine_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), partial, (const bf16*)a,
                       (const bf16*)b, (bf16*)da, (bf16*)db, N, H, W);
    DH_CHECK_LAUNCH("absdiff_up4_combine");
    return 0;
}
extern "C" int dh_absdiff(int dtype, const void* a, const void* b, void* y, long n, void* stream) {
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (bf16*)y, n);
    else hipLaunchKernelGGL(absdiff_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (float*)y, n);
    DH_CHECK_LAUNCH("absdiff");
    return 0;
}
extern "C" int dh_absdiff_bwd(int dtype, const void* a, const void* b, const void* dy, void* da, void* db, long n, int accumulate, void* stream) {
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_bwd_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)a, (const bf16*)b, (const bf16*)dy, (bf16*)da, (bf16*)db, n, accumulate);
    else hipLaunchKernelGGL(absdiff_bwd_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)a, (const float*)b, (const float*)dy, (float*)da, (float*)db, n, accumulate);
    DH_CHECK_LAUNCH("absdiff_bwd");
    return 0;
}
extern "C" int dh_zero_insert2(int dtype, const void* dy, void* z, int N, int OH, int OW, int H, int W, int C, void* stream) {
    DH_REQUIRE(C % 4 == 0, "zero_insert2: C %% 4");
    const long n = (long)N * H * W * (C / 4);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(zero_insert2_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)dy, (bf16*)z, N, OH, OW, H, W, C);
    else hipLaunchKernelGGL(zero_insert2_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)dy, (float*)z, N, OH, OW, H, W, C);
    DH_CHECK_LAUNCH("zero_insert2");
    return 0;
}
extern "C" int dh_add_coarse(int dtype, void* x, const void* coarse, int N, int OH, int OW, int H, int W, int C, void* stream) {
    const int V = dtype == DH_DTYPE_BF16 ? 8 : 4;
    DH_REQUIRE(C % V == 0 && x && coarse && OH == (H + 1) / 2 && OW == (W + 1) / 2, "add_coarse: C=%d, %dx%d <- %dx%d", C, H, W, OH, OW);
    const long n = (long)N * OH * OW * (C / V);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(add_coarse_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (bf16*)x, (const bf16*)coarse, N, OH, OW, H, W, C);
    else hipLaunchKernelGGL(add_coarse_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (float*)x, (const float*)coarse, N, OH, OW, H, W, C);
    DH_CHECK_LAUNCH("add_coarse");
    return 0;
}
extern "C" int dh_pack_weight(int dtype, const float* w_oihw, const float* out_scale, int O, int I, int ks, int OPad, void* fwd, int IPad, int dgrad_inner, void* dgrad, void* stream) {
    const int OK = dgrad_inner > O ? dgrad_inner : O;
    const long n = (long)ks * ks * (OPad > IPad ? OPad : IPad) * (OK > I ? OK : I);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), w_oihw, out_scale, O, I, ks, OPad, (bf16*)fwd, IPad, OK, (bf16*)dgrad);
    else hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), w_oihw, out_scale, O, I, ks, OPad, (float*)fwd, IPad, OK, (float*)dgrad);
    DH_CHECK_LAUNCH("pack_weight");
    return 0;
}
extern "C" int dh_pack_job_size(void) { return (int)sizeof(PackJob); }
// jobs_dev: njobs PackJob records in device memory {w, fwd, dgrad: 3 pointers; O, I, KS, OPad, IPad, OK, dtype,
// first_block, nblocks: 9 ints; padded to dh_pack_job_size() bytes}, sorted by first_block; total_blocks = sum nblocks
extern "C" int dh_pack_weights_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (njobs <= 0) return 0;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, ST(stream),
                       reinterpret_cast<const PackJob*>(jobs_dev), njobs);
    DH_CHECK_LAUNCH("pack_weights_multi");
    return 0;
}
// workspace: 1024 * C floats
extern "C" int dh_colsum(int dtype, const void* x, long P, int C, float* out, int accumulate, void* workspace, void* stream) {
    DH_REQUIRE(C >= 1 && C <= 256, "colsum: C=%d out of range", C);
    const long rows_per_pass = 256 / (C >= 4 && (C & 3) == 0 ? C / 4 : C);       // rows one workgroup covers per step
    long want = (P + rows_per_pass * 8 - 1) / (rows_per_pass * 8);               // >= 8 steps per workgroup
    // at most one workgroup per CU: the second stage (reduce_partials, one workgroup for <= 32 channels) walks the partial
    // rows as a latency chain -- 1024 rows cost it 9.5 us, 256 rows ~4 us, and the first stage streams at HBM rate either way
    const int grid = (int)(want < 64 ? 64 : (want > 256 ? 256 : want));
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (const bf16*)x, P, C, partial);
    else hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)x, P, C, partial);
    DH_CHECK_LAUNCH("colsum");
    return dh_reduce_partials(partial, grid, C, 1.0f, out, accumulate, stream);
}
extern "C" int dh_cast_from_f32(int dtype, const float* src, void* dst, long n, void* stream) {
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), src, (bf16*)dst, n);
    else hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), src, (float*)dst, n);
    DH_CHECK_LAUNCH("cast_from_f32");
    return 0;
}
extern "C" int dh_cast_to_f32(int dtype, const void* src, float* dst, long n, int accumulate, void* stream) {
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const bf16*)src, dst, n, accumulate);
    else hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(ew_grid(n, 256)), dim3(256), 0, ST(stream), (const float*)src, dst, n, accumulate);
    DH_CHECK_LAUNCH("cast_to_f32");
    return 0;
}

// ---- |second half - first half| of [B][2][n] token sets (models/networks.py:1311) and its gradient
namespace {
template <typename T>
__device__ __forceinline__ void absdiff_halves_body(const T* __restrict__ tok, T* __restrict__ out, int B, long n, int bid, int nb) {
    GSLB(i, (long)B * n, bid, nb) {
        const long b = i / n, j = i % n;
        stf(out + i, fabsf(ldf(tok + (b * 2 + 1) * n + j) - ldf(tok + (b * 2) * n + j)));
    }
}
template <typename T>
__global__ void absdiff_halves_kernel(const T* __restrict__ tok, T* __restrict__ out, int B, long n) {
    absdiff_halves_body<T>(tok, out, B, n, blockIdx.x, gridDim.x);
}
template <typename T>
__device__ __forceinline__ void absdiff_halves_bwd_body(const T* __restrict__ tok, const T* __restrict__ dout, T* __restrict__ dtok,
                                                        int B, long n, int bid, int nb) {
    GSLB(i, (long)B * n, bid, nb) {      // accumulates into dtok
        const long b = i / n, j = i % n;
        const float df = ldf(tok + (b * 2 + 1) * n + j) - ldf(tok + (b * 2) * n + j);
        const float g = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * ldf(dout + i);
        T* p1 = dtok + (b * 2 + 1) * n + j;
        T* p0 = dtok + (b * 2) * n + j;
        stf(p1, ldf(p1) + g);
        stf(p0, ldf(p0) - g);
    }
}
template <typename T>
__global__ void absdiff_halves_bwd_kernel(const T* __restrict__ tok, const T* __restrict__ dout, T* __restrict__ dtok,
                                          int B, long n) {
    absdiff_halves_bwd_body<T>(tok, dout, dtok, B, n, blockIdx.x, gridDim.x);
}

// ---- a JOB TABLE of the small element-wise kernels above: the three levels of _forward_trans_module (models/networks.py:1297-1318)
// each issue a positional add, a channel concatenation, |token2 - token1| (and their gradients) between the launches they share;
// recorded per round (ops.EwBatch) the independent ones go out as ONE launch: a job is a run of workgroups of it
enum { DH_EW_ADD_POS = 1, DH_EW_CAT_HALVES = 2, DH_EW_SPLIT_HALVES = 3, DH_EW_ABSDIFF_HALVES = 4, DH_EW_ABSDIFF_HALVES_BWD = 5,
       DH_EW_ADD_POS_BWD = 6 };       // (= include/dahitra_hip.h)
struct EwJob {
    int op, blk0, nblk, i0, i1;
    long l0;
    const void* a;
    const void* b;
    void* c;
};
constexpr int EW_MAXJ = 12;
struct EwMulti {
    EwJob j[EW_MAXJ];
    int n;
};
__global__ __launch_bounds__(256) void ew_multi_kernel(EwMulti m) {
    int k = 0;
    while (k + 1 < m.n && (int)blockIdx.x >= m.j[k + 1].blk0) ++k;
    const EwJob& J = m.j[k];
    const int bid = (int)blockIdx.x - J.blk0, nb = J.nblk;
    switch (J.op) {
        case DH_EW_ADD_POS: add_pos16_body<bf16>((const bf16*)J.a, (const float*)J.b, (bf16*)J.c, J.i0, J.l0, J.i1, bid, nb); break;
        case DH_EW_CAT_HALVES: cat_halves_body<bf16, false>((bf16*)const_cast<void*>(J.a), (bf16*)J.c, J.i0, J.l0, bid, nb); break;
        case DH_EW_SPLIT_HALVES: cat_halves_body<bf16, true>((bf16*)J.c, (bf16*)const_cast<void*>(J.a), J.i0, J.l0, bid, nb); break;
        case DH_EW_ABSDIFF_HALVES: absdiff_halves_body<float>((const float*)J.a, (float*)J.c, J.i0, J.l0, bid, nb); break;
        case DH_EW_ABSDIFF_HALVES_BWD: absdiff_halves_bwd_body<float>((const float*)J.a, (const float*)J.b, (float*)J.c, J.i0, J.l0, bid, nb); break;
        case DH_EW_ADD_POS_BWD: add_pos_bwd32_body((const bf16*)J.a, (float*)J.c, J.i0, J.l0, J.i1, bid); break;
        default: break;
    }
}
}  // namespace
// n jobs, one launch.  op[k] and its operands:
//   DH_EW_ADD_POS            c = a + pos (dh_add_pos, bf16, C % 8 == 0):        a = x, b = pos, c = y, i0 = N, i1 = C, l0 = H W
//   DH_EW_CAT_HALVES         dh_cat_halves(inverse = 0), bf16, C % 8 == 0:       a = t [2 P][C], c = cat [P][2 C], i0 = C, l0 = P
//   DH_EW_SPLIT_HALVES       dh_cat_halves(inverse = 1):                         a = cat, c = t, i0 = C, l0 = P
//   DH_EW_ABSDIFF_HALVES     dh_absdiff_halves, fp32:                            a = tok, c = out, i0 = B, l0 = n
//   DH_EW_ABSDIFF_HALVES_BWD dh_absdiff_halves_bwd, fp32:                        a = tok, b = dout, c = dtok (accumulated), i0 = B, l0 = n
//   DH_EW_ADD_POS_BWD        dh_add_pos_bwd, bf16, C = 32:                       a = dy, c = dpos, i0 = N, i1 = accumulate, l0 = H W
// The jobs of one call must not depend on each other (they run side by side).
extern "C" int dh_ew_multi(int n, const int* op, const void* const* a, const void* const* b, void* const* c, const int* i0,
                           const int* i1, const long* l0, void* stream) {
    DH_REQUIRE(n >= 1 && n <= EW_MAXJ && op && a && b && c && i0 && i1 && l0, "ew_multi: 1 .. %d jobs, got %d", EW_MAXJ, n);
    EwMulti m;
    int nblk = 0;
    for (int k = 0; k < n; ++k) {
        EwJob& J = m.j[k];
        J.op = op[k]; J.a = a[k]; J.b = b[k]; J.c = c[k]; J.i0 = i0[k]; J.i1 = i1[k]; J.l0 = l0[k];
        long work;
        switch (op[k]) {
            case DH_EW_ADD_POS:
                DH_REQUIRE(i1[k] % 8 == 0 && a[k] && b[k] && c[k], "ew_multi: add_pos job %d", k);
                work = ew_grid((long)i0[k] * l0[k] * i1[k] / 8, 256); break;
            case DH_EW_CAT_HALVES: case DH_EW_SPLIT_HALVES:
                DH_REQUIRE(i0[k] % 8 == 0 && a[k] && c[k], "ew_multi: cat_halves job %d", k);
                work = ew_grid(l0[k] * (2 * i0[k] / 8), 256); break;
            case DH_EW_ABSDIFF_HALVES:
                DH_REQUIRE(a[k] && c[k], "ew_multi: absdiff_halves job %d", k);
                work = ew_grid((long)i0[k] * l0[k], 256); break;
            case DH_EW_ABSDIFF_HALVES_BWD:
                DH_REQUIRE(a[k] && b[k] && c[k], "ew_multi: absdiff_halves_bwd job %d", k);
                work = ew_grid((long)i0[k] * l0[k], 256); break;
            case DH_EW_ADD_POS_BWD:
                DH_REQUIRE(a[k] && c[k], "ew_multi: add_pos_bwd job %d", k);
                work = (l0[k] + 15) / 16; break;
            default: DH_FAIL("ew_multi: unknown op %d (job %d)", op[k], k);
        }
        J.blk0 = nblk; J.nblk = (int)work;
        nblk += (int)work;
    }
    m.n = n;
    hipLaunchKernelGGL(ew_multi_kernel, dim3(nblk), dim3(256), 0, ST(stream), m);
    DH_CHECK_LAUNCH("ew_multi");
    return 0;
}
extern "C" int dh_absdiff_halves(int dtype, const void* tok, void* out, int B, long n, void* stream) {
    const long t = (long)B * n;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_halves_kernel<bf16>, dim3(ew_grid(t, 256)), dim3(256), 0, ST(stream), (const bf16*)tok, (bf16*)out, B, n);
    else hipLaunchKernelGGL(absdiff_halves_kernel<float>, dim3(ew_grid(t, 256)), dim3(256), 0, ST(stream), (const float*)tok, (float*)out, B, n);
    DH_CHECK_LAUNCH("absdiff_halves");
    return 0;
}
extern "C" int dh_absdiff_halves_bwd(int dtype, const void* tok, const void* dout, void* dtok_accum, int B, long n, void* stream) {
    const long t = (long)B * n;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(absdiff_halves_bwd_kernel<bf16>, dim3(ew_grid(t, 256)), dim3(256), 0, ST(stream), (const bf16*)tok, (const bf16*)dout, (bf16*)dtok_accum, B, n);
    else hipLaunchKernelGGL(absdiff_halves_bwd_kernel<float>, dim3(ew_grid(t, 256)), dim3(256), 0, ST(stream), (const float*)tok, (const float*)dout, (float*)dtok_accum, B, n);
    DH_CHECK_LAUNCH("absdiff_halves_bwd");
    return 0;
}
