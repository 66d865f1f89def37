// Token-side kernels of the bitemporal transformer (gfx950).  All of this is a few hundred KFLOP
// per image: latency-, not roofline-bound.  Internal math is fp32; tensors crossing to the pixel
// side are in the activation type T.  The TOKENS themselves (a few KB) are always fp32, whatever T is:
// the hierarchical nets decode against |token2 - token1| (networks.py:1311), and two bf16-rounded tokens
// that are close cancel catastrophically (measured: 0.4 % token error -> 40 % decoder-output error).
//
//  * semantic tokenizer  (reference models/networks.py:312-319): 1x1 conv 32->L, softmax over the
//    HW pixels, attention-weighted pooling -> L tokens of width 32 per image; + learned pos
//    (networks.py:332-334) fused into the store of the concatenated [A;B] token set.
//  * cross-attention operand preparation (models/help_funcs.py:66-114).  For a pixel row x and the
//    L tokens m of its image, with k_l = Wk LN(m_l), v_l = Wv LN(m_l):
//        dots[h,l] = scale * <Wq_h LN(x), k_{l,h}>  = LN(x) . Kq[h,l,:],  Kq[h,l,:] = scale * Wq_h^T k_{l,h}
//        out       = Wo concat_h(sum_l a[h,l] v_{l,h}) = sum_{h,l} a[h,l] Vo[h,l,:], Vo[h,l,:] = Wo_h v_{l,h}
//    i.e. two dense (H*L)x32 products per pixel instead of materialising the 4096x512 q / out
//    tensors (fp re-association of the same sums; ~17x fewer FLOPs).  Kq / Vo are produced here
//    per image, the pixel-side products run on the MFMA conv kernel with per-image weights.
//  * grouped softmax over the L keys of each head (help_funcs.py:103), forward and backward.
//  * token self-attention core of the encoder (models/networks.py:457-488) for <= 16 tokens.
#include "common.h"

namespace {

constexpr int D = 32;   // transformer width

__device__ __forceinline__ float quad_sum(float v) {       // over the 4 lanes of a quad (DPP quad permutes)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    return v;
}

// ------------------------------------------------------------------------------------------
// tokenizer
// ------------------------------------------------------------------------------------------
// One tokenizer call (forward: stages 0-3, backward: stages 4-7), by value in the launch: the stages are body functions of
// (job, virtual block index), launched through tok_multi_kernel for one call or for the recorded calls of several independent
// levels (dh_xprep_batch_*: DAHiTra's three levels tokenize 64 images each -- 12 + 12 launches of a few dozen workgroups).
struct TokJob {
    const void* x;                       // [S*HW][32]
    const float *wa, *pos;               // [L][32]; [2L][32] or null
    float *logits, *stats, *pooled;      // [S*HW][L], [S][L][2], [S][L][32]: written forward, read backward
    float *tok_cat, *part;               // forward: [B][2L][32]; partial pooled sums [S][nch][L*32]
    const float* dtok_cat;               // backward
    void* dx;
    float *dlogits, *partial, *dwa, *dpos;
    long P;
    int S, B, HW, nch, chunk, nblk, accumulate;
};
constexpr int TB_MAXJ = 4;
struct TokMulti {
    int n;
    int first[TB_MAXJ + 1];
    TokJob j[TB_MAXJ];
};
constexpr int TOK_DWA_CHUNK = 1024;      // pixel rows per tok_dwa workgroup (16 per thread)

template <typename T, int L>
__device__ __forceinline__ void tok_logits_body(const T* __restrict__ x, const float* __restrict__ wa, float* __restrict__ logits,
                                                long P, int bx) {
    __shared__ float w[L * D];
    for (int i = threadIdx.x; i < L * D; i += 256) w[i] = wa[i];
    __syncthreads();
    const long p = (long)bx * 256 + threadIdx.x;
    if (p >= P) return;
    float acc[L];
#pragma unroll
    for (int l = 0; l < L; ++l) acc[l] = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        float v[4];
        ld4(x + p * D + c, v);
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[l] += v[j] * w[l * D + c + j];
    }
#pragma unroll
    for (int l = 0; l < L; ++l) logits[p * L + l] = acc[l];
}

// softmax statistics over the HW pixels of one (image, token): one 256-thread workgroup each
template <int L>
__device__ __forceinline__ void tok_stats_body(const float* __restrict__ logits, int HW, float* __restrict__ stats /*[S][L][2]*/,
                                               int bx) {
    __shared__ float red[4];
    const int s = bx / L, l = bx % L, tid = threadIdx.x;
    const float* lg = logits + (size_t)s * HW * L + l;
    float m = -INFINITY;
#pragma unroll 8
    for (int n = tid; n < HW; n += 256) m = fmaxf(m, lg[(size_t)n * L]);        // (loads in flight; same order)
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float e = 0.f;
#pragma unroll 8
    for (int n = tid; n < HW; n += 256) e += __expf(lg[(size_t)n * L] - m);
    e = wave_sum(e);
    if ((tid & 63) == 0) red[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) {
        stats[(size_t)bx * 2 + 0] = m;
        stats[(size_t)bx * 2 + 1] = 1.f / (red[0] + red[1] + red[2] + red[3]);
    }
}

// sum over the pixel rows [n0, n1) of weight(n, l) * x[n][c], for all l < L and c < 32, by one 256-thread workgroup:
// thread (phase = tid / 4, q = tid % 4) walks rows n0 + phase, + 64, ... with ONE 16-byte piece of x (8 channels for bf16,
// two pieces in fp32) per row and keeps L x 8 sums; the 64 phases are combined through `red` [32][L*32] in a fixed order
// (upper 32 phases add onto the lower 32, then 32 rows are summed).  The first form gave every (l, c) its own lane:
// 2-byte loads and each x element fetched L times (17.4 / 12.8 us for the 16.8 MB of x in tok_pool_partial / tok_dwa).
template <typename T, int L, typename W>
__device__ __forceinline__ void weighted_colsum(const T* __restrict__ xs, long n0, long n1, W weight, float* red,
                                                float* __restrict__ out) {
    constexpr int LD = L * D;
    const int tid = threadIdx.x, q = tid & 3, ph = tid >> 2;
    float acc[L][8];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[l][c] = 0.f;
#pragma unroll 4
    for (long n = n0 + ph; n < n1; n += 64) {      // (four rows' loads in flight; the sums keep their order)
        float xv[8], wl[L];
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
            float v[4];
            ld4(xs + n * D + q * 8 + c, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[c + j] = v[j];
        }
        weight(n, wl);
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[l][c] += wl[l] * xv[c];
    }
    float* mine = red + (ph & 31) * LD + q * 8;
    if (ph < 32) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int c = 0; c < 8; ++c) mine[l * D + c] = acc[l][c];
    }
    __syncthreads();
    if (ph >= 32) {
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int c = 0; c < 8; ++c) mine[l * D + c] += acc[l][c];
    }
    __syncthreads();
    if (tid < LD) {
        float t = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) t += red[k * LD + tid];
        out[tid] = t;
    }
}

// partial pooled sums over one chunk of pixels; grid (chunks, S), 256 threads
template <typename T, int L>
__device__ __forceinline__ void tok_pool_partial_body(const T* __restrict__ x, const float* __restrict__ logits,
                                                      const float* __restrict__ stats, int HW, int chunk, int nch,
                                                      float* __restrict__ partial /*[S][chunks][L*32]*/, int bx, int by, float* red) {
    const int s = by;
    const float* lg = logits + (size_t)s * HW * L;
    float mx[L];
#pragma unroll
    for (int l = 0; l < L; ++l) mx[l] = stats[((size_t)s * L + l) * 2];
    const int n0 = bx * chunk, n1 = min(n0 + chunk, HW);
    weighted_colsum<T, L>(x + (size_t)s * HW * D, n0, n1,
                          [&](long n, float (&wl)[L]) {
#pragma unroll
                              for (int l = 0; l < L; ++l) wl[l] = __expf(lg[(size_t)n * L + l] - mx[l]);
                          },
                          red, partial + ((size_t)s * nch + bx) * (L * D));
}

// combine chunks, normalise, add the learned positional embedding and place into [B][2L][32]
template <typename T, int L>
__device__ __forceinline__ void tok_finish_body(const float* __restrict__ partial, const float* __restrict__ stats,
                                                const float* __restrict__ pos /*[2L][32] or null*/, int chunks, int B,
                                                float* __restrict__ pooled, float* __restrict__ tok_cat, int bx) {
    const int s = bx, tid = threadIdx.x;
    if (tid >= L * D) return;                                // (launched with 256 threads)
    const int l = tid / D, c = tid % D;
    float acc = 0.f;
    for (int k = 0; k < chunks; ++k) acc += partial[((size_t)s * chunks + k) * (L * D) + tid];
    acc *= stats[((size_t)s * L + l) * 2 + 1];
    pooled[((size_t)s * L + l) * D + c] = acc;
    const int b = s % B, stream = s / B;
    const int j = stream * L + l;
    tok_cat[((size_t)b * 2 * L + j) * D + c] = acc + (pos ? pos[j * D + c] : 0.f);   // tokens stay fp32 (see header)
}

// per pixel: dlogit and the tokenizer's contribution to dx (accumulated into dx in place)
template <typename T, int L>
__device__ __forceinline__ void tok_bwd_body(const T* __restrict__ x, const float* __restrict__ logits,
                                             const float* __restrict__ stats, const float* __restrict__ pooled,
                                             const float* __restrict__ dtok_cat, const float* __restrict__ wa, int HW, int B,
                                             T* __restrict__ dx, float* __restrict__ dlogits, int bx, int by) {
    // FOUR lanes per pixel row, 8 channels (one 16-byte piece of x / dx for bf16) each: a wave's loads are 64 consecutive
    // pieces.  (One lane per pixel walked its 64-byte row with eight 8-byte loads 64 bytes apart: 34 us for 55 MB.)  The
    // channel dot products meet over the quad (DPP); every lane keeps its 8 channels of dtok / Wa in registers.
    __shared__ float sdt[L * D], sw[L * D], sdot[L], smx[L], siv[L];
    const int s = by;
    const int b = s % B, stream = s / B;
    for (int i = threadIdx.x; i < L * D; i += 256) {
        sdt[i] = dtok_cat[((size_t)b * 2 * L + stream * L) * D + i];
        sw[i] = wa[i];
    }
    __syncthreads();
    if (threadIdx.x < L) {
        float t = 0.f;
        for (int c = 0; c < D; ++c) t += sdt[threadIdx.x * D + c] * pooled[((size_t)s * L + threadIdx.x) * D + c];
        sdot[threadIdx.x] = t;
        smx[threadIdx.x] = stats[((size_t)s * L + threadIdx.x) * 2];
        siv[threadIdx.x] = stats[((size_t)s * L + threadIdx.x) * 2 + 1];
    }
    __syncthreads();
    const int q = threadIdx.x & 3, n = bx * 64 + (threadIdx.x >> 2);
    const bool live = n < HW;                            // (dead lanes keep running: the quad sums are DPP moves)
    const size_t row = (size_t)s * HW + (live ? n : 0);
    float dt[L][8], w[L][8];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int c = 0; c < 8; ++c) { dt[l][c] = sdt[l * D + q * 8 + c]; w[l][c] = sw[l * D + q * 8 + c]; }
    float xv[8], g[8];
#pragma unroll
    for (int c = 0; c < 8; c += 4) {
        float v[4];
        ld4(x + row * D + q * 8 + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) xv[c + j] = v[j];
        ld4(dx + row * D + q * 8 + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) g[c + j] = v[j];
    }
    float pr[L], dl[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        float dp = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) dp += dt[l][c] * xv[c];
        dp = quad_sum(dp);
        pr[l] = __expf(logits[row * L + l] - smx[l]) * siv[l];
        dl[l] = pr[l] * (dp - sdot[l]);
    }
    if (live) {
#pragma unroll
        for (int l = 0; l < L; ++l)
            if ((l & 3) == q) dlogits[row * L + l] = dl[l];
#pragma unroll
        for (int c = 0; c < 8; c += 4) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.f;
#pragma unroll
                for (int l = 0; l < L; ++l) t += pr[l] * dt[l][c + j] + dl[l] * w[l][c + j];
                o[j] = g[c + j] + t;
            }
            st4(dx + row * D + q * 8 + c, o);
        }
    }
}

// partial dWa[l][c] = sum over a chunk of pixel rows of dlogits[p][l] * x[p][c] (weighted_colsum)
template <typename T, int L>
__device__ __forceinline__ void tok_dwa_body(const T* __restrict__ x, const float* __restrict__ dlogits, long P, long chunk,
                                             float* __restrict__ partial, int bx, float* red) {
    const long p0 = (long)bx * chunk, p1 = (p0 + chunk < P) ? p0 + chunk : P;
    weighted_colsum<T, L>(x, p0, p1,
                          [&](long p, float (&wl)[L]) {
#pragma unroll
                              for (int l = 0; l < L; ++l) wl[l] = dlogits[p * L + l];
                          },
                          red, partial + (size_t)bx * (L * D));
}

// dpos[j][c] (+)= sum_b dtok_cat[b][j][c]
__device__ __forceinline__ void tok_dpos_body(const float* __restrict__ dtok_cat, int B, int n, float* __restrict__ dpos,
                                              int accumulate, int bx) {
    const int i = bx * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
#pragma unroll 16
    for (int b = 0; b < B; ++b) s += dtok_cat[(size_t)b * n + i];       // (one workgroup: the loads in flight, the adds in order)
    if (accumulate) dpos[i] += s; else dpos[i] = s;
}

// workgroups of stage STAGE that job `a` needs
template <int L> __host__ __device__ inline int tok_stage_blocks(const TokJob& a, int stage) {
    switch (stage) {
        case 0: return (int)((a.P + 255) / 256);            // logits
        case 1: return a.S * L;                             // softmax statistics
        case 2: return a.nch * a.S;                         // partial pooled sums
        case 3: return a.S;                                 // finish
        case 4: return ((a.HW + 63) / 64) * a.S;            // dlogits, dx
        case 5: return a.nblk;                              // dWa partials
        case 6: return a.dpos ? (2 * L * 32 + 255) / 256 : 0;
        default: return (L * 32 + 31) / 32;                 // dWa reduction
    }
}
template <typename T, int L, int STAGE>
__global__ __launch_bounds__(256) void tok_multi_kernel(TokMulti m) {
    __shared__ __attribute__((aligned(16))) float red[(STAGE == 2 || STAGE == 5) ? 32 * L * D : (STAGE == 7 ? 8 * 32 * 2 : 1)];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const TokJob& a = m.j[j];
    const int local = (int)blockIdx.x - m.first[j];
    if constexpr (STAGE == 0) tok_logits_body<T, L>((const T*)a.x, a.wa, a.logits, a.P, local);
    else if constexpr (STAGE == 1) tok_stats_body<L>(a.logits, a.HW, a.stats, local);
    else if constexpr (STAGE == 2) tok_pool_partial_body<T, L>((const T*)a.x, a.logits, a.stats, a.HW, a.chunk, a.nch, a.part, local % a.nch, local / a.nch, red);
    else if constexpr (STAGE == 3) tok_finish_body<T, L>(a.part, a.stats, a.pos, a.nch, a.B, a.pooled, a.tok_cat, local);
    else if constexpr (STAGE == 4) {
        const int gx = (a.HW + 63) / 64;
        tok_bwd_body<T, L>((const T*)a.x, a.logits, a.stats, a.pooled, a.dtok_cat, a.wa, a.HW, a.B, (T*)a.dx, a.dlogits, local % gx, local / gx);
    } else if constexpr (STAGE == 5) tok_dwa_body<T, L>((const T*)a.x, a.dlogits, a.P, (long)TOK_DWA_CHUNK, a.partial, local, red);
    else if constexpr (STAGE == 6) tok_dpos_body(a.dtok_cat, a.B, 2 * L * 32, a.dpos, a.accumulate, local);
    else dh_reduce_partials_body(a.partial, a.nblk, (long)L * 32, 1.0f, a.dwa, a.accumulate, local, reinterpret_cast<double(*)[32]>(red));
}

// ------------------------------------------------------------------------------------------
// cross-attention operand preparation (per image)
// ------------------------------------------------------------------------------------------
// Weight access is coalesced everywhere: thread index runs along the contiguous dimension of whichever form
// (fp32 master [out][in], or the packed transpose in T produced by dh_pack_weight) makes that possible.
// ---- the small per-image token-side kernels of SEVERAL independent decoder stacks in one launch (dh_xprep_batch_*): every
// kernel below is a body function of (arguments, virtual block index) with two entry points -- its own launch, and a
// multi-launch whose workgroups [first[j], first[j + 1]) run job j's gx[j] x gy[j] x . grid (arguments by value).  DAHiTra's three
// levels each issue these 11 - 13 us, few-dozen-workgroup launches twice per direction; alone they are latency, together one.
constexpr int XB_MAXJ = 4;
template <typename A> struct XMulti {
    int n;
    int first[XB_MAXJ + 1], gx[XB_MAXJ], gy[XB_MAXJ];
    A a[XB_MAXJ];
};
template <typename A> __device__ __forceinline__ int xb_locate(const XMulti<A>& m, uint3& bid) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    int local = (int)blockIdx.x - m.first[j];
    bid.x = local % m.gx[j]; local /= m.gx[j];
    bid.y = local % m.gy[j];
    bid.z = local / m.gy[j];
    return j;
}
#define XB_BID make_uint3(blockIdx.x, blockIdx.y, blockIdx.z)

struct PrepArgs {
    const void* tok;        // token rows, fp32
    long tok_bstride, tok_sstride;   // elements between batch items / streams
    int B, S, L, heads, dh, HLP;
    float scale, eps;
    const float *ln_g, *ln_b, *wq;   // wq fp32 [inner][32]
    const void *wkT, *wvT, *woT;     // T: [32][inner], [32][inner], [inner][32]
    float *mn, *mstats, *k, *v;      // saved fp32: [S][L][32], [S][L][2], [S][L][inner] x2
    void *kq, *kqT, *vo, *voT;       // packed T: [S][HLP][32], [S][32][HLP], [S][HLP][32], [S][32][HLP]
    // All layers of one decoder stack read the SAME tokens, so they are prepared by one launch (blockIdx.y = layer).
    // ls_param: floats between the fp32 parameters of consecutive layers (they sit in the net's flat arena at a
    // constant layer pitch); ls_pack: elements between the packed transposes; outputs are stacked [layer][...].
    long ls_param, ls_pack;
};

template <typename T>
__global__ __launch_bounds__(1024) void xattn_prep_kernel(PrepArgs a) {
    extern __shared__ float sm[];
    const int s = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int L = a.L, inner = a.heads * a.dh, HL = a.heads * L;
    {
        const size_t ly = blockIdx.y;
        a.ln_g += ly * a.ls_param; a.ln_b += ly * a.ls_param; a.wq += ly * a.ls_param;
        a.wkT = reinterpret_cast<const T*>(a.wkT) + ly * a.ls_pack;
        a.wvT = reinterpret_cast<const T*>(a.wvT) + ly * a.ls_pack;
        a.woT = reinterpret_cast<const T*>(a.woT) + ly * a.ls_pack;
        a.mn += ly * a.S * L * D; a.mstats += ly * a.S * L * 2;
        a.k += ly * a.S * L * inner; a.v += ly * a.S * L * inner;
        a.kq = reinterpret_cast<T*>(a.kq) + ly * a.S * a.HLP * D; a.kqT = reinterpret_cast<T*>(a.kqT) + ly * a.S * a.HLP * D;
        a.vo = reinterpret_cast<T*>(a.vo) + ly * a.S * a.HLP * D; a.voT = reinterpret_cast<T*>(a.voT) + ly * a.S * a.HLP * D;
    }
    float* smn = sm;                    // [L][32]
    float* sk = smn + L * D;            // [L][inner]
    float* sv = sk + L * inner;         // [L][inner]
    const float* m = reinterpret_cast<const float*>(a.tok) + (size_t)(s % a.B) * a.tok_bstride + (size_t)(s / a.B) * a.tok_sstride;
    // LayerNorm of the L token rows (shared LN of PreNorm2, help_funcs.py:48-49): 32 lanes per row
    {
        const int l = tid >> 5, c = tid & 31;
        if (l < L) {
            const float x = ldf(m + l * D + c);
            float t = x;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            const float mu = t * (1.f / D);
            float q = (x - mu) * (x - mu);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
            const float rstd = rsqrtf(q * (1.f / D) + a.eps);
            const float v = (x - mu) * rstd * a.ln_g[c] + a.ln_b[c];
            smn[l * D + c] = v;
            a.mn[((size_t)s * L + l) * D + c] = v;
            if (c == 0) { a.mstats[((size_t)s * L + l) * 2] = mu; a.mstats[((size_t)s * L + l) * 2 + 1] = rstd; }
        }
    }
    __syncthreads();
    const T* wkT = reinterpret_cast<const T*>(a.wkT);
    const T* wvT = reinterpret_cast<const T*>(a.wvT);
    for (int hd = tid; hd < inner; hd += NT) {
        float kk[8], vv[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) { kk[l] = 0.f; vv[l] = 0.f; }
#pragma unroll 8                         // eight iterations' weight loads in flight (the adds stay in order)
        for (int c = 0; c < D; ++c) {
            const float wk = ldf(wkT + (size_t)c * inner + hd), wv = ldf(wvT + (size_t)c * inner + hd);
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < L) { kk[l] += wk * smn[l * D + c]; vv[l] += wv * smn[l * D + c]; }
        }
#pragma unroll
        for (int l = 0; l < 8; ++l)
            if (l < L) {
                sk[l * inner + hd] = kk[l]; sv[l * inner + hd] = vv[l];
                a.k[((size_t)s * L + l) * inner + hd] = kk[l];
                a.v[((size_t)s * L + l) * inner + hd] = vv[l];
            }
    }
    __syncthreads();
    T* kq = reinterpret_cast<T*>(a.kq) + (size_t)s * a.HLP * D;
    T* kqT = reinterpret_cast<T*>(a.kqT) + (size_t)s * a.HLP * D;
    T* vo = reinterpret_cast<T*>(a.vo) + (size_t)s * a.HLP * D;
    T* voT = reinterpret_cast<T*>(a.voT) + (size_t)s * a.HLP * D;
    const T* woT = reinterpret_cast<const T*>(a.woT);
    for (int i = tid; i < a.HLP * D; i += NT) {
        const int hl = i / D, c = i % D;
        float q = 0.f, o = 0.f;
        if (hl < HL) {
            const int h = hl / L, l = hl % L;
#pragma unroll 8
            for (int d = 0; d < a.dh; ++d) {
                const int hd = h * a.dh + d;
                q += a.wq[hd * D + c] * sk[l * inner + hd];
                o += ldf(woT + (size_t)hd * D + c) * sv[l * inner + hd];
            }
            q *= a.scale;
        }
        stf(kq + hl * D + c, q);
        stf(kqT + c * a.HLP + hl, q);
        stf(vo + hl * D + c, o);
        stf(voT + c * a.HLP + hl, o);
    }
}

// ---- the same preparation on the matrix cores (bf16 nets, L = 4 tokens, images in groups of 4) -------------------------------
// The kernel above is one workgroup per (image, layer): every image's workgroup re-reads the layer's four weight matrices
// with 2- and 4-byte loads in dependent batches -- 400 MB of L2 -> L1 traffic and ~190 load instructions per thread for
// 0.3 MFLOP: 30 us per launch, on the critical path of the step.  Here a workgroup takes FOUR images = 16 token columns, i.e.
// the N of one MFMA, and each wave its share of the heads:
//   k_h, v_h  = W_{k,v}[head rows] . LN(tokens)      rows = inner index (16 per MFMA), K = the 32 channels
//   kq_h      = scale Wq_h^T . k_h                   rows = channel, K = the head's inner indices
//   vo_h      = Wo_h . v_h
// The D layout of two consecutive 16-row results IS the B operand of the next product (the k-permutation kappa of
// decoder_fused.hip), so nothing goes through LDS; weight fragments come straight from the fp32 masters (rounded to bf16 on
// the way, as the packed copies are) or from the stacked bf16 transposes.  k and v are rounded to bf16 between the two
// products (the separate-kernel path rounds them there too).
struct PrepMArgs {
    PrepArgs a;
    const float *wk, *wv, *wo;      // fp32 masters [inner][32], [inner][32], [32][inner] (first layer; + ls_param per layer)
    const bf16* wqT;                // stacked transposes [layers][32][inner]
};
union PU8 {
    uint4 u;
    uint2 h[2];
    s16x8 v;
};
__device__ __forceinline__ s16x8 ppack8(const float (&a)[4], const float (&b)[4]) {
    PU8 r;
    r.u.x = f2bf2(a[0], a[1]); r.u.y = f2bf2(a[2], a[3]); r.u.z = f2bf2(b[0], b[1]); r.u.w = f2bf2(b[2], b[3]);
    return r.v;
}
__device__ __forceinline__ s16x8 ppack8(const f32x4& a, const f32x4& b) {
    PU8 r;
    r.u.x = f2bf2(a[0], a[1]); r.u.y = f2bf2(a[2], a[3]); r.u.z = f2bf2(b[0], b[1]); r.u.w = f2bf2(b[2], b[3]);
    return r.v;
}
// A / B fragment of a 32-wide k range starting at `row32` (fp32 or bf16 source): elements kappa(g, e) = g*4+e | 16+g*4+(e-4)
__device__ __forceinline__ s16x8 frag32(const float* row32, int g) {
    const float4 lo = *reinterpret_cast<const float4*>(row32 + g * 4), hi = *reinterpret_cast<const float4*>(row32 + 16 + g * 4);
    const float a[4] = {lo.x, lo.y, lo.z, lo.w}, b[4] = {hi.x, hi.y, hi.z, hi.w};
    return ppack8(a, b);
}
__device__ __forceinline__ s16x8 frag32(const bf16* row32, int g) {
    PU8 r;
    r.h[0] = *reinterpret_cast<const uint2*>(row32 + g * 4);
    r.h[1] = *reinterpret_cast<const uint2*>(row32 + 16 + g * 4);
    return r.v;
}
__device__ __forceinline__ float rows4_sum(float v) {      // over the 4 rows of 16 lanes (decoder_fused.hip: group4_sum)
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void st4bf(bf16* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(f2bf2(v[0], v[1]), f2bf2(v[2], v[3]));
}

// -DXP_TIMING (debug builds only, tools/xprep_timeline.py): wall-clock stamps of lane 0 of wave 0
#ifdef XP_TIMING
__device__ long long g_xpt[256 * 16];
#define XP_T(k) do { if (threadIdx.x == 0) g_xpt[bid.x * 16 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define XP_T(k) do { } while (0)
#endif
template <int DH>        // dim_head: 32 or 64
__device__ __forceinline__ void xattn_prep_mfma_body(PrepMArgs m, const uint3 bid) {
    PrepArgs& a = m.a;
    constexpr int L = 4, NBLK = DH / 16, NKS = DH / 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    const int inner = a.heads * DH, HL = a.heads * L;
    {
        const size_t ly = bid.y;
        a.ln_g += ly * a.ls_param; a.ln_b += ly * a.ls_param;
        m.wk += ly * a.ls_param; m.wv += ly * a.ls_param; m.wo += ly * a.ls_param;
        m.wqT += ly * a.ls_pack;
        a.mn += ly * a.S * L * D; a.mstats += ly * a.S * L * 2;
        a.k += ly * a.S * L * inner; a.v += ly * a.S * L * inner;
        a.kq = reinterpret_cast<bf16*>(a.kq) + ly * a.S * a.HLP * D; a.kqT = reinterpret_cast<bf16*>(a.kqT) + ly * a.S * a.HLP * D;
        a.vo = reinterpret_cast<bf16*>(a.vo) + ly * a.S * a.HLP * D; a.voT = reinterpret_cast<bf16*>(a.voT) + ly * a.S * a.HLP * D;
    }
    // column pl of the MFMAs = token l of image s
    XP_T(0);
    const int s = bid.x * 4 + (pl >> 2), l = pl & 3;
    const float* tok = reinterpret_cast<const float*>(a.tok) + (size_t)(s % a.B) * a.tok_bstride + (size_t)(s / a.B) * a.tok_sstride + l * D;
    float x[2][4], mnv[2][4];
    {
        const float4 lo = *reinterpret_cast<const float4*>(tok + g * 4), hi = *reinterpret_cast<const float4*>(tok + 16 + g * 4);
        x[0][0] = lo.x; x[0][1] = lo.y; x[0][2] = lo.z; x[0][3] = lo.w;
        x[1][0] = hi.x; x[1][1] = hi.y; x[1][2] = hi.z; x[1][3] = hi.w;
    }
    // LayerNorm of the token row (shared LN of PreNorm2, help_funcs.py:48-49): 8 channels per lane, 4 lane rows per token
    float sm = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) sm += x[h][j];
    const float mu = rows4_sum(sm) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = x[h][j] - mu; q += d * d; }
    const float rstd = rsqrtf(rows4_sum(q) * (1.f / D) + a.eps);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = h * 16 + g * 4 + j;
            mnv[h][j] = (x[h][j] - mu) * rstd * a.ln_g[c] + a.ln_b[c];
        }
    if (wv == 0) {
        float* mo = a.mn + ((size_t)s * L + l) * D;
        *reinterpret_cast<float4*>(mo + g * 4) = make_float4(mnv[0][0], mnv[0][1], mnv[0][2], mnv[0][3]);
        *reinterpret_cast<float4*>(mo + 16 + g * 4) = make_float4(mnv[1][0], mnv[1][1], mnv[1][2], mnv[1][3]);
        if (g == 0) { a.mstats[((size_t)s * L + l) * 2] = mu; a.mstats[((size_t)s * L + l) * 2 + 1] = rstd; }
    }
    const s16x8 bmn = ppack8(mnv[0], mnv[1]);
    XP_T(1);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    bf16* kq = reinterpret_cast<bf16*>(a.kq) + (size_t)s * a.HLP * D;
    bf16* kqT = reinterpret_cast<bf16*>(a.kqT) + (size_t)s * a.HLP * D;
    bf16* vo = reinterpret_cast<bf16*>(a.vo) + (size_t)s * a.HLP * D;
    bf16* voT = reinterpret_cast<bf16*>(a.voT) + (size_t)s * a.HLP * D;
    for (int h = wv; h < a.heads; h += 4) {
        // every weight fragment of the head is requested before the first product: the stores below may alias the weights for all
        // the compiler knows, so loads placed at their use sat behind them -- four L2 round trips per head in a 16-workgroup launch
        s16x8 kA[NBLK], vA[NBLK], qA[2][NKS], oA[2][NKS];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int hd = h * DH + b * 16 + pl;                     // A row of this lane
            kA[b] = frag32(m.wk + (size_t)hd * D, g);
            vA[b] = frag32(m.wv + (size_t)hd * D, g);
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const int c_row = rb * 16 + pl;                      // A row of this lane: output channel
                qA[rb][ks] = frag32(m.wqT + (size_t)c_row * inner + h * DH + ks * 32, g);
                oA[rb][ks] = frag32(m.wo + (size_t)c_row * inner + h * DH + ks * 32, g);
            }
        f32x4 kb[NBLK], vb[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            kb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kA[b], bmn, zero4, 0, 0, 0);
            vb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vA[b], bmn, zero4, 0, 0, 0);
        }
        XP_T(2 + 4 * (h >> 2));
        // saved for the backward: k, v [S][L][inner] fp32 (this lane: 4 consecutive inner indices of its token)
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const size_t o = ((size_t)s * L + l) * inner + h * DH + b * 16 + g * 4;
            *reinterpret_cast<float4*>(a.k + o) = make_float4(kb[b][0], kb[b][1], kb[b][2], kb[b][3]);
            *reinterpret_cast<float4*>(a.v + o) = make_float4(vb[b][0], vb[b][1], vb[b][2], vb[b][3]);
        }
        s16x8 kB[NKS], vB[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { kB[ks] = ppack8(kb[2 * ks], kb[2 * ks + 1]); vB[ks] = ppack8(vb[2 * ks], vb[2 * ks + 1]); }
        XP_T(3 + 4 * (h >> 2));
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x4 aq = zero4, ao = zero4;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                aq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qA[rb][ks], kB[ks], aq, 0, 0, 0);
                ao = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oA[rb][ks], vB[ks], ao, 0, 0, 0);
            }
            const int hl = h * L + l, c0 = rb * 16 + g * 4;          // this lane: channels c0 .. c0 + 3 of row hl
            float qv[4], ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { qv[j] = aq[j] * a.scale; ov[j] = ao[j]; }
            st4bf(kq + hl * D + c0, qv);
            st4bf(vo + hl * D + c0, ov);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                stf(kqT + (c0 + j) * a.HLP + hl, qv[j]);
                stf(voT + (c0 + j) * a.HLP + hl, ov[j]);
            }
            XP_T(4 + rb + 4 * (h >> 2));
        }
    }
    XP_T(10);
    // rows hl >= heads * L of the HLP-row operands are zero
    if (HL < a.HLP) {
        const int pad = a.HLP - HL;
        for (int i = tid; i < 4 * pad * D; i += 256) {
            const int c = i % D, r = (i / D) % pad, si = i / (D * pad);
            const size_t base = (size_t)(bid.x * 4 + si) * a.HLP * D;
            bf16* q0 = reinterpret_cast<bf16*>(a.kq) + base; bf16* o0 = reinterpret_cast<bf16*>(a.vo) + base;
            bf16* qT = reinterpret_cast<bf16*>(a.kqT) + base; bf16* oT = reinterpret_cast<bf16*>(a.voT) + base;
            stf(q0 + (HL + r) * D + c, 0.f); stf(o0 + (HL + r) * D + c, 0.f);
            stf(qT + c * a.HLP + HL + r, 0.f); stf(oT + c * a.HLP + HL + r, 0.f);
        }
    }
}
template <int DH>
__global__ __launch_bounds__(256) void xattn_prep_mfma_kernel(PrepMArgs m) { xattn_prep_mfma_body<DH>(m, XB_BID); }
template <int DH>
__global__ __launch_bounds__(256) void xattn_prep_mfma_multi_kernel(XMulti<PrepMArgs> mm) {
    uint3 bid;
    const int j = xb_locate(mm, bid);
    xattn_prep_mfma_body<DH>(mm.a[j], bid);
}

struct PrepBwdArgs {
    long tok_bstride, tok_sstride;
    int B, S, L, heads, dh, HLP;
    float scale;
    const void* tok;                 // fp32, forward token rows (for the LN backward)
    void* dtok;                      // fp32, accumulated in place (same addressing as tok)
    const float *ln_g, *wk, *wv, *wo;   // fp32 masters: wk, wv [inner][32]; wo [32][inner]
    const void* wqT;                 // T [32][inner]
    const float *mn, *mstats;
    const float *dkq, *dvoT;         // [S][HLP][32], [S][32][HLP] fp32 (per-image weight gradients)
    float *dk, *dv;                  // out [S][L][inner]
    float* ln_partial;               // out [S][2][32]  (dgamma, dbeta) contributions
    // stacked form (blockIdx.y = layer, see PrepArgs): the layers' token gradients cannot be accumulated in place
    // by concurrent workgroups, so each layer writes dtok_part[layer][S][L][32] and dtok_reduce_kernel adds them up
    float* dtok_part;                // null: accumulate into dtok directly (single layer)
    long ls_param, ls_pack;
};

template <typename T>
__global__ __launch_bounds__(1024) void xattn_prep_bwd_kernel(PrepBwdArgs a) {
    extern __shared__ float sm[];
    const int s = blockIdx.x, tid = threadIdx.x, NT = blockDim.x;
    const int L = a.L, inner = a.heads * a.dh;
    {
        const size_t ly = blockIdx.y;
        a.ln_g += ly * a.ls_param; a.wk += ly * a.ls_param; a.wv += ly * a.ls_param; a.wo += ly * a.ls_param;
        a.wqT = reinterpret_cast<const T*>(a.wqT) + ly * a.ls_pack;
        a.mn += ly * a.S * L * D; a.mstats += ly * a.S * L * 2;
        a.dkq += ly * a.S * a.HLP * D; a.dvoT += ly * a.S * a.HLP * D;
        a.dk += ly * a.S * L * inner; a.dv += ly * a.S * L * inner;
        a.ln_partial += ly * a.S * 2 * D;
        if (a.dtok_part) a.dtok_part += ly * a.S * L * D;
    }
    float* sdk = sm;                 // [L][inner]
    float* sdv = sdk + L * inner;    // [L][inner]
    float* sdmn = sdv + L * inner;   // [L][32]
    float* sgk = sdmn + L * D;       // [HLP][32]   staged dKq
    float* sgv = sgk + a.HLP * D;    // [32][HLP]   staged dVoT
    float* sred = sgv + a.HLP * D;   // [8][L][32]
    for (int i = tid; i < a.HLP * D; i += NT) {
        sgk[i] = a.dkq[(size_t)s * a.HLP * D + i];
        sgv[i] = a.dvoT[(size_t)s * a.HLP * D + i];
    }
    __syncthreads();
    const T* wqT = reinterpret_cast<const T*>(a.wqT);
    for (int hd = tid; hd < inner; hd += NT) {
        const int h = hd / a.dh;
        float gk[8], gv[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) { gk[l] = 0.f; gv[l] = 0.f; }
#pragma unroll 8
        for (int c = 0; c < D; ++c) {
            const float wq = ldf(wqT + (size_t)c * inner + hd), wo = a.wo[(size_t)c * inner + hd];
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < L) { gk[l] += sgk[(h * L + l) * D + c] * wq; gv[l] += sgv[c * a.HLP + h * L + l] * wo; }
        }
#pragma unroll
        for (int l = 0; l < 8; ++l)
            if (l < L) {
                const float t = gk[l] * a.scale;
                sdk[l * inner + hd] = t; sdv[l * inner + hd] = gv[l];
                a.dk[((size_t)s * L + l) * inner + hd] = t;
                a.dv[((size_t)s * L + l) * inner + hd] = gv[l];
            }
    }
    __syncthreads();
    // dmn[l][c] = sum_hd dk[l][hd] wk[hd][c] + dv[l][hd] wv[hd][c]: 8 hd-chunks x 32 channels, then LDS reduce
    {
        const int c = tid & 31, ch = tid >> 5, nch = NT >> 5, per = (inner + nch - 1) / nch;
        float acc[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[l] = 0.f;
#pragma unroll 8
        for (int hd = ch * per; hd < (ch + 1) * per && hd < inner; ++hd) {
            const float wk = a.wk[(size_t)hd * D + c], wv = a.wv[(size_t)hd * D + c];
#pragma unroll
            for (int l = 0; l < 8; ++l)
                if (l < L) acc[l] += sdk[l * inner + hd] * wk + sdv[l * inner + hd] * wv;
        }
#pragma unroll
        for (int l = 0; l < 8; ++l)
            if (l < L) sred[(ch * L + l) * D + c] = acc[l];
    }
    __syncthreads();
    for (int i = tid; i < L * D; i += NT) {
        float t = 0.f;
        for (int ch = 0; ch < (NT >> 5); ++ch) t += sred[ch * L * D + i];
        sdmn[i] = t;
    }
    __syncthreads();
    // LayerNorm backward on the L rows (32 lanes per row); accumulate into dtok; per-image dgamma/dbeta
    const float* m = reinterpret_cast<const float*>(a.tok) + (size_t)(s % a.B) * a.tok_bstride + (size_t)(s / a.B) * a.tok_sstride;
    float* dm = reinterpret_cast<float*>(a.dtok) + (size_t)(s % a.B) * a.tok_bstride + (size_t)(s / a.B) * a.tok_sstride;
    {
        const int l = tid >> 5, c = tid & 31;
        float pg = 0.f, pb = 0.f;
        if (l < L) {
            const float mu = a.mstats[((size_t)s * L + l) * 2], rstd = a.mstats[((size_t)s * L + l) * 2 + 1];
            const float xh = (ldf(m + l * D + c) - mu) * rstd, g = sdmn[l * D + c], gh = g * a.ln_g[c];
            float sa = gh, sb = gh * xh;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); sb += __shfl_xor(sb, o, 64); }
            const float dt = rstd * (gh - (sa + xh * sb) * (1.f / D));
            if (a.dtok_part) a.dtok_part[((size_t)s * L + l) * D + c] = dt;
            else stf(dm + l * D + c, ldf(dm + l * D + c) + dt);
            pg = g * xh; pb = g;
        }
        // sum the L rows' contributions per channel through LDS (rows live in different half-waves)
        if (l < 8) {
            sred[(0 * 8 + l) * D + c] = pg;
            sred[(1 * 8 + l) * D + c] = pb;
        }
    }
    __syncthreads();
    if (tid < 2 * D) {
        const int which = tid / D, c = tid % D;
        float t = 0.f;
        for (int l = 0; l < L; ++l) t += sred[(which * 8 + l) * D + c];
        a.ln_partial[((size_t)s * 2 + which) * D + c] = t;
    }
}

// ---- the backward of the preparation on the matrix cores (same decomposition as xattn_prep_mfma_kernel) ----------------------
//   dk_h = scale Wq[head rows] . dKq_h^T        rows = inner index, K = channel      (B operand: the image's dKq rows, bf16)
//   dv_h = Wo^T[head rows] . dVoT_h
//   dmn  = sum_h Wk_h^T . dk_h + Wv_h^T . dv_h   rows = channel, K = the head's inner indices (B = the D layout of dk / dv)
// then the shared LayerNorm's backward per token row (lane-local + the cross-row swaps), its per-image dgamma / dbeta
// partials, and the token gradient.  The four waves' dmn parts meet in LDS.
struct PrepBwdMArgs {
    PrepBwdArgs a;
    const float* wq;                 // fp32 master [inner][32] (first layer)
    const bf16 *woT, *wkT, *wvT;     // stacked transposes [layers][inner][32], [layers][32][inner] x2
};
template <int DH>
__device__ __forceinline__ void xattn_prep_bwd_mfma_body(PrepBwdMArgs m, const uint3 bid) {
    PrepBwdArgs& a = m.a;
    constexpr int L = 4, NBLK = DH / 16, NKS = DH / 32;
    __shared__ __attribute__((aligned(16))) float red[4][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    const int inner = a.heads * DH;
    {
        const size_t ly = bid.y;
        a.ln_g += ly * a.ls_param; m.wq += ly * a.ls_param;
        m.woT += ly * a.ls_pack; m.wkT += ly * a.ls_pack; m.wvT += ly * a.ls_pack;
        a.mn += ly * a.S * L * D; a.mstats += ly * a.S * L * 2;
        a.dkq += ly * a.S * a.HLP * D; a.dvoT += ly * a.S * a.HLP * D;
        a.dk += ly * a.S * L * inner; a.dv += ly * a.S * L * inner;
        a.ln_partial += ly * a.S * 2 * D;
        if (a.dtok_part) a.dtok_part += ly * a.S * L * D;
    }
    const int s = bid.x * 4 + (pl >> 2), l = pl & 3;
    const float* dkq = a.dkq + (size_t)s * a.HLP * D;
    const float* dvoT = a.dvoT + (size_t)s * a.HLP * D;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2] = {zero4, zero4};
    for (int h = wv; h < a.heads; h += 4) {
        const int hl = h * L + l;
        const s16x8 bdq = frag32(dkq + (size_t)hl * D, g);
        float o0[4], o1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = dvoT[(size_t)(g * 4 + j) * a.HLP + hl];
            o1[j] = dvoT[(size_t)(16 + g * 4 + j) * a.HLP + hl];
        }
        const s16x8 bdo = ppack8(o0, o1);
        // (every weight fragment of the head requested before the first product, as in the forward: the dk / dv stores may alias them)
        s16x8 qA[NBLK], oA[NBLK], kA[NKS][2], vA[NKS][2];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const int hd = h * DH + b * 16 + pl;
            qA[b] = frag32(m.wq + (size_t)hd * D, g);
            oA[b] = frag32(m.woT + (size_t)hd * D, g);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const size_t wo_ = (size_t)(rb * 16 + pl) * inner + h * DH + ks * 32;
                kA[ks][rb] = frag32(m.wkT + wo_, g);
                vA[ks][rb] = frag32(m.wvT + wo_, g);
            }
        f32x4 dkb[NBLK], dvb[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            dkb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qA[b], bdq, zero4, 0, 0, 0);
            dvb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oA[b], bdo, zero4, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) dkb[b][j] *= a.scale;
            const size_t o = ((size_t)s * L + l) * inner + h * DH + b * 16 + g * 4;
            *reinterpret_cast<float4*>(a.dk + o) = make_float4(dkb[b][0], dkb[b][1], dkb[b][2], dkb[b][3]);
            *reinterpret_cast<float4*>(a.dv + o) = make_float4(dvb[b][0], dvb[b][1], dvb[b][2], dvb[b][3]);
        }
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const s16x8 dkB = ppack8(dkb[2 * ks], dkb[2 * ks + 1]), dvB = ppack8(dvb[2 * ks], dvb[2 * ks + 1]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kA[ks][rb], dkB, acc[rb], 0, 0, 0);
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vA[ks][rb], dvB, acc[rb], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
        *reinterpret_cast<float4*>(&red[wv][rb][lane][0]) = make_float4(acc[rb][0], acc[rb][1], acc[rb][2], acc[rb][3]);
    __syncthreads();
    if (wv != 0) return;
    float dmn[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float4 p0 = *reinterpret_cast<const float4*>(&red[0][rb][lane][0]), p1 = *reinterpret_cast<const float4*>(&red[1][rb][lane][0]);
        const float4 p2 = *reinterpret_cast<const float4*>(&red[2][rb][lane][0]), p3 = *reinterpret_cast<const float4*>(&red[3][rb][lane][0]);
        dmn[rb][0] = ((p0.x + p1.x) + p2.x) + p3.x; dmn[rb][1] = ((p0.y + p1.y) + p2.y) + p3.y;
        dmn[rb][2] = ((p0.z + p1.z) + p2.z) + p3.z; dmn[rb][3] = ((p0.w + p1.w) + p2.w) + p3.w;
    }
    // LayerNorm backward of token row (s, l): this lane holds channels rb * 16 + g * 4 + j
    const size_t toff = (size_t)(s % a.B) * a.tok_bstride + (size_t)(s / a.B) * a.tok_sstride + l * D;
    const float* tok = reinterpret_cast<const float*>(a.tok) + toff;
    const float mu = a.mstats[((size_t)s * L + l) * 2], rstd = a.mstats[((size_t)s * L + l) * 2 + 1];
    float xh[2][4], gh[2][4], sa = 0.f, sb = 0.f;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float4 xv = *reinterpret_cast<const float4*>(tok + rb * 16 + g * 4);
        const float4 gv = *reinterpret_cast<const float4*>(a.ln_g + rb * 16 + g * 4);
        const float x4[4] = {xv.x, xv.y, xv.z, xv.w}, g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[rb][j] = (x4[j] - mu) * rstd;
            gh[rb][j] = dmn[rb][j] * g4[j];
            sa += gh[rb][j];
            sb += gh[rb][j] * xh[rb][j];
        }
    }
    sa = rows4_sum(sa);
    sb = rows4_sum(sb);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        float dt[4], pg[4], pb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dt[j] = rstd * (gh[rb][j] - (sa + xh[rb][j] * sb) * (1.f / D));
            pg[j] = quad_sum(dmn[rb][j] * xh[rb][j]);          // over the image's L tokens: per-image dgamma / dbeta parts
            pb[j] = quad_sum(dmn[rb][j]);
        }
        const int c0 = rb * 16 + g * 4;
        if (a.dtok_part) {
            *reinterpret_cast<float4*>(a.dtok_part + ((size_t)s * L + l) * D + c0) = make_float4(dt[0], dt[1], dt[2], dt[3]);
        } else {
            float* dm = reinterpret_cast<float*>(a.dtok) + toff + c0;
            const float4 o = *reinterpret_cast<const float4*>(dm);
            *reinterpret_cast<float4*>(dm) = make_float4(o.x + dt[0], o.y + dt[1], o.z + dt[2], o.w + dt[3]);
        }
        if (l == 0) {
            *reinterpret_cast<float4*>(a.ln_partial + ((size_t)s * 2 + 0) * D + c0) = make_float4(pg[0], pg[1], pg[2], pg[3]);
            *reinterpret_cast<float4*>(a.ln_partial + ((size_t)s * 2 + 1) * D + c0) = make_float4(pb[0], pb[1], pb[2], pb[3]);
        }
    }
}
template <int DH>
__global__ __launch_bounds__(256) void xattn_prep_bwd_mfma_kernel(PrepBwdMArgs m) { xattn_prep_bwd_mfma_body<DH>(m, XB_BID); }
template <int DH>
__global__ __launch_bounds__(256) void xattn_prep_bwd_mfma_multi_kernel(XMulti<PrepBwdMArgs> mm) {
    uint3 bid;
    const int j = xb_locate(mm, bid);
    xattn_prep_bwd_mfma_body<DH>(mm.a[j], bid);
}

// weight gradients of to_q / to_k / to_v / to_out: one workgroup per (matrix, hd); 8 image phases x 32 channels,
// LDS reduce.  blockIdx.y == 4: the shared LayerNorm's dgamma / dbeta from the per-image partials.
struct PrepWgArgs {
    int S, L, heads, dh, HLP;
    float scale;
    const float *mn, *k, *v, *dk, *dv, *dkq, *dvoT, *ln_partial;
    float *dwq, *dwk, *dwv, *dwo, *dln_g, *dln_b;
    int accumulate;
    long ls_param;                   // stacked form: blockIdx.z = layer (inputs stacked, gradients at the arena's layer pitch)
    int ln_only;                     // 1: grid (2, 1, layers) -- only the LayerNorm sums (the matrices went to the MFMA kernel)
};
__device__ __forceinline__ void xattn_prep_wgrad_body(PrepWgArgs a, const uint3 bid) {
    // one workgroup per (matrix, block of HB = 8 consecutive inner columns hd): 8 image phases x 32 channels, every
    // thread carries the 8 columns (their k / v / dk / dv values are two 16-byte broadcast loads).  One workgroup per
    // single column was 2560 - 10240 tiny workgroups per launch: dispatch-bound (72 us for 17 MFLOP).
    constexpr int HB = 8;
    __shared__ float red[8][HB][33];
    const int inner = a.heads * a.dh, L = a.L;
    {
        const size_t ly = bid.z;
        a.mn += ly * a.S * L * D; a.k += ly * a.S * L * inner; a.v += ly * a.S * L * inner;
        a.dk += ly * a.S * L * inner; a.dv += ly * a.S * L * inner;
        a.dkq += ly * a.S * a.HLP * D; a.dvoT += ly * a.S * a.HLP * D; a.ln_partial += ly * a.S * 2 * D;
        a.dwq += ly * a.ls_param; a.dwk += ly * a.ls_param; a.dwv += ly * a.ls_param; a.dwo += ly * a.ls_param;
        a.dln_g += ly * a.ls_param; a.dln_b += ly * a.ls_param;
    }
    const int c = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int which = a.ln_only ? 4 : bid.y, hd0 = bid.x * HB;
    float acc[HB];
#pragma unroll
    for (int j = 0; j < HB; ++j) acc[j] = 0.f;
    if (which == 4) {
        if (bid.x >= 2) return;                           // bid.x = 0: dgamma, 1: dbeta
        for (int s = ph; s < a.S; s += 8) acc[0] += a.ln_partial[((size_t)s * 2 + bid.x) * D + c];
    } else {
        const int h = hd0 / a.dh;                              // HB divides dh: the block lies inside one head
        const float* colsrc = which == 0 ? a.k : (which == 1 ? a.dk : (which == 2 ? a.dv : a.v));
        for (int s = ph; s < a.S; s += 8)
            for (int l = 0; l < L; ++l) {
                const size_t r = (size_t)s * L + l;
                float rowv;                                    // the per-channel factor of this (image, token)
                if (which == 0) rowv = a.dkq[((size_t)s * a.HLP + h * L + l) * D + c];
                else if (which == 3) rowv = a.dvoT[((size_t)s * D + c) * a.HLP + h * L + l];
                else rowv = a.mn[r * D + c];
                const float4 c0 = *reinterpret_cast<const float4*>(colsrc + r * inner + hd0);
                const float4 c1 = *reinterpret_cast<const float4*>(colsrc + r * inner + hd0 + 4);
                // (operand order of the former one-column form: which 0 / 3: row * col, which 1 / 2: col * row)
                acc[0] += rowv * c0.x; acc[1] += rowv * c0.y; acc[2] += rowv * c0.z; acc[3] += rowv * c0.w;
                acc[4] += rowv * c1.x; acc[5] += rowv * c1.y; acc[6] += rowv * c1.z; acc[7] += rowv * c1.w;
            }
    }
#pragma unroll
    for (int j = 0; j < HB; ++j) red[ph][j][c] = acc[j];
    __syncthreads();
    if (which == 4) {
        if (ph == 0) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += red[r][0][c];
            float* out = bid.x == 0 ? a.dln_g : a.dln_b;
            out[c] += t;                                       // the pixel-side LN backward wrote its part already
        }
        return;
    }
    {
        const int j = ph;                                      // phase group j finishes column hd0 + j
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][j][c];
        if (which == 0) t *= a.scale;
        const int hd = hd0 + j;
        float* out = which == 0 ? a.dwq : (which == 1 ? a.dwk : (which == 2 ? a.dwv : a.dwo));
        const size_t o = which == 3 ? (size_t)c * inner + hd : (size_t)hd * D + c;     // to_out weight is [32][inner]
        if (a.accumulate) out[o] += t; else out[o] = t;
    }
}
__global__ __launch_bounds__(256) void xattn_prep_wgrad_kernel(PrepWgArgs a) { xattn_prep_wgrad_body(a, XB_BID); }
__global__ __launch_bounds__(256) void xattn_prep_wgrad_multi_kernel(XMulti<PrepWgArgs> mm) {
    uint3 bid;
    const int j = xb_locate(mm, bid);
    xattn_prep_wgrad_body(mm.a[j], bid);
}

// The same weight gradients on the matrix cores (bf16 nets, dim_head = 64, L = 4, S a multiple of 8): every one of the four is
//   dW[hd][c] = sum over the K = S * L token rows of col[K][hd] * row[K][c]
// (col = k / dk / dv / v, row = dKq rows / LN(tokens) / dVoT columns).  One workgroup per (matrix, 64 inner indices = one head):
// per 32 token rows both operands are staged as bf16 [K][channel] tiles (coalesced 16-byte loads) and read back k-contiguous
// with ds_read_b64_tr_b16 -- the pixel-reduction idiom of decoder_fused.hip; wave w owns inner indices 16 w .. 16 w + 15.
// 2560 workgroups of the FMA form, each a chain of dependent 4-byte loads, become 256 with eight 32-row steps.
constexpr int wg_pitch(int row_bytes) { return ((row_bytes / 32) & 1) ? row_bytes : row_bytes + 32; }
__device__ __forceinline__ s16x8 wg_tile_frag(const unsigned char* tile, int pitch, int cs, int pl, int g) {
    const unsigned char* base = tile + (g * 4 + (pl >> 2)) * pitch + (cs * 16 + (pl & 3) * 4) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 16 * pitch));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void xattn_prep_wgrad_mfma_body(PrepWgArgs a, const uint3 bid) {
    constexpr int L = 4, DH = 64, CP = wg_pitch(128), RP = wg_pitch(64);
    __shared__ __attribute__((aligned(16))) unsigned char colT[32 * CP], rowT[32 * RP];
    const int inner = a.heads * DH;
    {
        const size_t ly = bid.z;
        a.mn += ly * a.S * L * D; a.k += ly * a.S * L * inner; a.v += ly * a.S * L * inner;
        a.dk += ly * a.S * L * inner; a.dv += ly * a.S * L * inner;
        a.dkq += ly * a.S * a.HLP * D; a.dvoT += ly * a.S * a.HLP * D;
        a.dwq += ly * a.ls_param; a.dwk += ly * a.ls_param; a.dwv += ly * a.ls_param; a.dwo += ly * a.ls_param;
    }
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    const int which = bid.y, h = bid.x, hd0 = h * DH;
    const float* colsrc = which == 0 ? a.k : (which == 1 ? a.dk : (which == 2 ? a.dv : a.v));
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2] = {zero4, zero4};
    // the tiles of step s0 + 8 are requested before the products of step s0 (registers: three 16-byte pieces per lane): a step was
    // one exposed L2 round trip + two barriers, 8 steps for 64 images in a 32-workgroup launch
    float4 cv[2], rv;
    auto request = [&](int s0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {       // col tile: 32 token rows x 64 inner indices (two 16-byte pieces per thread)
            const int r = (tid >> 4) + 16 * i, q = tid & 15;
            cv[i] = *reinterpret_cast<const float4*>(colsrc + ((size_t)(s0 * L + r)) * inner + hd0 + q * 4);
        }
        {                                   // row tile: 32 token rows x 32 channels
            const int r = tid >> 3, q = tid & 7, s = s0 + (r >> 2), l = r & 3;
            if (which == 0) rv = *reinterpret_cast<const float4*>(a.dkq + ((size_t)s * a.HLP + h * L + l) * D + q * 4);
            else if (which == 3) {
                const float* p = a.dvoT + ((size_t)s * D + q * 4) * a.HLP + h * L + l;
                rv = make_float4(p[0], p[a.HLP], p[2 * a.HLP], p[3 * a.HLP]);
            } else rv = *reinterpret_cast<const float4*>(a.mn + ((size_t)s * L + l) * D + q * 4);
        }
    };
    request(0);
    for (int s0 = 0; s0 < a.S; s0 += 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (tid >> 4) + 16 * i, q = tid & 15;
            *reinterpret_cast<uint2*>(colT + r * CP + q * 8) = make_uint2(f2bf2(cv[i].x, cv[i].y), f2bf2(cv[i].z, cv[i].w));
        }
        {
            const int r = tid >> 3, q = tid & 7;
            *reinterpret_cast<uint2*>(rowT + r * RP + q * 8) = make_uint2(f2bf2(rv.x, rv.y), f2bf2(rv.z, rv.w));
        }
        __syncthreads();
        if (s0 + 8 < a.S) request(s0 + 8);
        const s16x8 fa = wg_tile_frag(colT, CP, wv, pl, g);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, wg_tile_frag(rowT, RP, cb, pl, g), acc[cb], 0, 0, 0);
        __syncthreads();
    }
    float* out = which == 0 ? a.dwq : (which == 1 ? a.dwk : (which == 2 ? a.dwv : a.dwo));
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hd = hd0 + wv * 16 + g * 4 + j, c = cb * 16 + pl;
            const float t = which == 0 ? acc[cb][j] * a.scale : acc[cb][j];
            const size_t o = which == 3 ? (size_t)c * inner + hd : (size_t)hd * D + c;     // to_out weight is [32][inner]
            if (a.accumulate) out[o] += t; else out[o] = t;
        }
}
__global__ __launch_bounds__(256) void xattn_prep_wgrad_mfma_kernel(PrepWgArgs a) { xattn_prep_wgrad_mfma_body(a, XB_BID); }
__global__ __launch_bounds__(256) void xattn_prep_wgrad_mfma_multi_kernel(XMulti<PrepWgArgs> mm) {
    uint3 bid;
    const int j = xb_locate(mm, bid);
    xattn_prep_wgrad_mfma_body(mm.a[j], bid);
}

// dtok[token rows of image s] += sum over the layers of dtok_part[layer][s]   (fixed order: deterministic)
struct DtokArgs {
    const float* part;
    int layers, S, L, B;
    long bstride, sstride;
    float* dtok;
};
__device__ __forceinline__ void dtok_reduce_body(const DtokArgs& a, const uint3 bid) {
    const int s = bid.x, i = threadIdx.x;           // blockDim = L*32
    float t = 0.f;
    for (int ly = 0; ly < a.layers; ++ly) t += a.part[((size_t)ly * a.S + s) * a.L * D + i];
    float* dm = a.dtok + (size_t)(s % a.B) * a.bstride + (size_t)(s / a.B) * a.sstride;
    dm[i] += t;
}
__global__ void dtok_reduce_kernel(DtokArgs a) { dtok_reduce_body(a, XB_BID); }
__global__ void dtok_reduce_multi_kernel(XMulti<DtokArgs> mm) {
    uint3 bid;
    const int j = xb_locate(mm, bid);
    dtok_reduce_body(mm.a[j], bid);
}

// ------------------------------------------------------------------------------------------
// grouped softmax over the L keys of each head: x, y [rows][HLP]; columns >= heads*L are zero
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void softmax_groups_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long rows, int heads, int L,
                                          int HLP) {
    const int gpr = HLP / L;     // groups per row incl. padding groups
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * gpr) return;
    const long r = i / gpr;
    const int h = (int)(i % gpr);
    const T* xp = x + r * HLP + h * L;
    T* yp = y + r * HLP + h * L;
    if (h >= heads) { for (int l = 0; l < L; ++l) stf(yp + l, 0.f); return; }
    float v[16], m = -INFINITY, s = 0.f;
    for (int l = 0; l < L; ++l) { v[l] = ldf(xp + l); m = fmaxf(m, v[l]); }
    for (int l = 0; l < L; ++l) { v[l] = __expf(v[l] - m); s += v[l]; }
    const float inv = 1.f / s;
    for (int l = 0; l < L; ++l) stf(yp + l, v[l] * inv);
}
template <typename T>
__global__ void softmax_groups_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx,
                                          long rows, int heads, int L, int HLP) {
    const int gpr = HLP / L;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * gpr) return;
    const long r = i / gpr;
    const int h = (int)(i % gpr);
    const size_t o = r * HLP + h * L;
    if (h >= heads) { for (int l = 0; l < L; ++l) stf(dx + o + l, 0.f); return; }
    float p[16], g[16], dot = 0.f;
    for (int l = 0; l < L; ++l) { p[l] = ldf(y + o + l); g[l] = ldf(dy + o + l); dot += p[l] * g[l]; }
    for (int l = 0; l < L; ++l) stf(dx + o + l, p[l] * (g[l] - dot));
}

// ------------------------------------------------------------------------------------------
// token self-attention core (encoder): qkv [B*n][3*inner] -> o [B*n][inner], n <= 16 tokens
// one 64-thread workgroup per (batch item, head)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void self_attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                           float* __restrict__ attn /*[B][heads][n][n]*/, int n,
                                                           int heads, int dh, float scale) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x, inner = heads * dh;
    float* q = sm;
    float* k = q + n * dh;
    float* v = k + n * dh;
    float* a = v + n * dh;     // [n][n]
    for (int i = tid; i < n * dh; i += 64) {
        const int t = i / dh, d = i % dh;
        const T* row = qkv + ((size_t)b * n + t) * 3 * inner + h * dh + d;
        q[i] = ldf(row); k[i] = ldf(row + inner); v[i] = ldf(row + 2 * inner);
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += q[i * dh + d] * k[j * dh + d];
        a[e] = s * scale;
    }
    __syncthreads();
    if (tid < n) {
        float m = -INFINITY, s = 0.f;
        for (int j = 0; j < n; ++j) m = fmaxf(m, a[tid * n + j]);
        for (int j = 0; j < n; ++j) { const float e = __expf(a[tid * n + j] - m); a[tid * n + j] = e; s += e; }
        const float inv = 1.f / s;
        for (int j = 0; j < n; ++j) {
            a[tid * n + j] *= inv;
            attn[(((size_t)b * heads + h) * n + tid) * n + j] = a[tid * n + j];
        }
    }
    __syncthreads();
    for (int i = tid; i < n * dh; i += 64) {
        const int t = i / dh, d = i % dh;
        float s = 0.f;
        for (int j = 0; j < n; ++j) s += a[t * n + j] * v[j * dh + d];
        stf(o + ((size_t)b * n + t) * inner + h * dh + d, s);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void self_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ attn,
                                                           const T* __restrict__ dout, T* __restrict__ dqkv, int n,
                                                           int heads, int dh, float scale) {
    extern __shared__ float sm[];
    const int b = blockIdx.x / heads, h = blockIdx.x % heads, tid = threadIdx.x, inner = heads * dh;
    float* q = sm;
    float* k = q + n * dh;
    float* v = k + n * dh;
    float* go = v + n * dh;
    float* a = go + n * dh;    // [n][n]
    float* ds = a + n * n;     // [n][n]
    for (int i = tid; i < n * dh; i += 64) {
        const int t = i / dh, d = i % dh;
        const T* row = qkv + ((size_t)b * n + t) * 3 * inner + h * dh + d;
        q[i] = ldf(row); k[i] = ldf(row + inner); v[i] = ldf(row + 2 * inner);
        go[i] = ldf(dout + ((size_t)b * n + t) * inner + h * dh + d);
    }
    for (int e = tid; e < n * n; e += 64) a[e] = attn[((size_t)b * heads + h) * n * n + e];
    __syncthreads();
    for (int e = tid; e < n * n; e += 64) {
        const int i = e / n, j = e % n;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += go[i * dh + d] * v[j * dh + d];
        ds[e] = s;      // dA
    }
    __syncthreads();
    if (tid < n) {
        float dot = 0.f;
        for (int j = 0; j < n; ++j) dot += a[tid * n + j] * ds[tid * n + j];
        for (int j = 0; j < n; ++j) ds[tid * n + j] = a[tid * n + j] * (ds[tid * n + j] - dot) * scale;
    }
    __syncthreads();
    for (int i = tid; i < n * dh; i += 64) {
        const int t = i / dh, d = i % dh;
        float gq = 0.f, gk = 0.f, gv = 0.f;
        for (int j = 0; j < n; ++j) {
            gq += ds[t * n + j] * k[j * dh + d];
            gk += ds[j * n + t] * q[j * dh + d];
            gv += a[j * n + t] * go[j * dh + d];
        }
        T* row = dqkv + ((size_t)b * n + t) * 3 * inner + h * dh + d;
        stf(row, gq); stf(row + inner, gk); stf(row + 2 * inner, gv);
    }
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)
extern "C" int dh_reduce_partials(const float* partial, long nt, long n, float scale, float* out, int accumulate,
                                  void* stream);

// logits [S*HW][L] fp32, stats [S][L][2], pooled [S][L][32] are saved for backward
static inline int tok_chunks(int HW) { int c = HW / 1024; return c < 1 ? 1 : (c > 64 ? 64 : c); }      // 1024 rows per workgroup
extern "C" long dh_tokenizer_fwd_workspace_size(int S, int HW, int L) { return (long)S * tok_chunks(HW) * L * 32 * 4; }

// recorded tokenizer calls (dh_xprep_batch_*: same switch as the cross-attention preparation).  One batch holds calls of ONE
// (dtype, token_len): a call of another kind issues what is recorded first.
struct TokBatch {
    int nf = 0, nb = 0, dtype_f = 0, L_f = 0, dtype_b = 0, L_b = 0;
    TokJob f[TB_MAXJ], b[TB_MAXJ];
};
static thread_local TokBatch g_tb;
template <typename T, int L, int STAGE> static void tok_launch_stage(const TokJob* jobs, int n, hipStream_t st) {
    TokMulti m;
    m.n = n;
    m.first[0] = 0;
    for (int j = 0; j < n; ++j) { m.j[j] = jobs[j]; m.first[j + 1] = m.first[j] + tok_stage_blocks<L>(jobs[j], STAGE); }
    if (m.first[n] > 0) hipLaunchKernelGGL((tok_multi_kernel<T, L, STAGE>), dim3(m.first[n]), dim3(256), 0, st, m);
}
template <typename T, int L> static void tok_launch_fwd(const TokJob* jobs, int n, hipStream_t st) {
    tok_launch_stage<T, L, 0>(jobs, n, st);
    tok_launch_stage<T, L, 1>(jobs, n, st);
    tok_launch_stage<T, L, 2>(jobs, n, st);
    tok_launch_stage<T, L, 3>(jobs, n, st);
}
template <typename T, int L> static void tok_launch_bwd(const TokJob* jobs, int n, hipStream_t st) {
    tok_launch_stage<T, L, 4>(jobs, n, st);
    tok_launch_stage<T, L, 5>(jobs, n, st);
    tok_launch_stage<T, L, 6>(jobs, n, st);
    tok_launch_stage<T, L, 7>(jobs, n, st);
}
static int tok_issue(bool fwd, int dtype, int L, const TokJob* jobs, int n, hipStream_t st) {
    if (!n) return 0;
#define TOKGO(TT, LL) do { if (fwd) tok_launch_fwd<TT, LL>(jobs, n, st); else tok_launch_bwd<TT, LL>(jobs, n, st); } while (0)
    if (dtype == DH_DTYPE_BF16) { if (L == 4) TOKGO(bf16, 4); else TOKGO(bf16, 8); }
    else { if (L == 4) TOKGO(float, 4); else TOKGO(float, 8); }
#undef TOKGO
    DH_CHECK_LAUNCH(fwd ? "tokenizer_fwd" : "tokenizer_bwd");
    return 0;
}
static int tok_flush_fwd(hipStream_t st) { const int n = g_tb.nf; g_tb.nf = 0; return tok_issue(true, g_tb.dtype_f, g_tb.L_f, g_tb.f, n, st); }
static int tok_flush_bwd(hipStream_t st) { const int n = g_tb.nb; g_tb.nb = 0; return tok_issue(false, g_tb.dtype_b, g_tb.L_b, g_tb.b, n, st); }
static bool xprep_recording();

extern "C" int dh_tokenizer_fwd(int dtype, const void* x, const float* wa, const float* pos, int S, int B, int HW,
                                int L, float* logits, float* stats, float* pooled, float* tok_cat, void* workspace,
                                void* stream) {
    DH_REQUIRE(L == 4 || L == 8, "tokenizer: token_len must be 4 or 8, got %d", L);
    DH_REQUIRE(S % B == 0 && S / B <= 2, "tokenizer: S=%d must be B or 2B (B=%d)", S, B);
    TokJob a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wa = wa; a.pos = pos; a.logits = logits; a.stats = stats; a.pooled = pooled; a.tok_cat = tok_cat;
    a.part = reinterpret_cast<float*>(workspace);
    a.P = (long)S * HW; a.S = S; a.B = B; a.HW = HW; a.nch = tok_chunks(HW); a.chunk = dh_cdiv(HW, a.nch);
    if (xprep_recording()) {            // issued with the other levels' calls by dh_xprep_batch_launch_fwd
        if (g_tb.nf && (g_tb.nf == TB_MAXJ || g_tb.dtype_f != dtype || g_tb.L_f != L)) { const int rc = tok_flush_fwd(ST(stream)); if (rc) return rc; }
        g_tb.dtype_f = dtype; g_tb.L_f = L;
        g_tb.f[g_tb.nf++] = a;
        return 0;
    }
    return tok_issue(true, dtype, L, &a, 1, ST(stream));
}

// workspace: dlogits [S*HW][L] floats + partial [nblk][L*32] floats, nblk = ceil(S*HW / 1024)
extern "C" long dh_tokenizer_bwd_workspace_size(int S, int HW, int L) {
    const long P = (long)S * HW;
    return (P * L + (long)dh_cdiv(P, 256) * L * 32) * 4;
}
extern "C" int dh_tokenizer_bwd(int dtype, const void* x, const float* wa, int S, int B, int HW, int L,
                                const float* logits, const float* stats, const float* pooled, const float* dtok_cat,
                                void* dx_accum, float* dwa, float* dpos, int accumulate, void* workspace,
                                void* stream) {
    DH_REQUIRE(L == 4 || L == 8, "tokenizer_bwd: token_len must be 4 or 8, got %d", L);
    TokJob a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wa = wa; a.logits = const_cast<float*>(logits); a.stats = const_cast<float*>(stats); a.pooled = const_cast<float*>(pooled);
    a.dtok_cat = dtok_cat; a.dx = dx_accum; a.dwa = dwa; a.dpos = dpos; a.accumulate = accumulate;
    a.P = (long)S * HW; a.S = S; a.B = B; a.HW = HW;
    a.dlogits = reinterpret_cast<float*>(workspace);
    a.partial = a.dlogits + a.P * L;
    a.nblk = dh_cdiv(a.P, TOK_DWA_CHUNK);
    if (xprep_recording()) {            // issued with the other levels' calls by dh_xprep_batch_launch_bwd
        if (g_tb.nb && (g_tb.nb == TB_MAXJ || g_tb.dtype_b != dtype || g_tb.L_b != L)) { const int rc = tok_flush_bwd(ST(stream)); if (rc) return rc; }
        g_tb.dtype_b = dtype; g_tb.L_b = L;
        g_tb.b[g_tb.nb++] = a;
        return 0;
    }
    return tok_issue(false, dtype, L, &a, 1, ST(stream));
}

// layers > 1: one launch prepares every layer of a decoder stack (same tokens): parameters ln_g / ln_b / wq (forward),
// ln_g / wk / wv / wo and all gradient outputs (backward) are the FIRST layer's pointers, consecutive layers lie
// param_stride floats apart (the net's flat arena); wkT / wvT / woT / wqT are stacked [layers][32*inner]; every saved /
// output tensor is stacked [layers][...] with the single-layer shape.
// ---- batched launches of the matrix-core preparation family (dh_xprep_batch_*) ---------------------------------------------------
// Between dh_xprep_batch_begin() and _end(), dh_xattn_prep_fwd_stack_mfma and dh_xattn_prep_bwd_stack_mfma only RECORD their
// kernels (dim_head 64, up to XB_MAXJ stacks per family).  dh_xprep_batch_launch_fwd(stream) issues the recorded forward
// preparations as one launch (call it BEFORE the decoder launches that read their outputs); dh_xprep_batch_launch_bwd(stream)
// issues, in dependency order, one launch per family: operand gradients -> token-gradient reduce -> weight gradients on the matrix
// cores -> LayerNorm sums (call it AFTER the decoder-stack finalize that writes dkq / dvoT).  Every buffer of a recorded call --
// its workspace included: one per call -- stays alive and unchanged until then.  Per host thread; _abort drops what was recorded.
template <typename A> struct XBatch {
    int n = 0;
    XMulti<A> m;
    bool full() const { return n == XB_MAXJ; }
    void add(const A& a, int gx, int gy, int gz) {
        if (n == 0) m.first[0] = 0;
        m.a[n] = a; m.gx[n] = gx; m.gy[n] = gy;
        m.first[n + 1] = m.first[n] + gx * gy * gz;
        ++n;
    }
};
struct XprepBatch {
    bool on = false, paused = false;       // paused: calls launch at once although a batch is open (dh_xprep_batch_pause)
    XBatch<PrepMArgs> fwd;
    XBatch<PrepBwdMArgs> bwd;
    XBatch<DtokArgs> dtok;
    XBatch<PrepWgArgs> wg_mfma, wg_ln;
};
static thread_local XprepBatch g_xb;
extern "C" int dh_decoder_batch_pending();                // csrc/decoder_fused.hip
extern "C" int dh_decoder_batch_launch(void* stream);
static bool xprep_recording() { return g_xb.on && !g_xb.paused; }
static int xprep_flush_fwd(hipStream_t st) {
    XprepBatch& b = g_xb;
    if (b.fwd.n) {
        b.fwd.m.n = b.fwd.n;
        hipLaunchKernelGGL(xattn_prep_mfma_multi_kernel<64>, dim3(b.fwd.m.first[b.fwd.n]), dim3(256), 0, st, b.fwd.m);
        b.fwd.n = 0;
        DH_CHECK_LAUNCH("xprep_batch_fwd");
    }
    return 0;
}
static int xprep_flush_bwd(hipStream_t st) {
    XprepBatch& b = g_xb;
    if (b.bwd.n) {
        b.bwd.m.n = b.bwd.n;
        hipLaunchKernelGGL(xattn_prep_bwd_mfma_multi_kernel<64>, dim3(b.bwd.m.first[b.bwd.n]), dim3(256), 0, st, b.bwd.m);
        b.bwd.n = 0;
    }
    if (b.dtok.n) {
        b.dtok.m.n = b.dtok.n;
        hipLaunchKernelGGL(dtok_reduce_multi_kernel, dim3(b.dtok.m.first[b.dtok.n]), dim3(4 * 32), 0, st, b.dtok.m);
        b.dtok.n = 0;
    }
    if (b.wg_mfma.n) {
        b.wg_mfma.m.n = b.wg_mfma.n;
        hipLaunchKernelGGL(xattn_prep_wgrad_mfma_multi_kernel, dim3(b.wg_mfma.m.first[b.wg_mfma.n]), dim3(256), 0, st, b.wg_mfma.m);
        b.wg_mfma.n = 0;
    }
    if (b.wg_ln.n) {
        b.wg_ln.m.n = b.wg_ln.n;
        hipLaunchKernelGGL(xattn_prep_wgrad_multi_kernel, dim3(b.wg_ln.m.first[b.wg_ln.n]), dim3(256), 0, st, b.wg_ln.m);
        b.wg_ln.n = 0;
    }
    DH_CHECK_LAUNCH("xprep_batch_bwd");
    return 0;
}
static void xprep_clear() { g_xb.fwd.n = g_xb.bwd.n = g_xb.dtok.n = g_xb.wg_mfma.n = g_xb.wg_ln.n = 0; g_tb.nf = g_tb.nb = 0; }
extern "C" int dh_xprep_batch_begin() { g_xb.on = true; g_xb.paused = false; xprep_clear(); return 0; }
extern "C" int dh_xprep_batch_pause(int paused) { g_xb.paused = paused != 0; return 0; }
extern "C" int dh_xprep_batch_pending() { return g_xb.fwd.n + g_xb.bwd.n + g_xb.dtok.n + g_xb.wg_mfma.n + g_xb.wg_ln.n + g_tb.nf + g_tb.nb; }
extern "C" int dh_xprep_batch_launch_fwd(void* stream) {
    const int rc = tok_flush_fwd(ST(stream));
    return rc ? rc : xprep_flush_fwd(ST(stream));
}
extern "C" int dh_xprep_batch_launch_bwd(void* stream) {
    const int rc = xprep_flush_bwd(ST(stream));
    return rc ? rc : tok_flush_bwd(ST(stream));
}
extern "C" int dh_xprep_batch_end(void* stream) {
    int rc = dh_xprep_batch_launch_fwd(stream);
    if (!rc) rc = dh_xprep_batch_launch_bwd(stream);
    g_xb.on = false;
    return rc;
}
extern "C" int dh_xprep_batch_abort() { g_xb.on = false; xprep_clear(); return 0; }

extern "C" int dh_xattn_prep_fwd_stack(int dtype, const void* tok, long tok_bstride, long tok_sstride, int B, int S, int L,
                                       int heads, int dim_head, int HLP, float scale, float eps, int layers,
                                       long param_stride, const float* ln_g, const float* ln_b, const float* wq,
                                       const void* wkT, const void* wvT, const void* woT, float* mn, float* mstats,
                                       float* k, float* v, void* kq, void* kqT, void* vo, void* voT, void* stream) {
    DH_REQUIRE(heads * L <= HLP && HLP % L == 0 && L <= 8, "xattn_prep: heads*L=%d exceeds HLP=%d (or L > 8)", heads * L, HLP);
    DH_REQUIRE(layers >= 1, "xattn_prep: layers=%d", layers);
    PrepArgs a;
    a.tok = tok; a.tok_bstride = tok_bstride; a.tok_sstride = tok_sstride; a.B = B; a.S = S; a.L = L;
    a.heads = heads; a.dh = dim_head; a.HLP = HLP; a.scale = scale; a.eps = eps; a.ln_g = ln_g; a.ln_b = ln_b;
    a.wq = wq; a.wkT = wkT; a.wvT = wvT; a.woT = woT; a.mn = mn; a.mstats = mstats; a.k = k; a.v = v;
    a.kq = kq; a.kqT = kqT; a.vo = vo; a.voT = voT;
    a.ls_param = param_stride; a.ls_pack = 32L * heads * dim_head;
    const size_t lds = (size_t)(L * 32 + 2 * L * heads * dim_head) * 4;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(xattn_prep_kernel<bf16>, dim3(S, layers), dim3(1024), lds, ST(stream), a);
    else hipLaunchKernelGGL(xattn_prep_kernel<float>, dim3(S, layers), dim3(1024), lds, ST(stream), a);
    DH_CHECK_LAUNCH("xattn_prep_fwd");
    return 0;
}
extern "C" int dh_xattn_prep_fwd(int dtype, const void* tok, long tok_bstride, long tok_sstride, int B, int S, int L,
                                 int heads, int dim_head, int HLP, float scale, float eps, const float* ln_g,
                                 const float* ln_b, const float* wq, const void* wkT, const void* wvT,
                                 const void* woT, float* mn, float* mstats, float* k, float* v, void* kq, void* kqT,
                                 void* vo, void* voT, void* stream) {
    return dh_xattn_prep_fwd_stack(dtype, tok, tok_bstride, tok_sstride, B, S, L, heads, dim_head, HLP, scale, eps, 1, 0,
                                   ln_g, ln_b, wq, wkT, wvT, woT, mn, mstats, k, v, kq, kqT, vo, voT, stream);
}

// C ABI (include/dahitra_hip.h): does the matrix-core preparation serve this shape?
extern "C" int dh_xattn_prep_mfma_supported(int dtype, int S, int L, int heads, int dim_head, int HLP) {
    static const bool off = getenv("DAHITRA_NO_PREP_MFMA") != nullptr;
    return !off && dtype == DH_DTYPE_BF16 && L == 4 && S % 4 == 0 && (dim_head == 32 || dim_head == 64) && heads >= 1 &&
           heads * L <= HLP && HLP == 32;
}
extern "C" int dh_xattn_prep_fwd_stack_mfma(const void* tok, long tok_bstride, long tok_sstride, int B, int S, int heads,
                                            int dim_head, int HLP, float scale, float eps, int layers, long param_stride,
                                            const float* ln_g, const float* ln_b, const float* wk, const float* wv,
                                            const float* wo, const void* wqT, float* mn, float* mstats, float* k, float* v,
                                            void* kq, void* kqT, void* vo, void* voT, void* stream) {
    DH_REQUIRE(dh_xattn_prep_mfma_supported(DH_DTYPE_BF16, S, 4, heads, dim_head, HLP) && layers >= 1,
               "xattn_prep_mfma: unsupported shape S=%d heads=%d dim_head=%d HLP=%d", S, heads, dim_head, HLP);
    PrepMArgs m;
    PrepArgs& a = m.a;
    memset(&m, 0, sizeof(m));
    a.tok = tok; a.tok_bstride = tok_bstride; a.tok_sstride = tok_sstride; a.B = B; a.S = S; a.L = 4;
    a.heads = heads; a.dh = dim_head; a.HLP = HLP; a.scale = scale; a.eps = eps; a.ln_g = ln_g; a.ln_b = ln_b;
    a.mn = mn; a.mstats = mstats; a.k = k; a.v = v; a.kq = kq; a.kqT = kqT; a.vo = vo; a.voT = voT;
    a.ls_param = param_stride; a.ls_pack = 32L * heads * dim_head;
    m.wk = wk; m.wv = wv; m.wo = wo; m.wqT = reinterpret_cast<const bf16*>(wqT);
    if (g_xb.on && !g_xb.paused && dim_head == 64) {                 // recorded: issued with the other stacks' by dh_xprep_batch_launch_fwd
        // (a full family cannot be flushed here: the recorded decoder / finalize launches of the other families depend on an
        // order only dh_xprep_batch_launch_* keeps -- more independent stacks per round than XB_MAXJ must fail, not reorder)
        DH_REQUIRE(!g_xb.fwd.full(), "xattn_prep_fwd_stack: more than %d stacks recorded in one round (dh_xprep_batch_launch_fwd first)", XB_MAXJ);
        g_xb.fwd.add(m, S / 4, layers, 1);
        return 0;
    }
    if (dim_head == 64) hipLaunchKernelGGL(xattn_prep_mfma_kernel<64>, dim3(S / 4, layers), dim3(256), 0, ST(stream), m);
    else hipLaunchKernelGGL(xattn_prep_mfma_kernel<32>, dim3(S / 4, layers), dim3(256), 0, ST(stream), m);
    DH_CHECK_LAUNCH("xattn_prep_fwd_mfma");
    return 0;
}

// workspace: ln_partial [layers][S][2][32] floats (+ dtok_part [layers][S][L][32] floats when layers > 1)
extern "C" long dh_xattn_prep_bwd_stack_workspace_size(int S, int L, int layers) {
    return ((long)layers * S * 64 + 64 + (layers > 1 ? (long)layers * S * L * 32 : 0)) * 4;
}
extern "C" int dh_xattn_prep_bwd_stack(int dtype, const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride,
                                       int B, int S, int L, int heads, int dim_head, int HLP, float scale, int layers,
                                       long param_stride, const float* ln_g, const void* wqT, const float* wk,
                                       const float* wv, const float* wo, const float* mn, const float* mstats,
                                       const float* k, const float* v, const float* dkq, const float* dvoT, float* dk,
                                       float* dv, float* dln_g, float* dln_b, float* dwq, float* dwk, float* dwv,
                                       float* dwo, int accumulate, void* workspace, void* stream) {
    DH_REQUIRE(L <= 8 && layers >= 1, "xattn_prep_bwd: L=%d layers=%d", L, layers);
    PrepBwdArgs a;
    a.tok_bstride = tok_bstride; a.tok_sstride = tok_sstride; a.B = B; a.S = S; a.L = L; a.heads = heads;
    a.dh = dim_head; a.HLP = HLP; a.scale = scale; a.tok = tok; a.dtok = dtok_accum; a.ln_g = ln_g; a.wqT = wqT;
    a.wk = wk; a.wv = wv; a.wo = wo; a.mn = mn; a.mstats = mstats; a.dkq = dkq; a.dvoT = dvoT; a.dk = dk; a.dv = dv;
    a.ln_partial = reinterpret_cast<float*>(workspace);
    a.dtok_part = layers > 1 ? a.ln_partial + (long)layers * S * 64 + 64 : nullptr;
    a.ls_param = param_stride; a.ls_pack = 32L * heads * dim_head;
    const int inner = heads * dim_head;
    const size_t lds = (size_t)(2 * L * inner + L * 32 + 2 * HLP * 32 + 16 * 32 + 32 * L * 32) * 4;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(xattn_prep_bwd_kernel<bf16>, dim3(S, layers), dim3(1024), lds, ST(stream), a);
    else hipLaunchKernelGGL(xattn_prep_bwd_kernel<float>, dim3(S, layers), dim3(1024), lds, ST(stream), a);
    if (layers > 1) {
        DtokArgs dt;
        dt.part = a.dtok_part; dt.layers = layers; dt.S = S; dt.L = L; dt.B = B; dt.bstride = tok_bstride; dt.sstride = tok_sstride;
        dt.dtok = reinterpret_cast<float*>(dtok_accum);
        hipLaunchKernelGGL(dtok_reduce_kernel, dim3(S), dim3(L * 32), 0, ST(stream), dt);
    }
    PrepWgArgs w;
    w.S = S; w.L = L; w.heads = heads; w.dh = dim_head; w.HLP = HLP; w.scale = scale; w.mn = mn; w.k = k; w.v = v;
    w.dk = dk; w.dv = dv; w.dkq = dkq; w.dvoT = dvoT; w.dwq = dwq; w.dwk = dwk; w.dwv = dwv; w.dwo = dwo;
    w.ln_partial = a.ln_partial; w.dln_g = dln_g; w.dln_b = dln_b;
    w.accumulate = accumulate; w.ls_param = param_stride; w.ln_only = 0;
    DH_REQUIRE(dim_head % 8 == 0, "xattn_prep_bwd: dim_head=%d must be a multiple of 8", dim_head);
    hipLaunchKernelGGL(xattn_prep_wgrad_kernel, dim3(inner / 8, 5, layers), dim3(256), 0, ST(stream), w);
    DH_CHECK_LAUNCH("xattn_prep_bwd");
    return 0;
}
// the matrix-core form of dh_xattn_prep_bwd_stack (dh_xattn_prep_mfma_supported shapes): wq is the first layer's fp32 master,
// woT ([inner][32]), wkT, wvT ([32][inner]) the stacked bf16 transposes; everything else as in dh_xattn_prep_bwd_stack
extern "C" int dh_xattn_prep_bwd_stack_mfma(const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride, int B, int S,
                                            int heads, int dim_head, int HLP, float scale, int layers, long param_stride,
                                            const float* ln_g, const float* wq, const void* woT, const void* wkT,
                                            const void* wvT, const float* mn, const float* mstats, const float* k, const float* v,
                                            const float* dkq, const float* dvoT, float* dk, float* dv, float* dln_g,
                                            float* dln_b, float* dwq, float* dwk, float* dwv, float* dwo, int accumulate,
                                            void* workspace, void* stream) {
    const int L = 4;
    DH_REQUIRE(dh_xattn_prep_mfma_supported(DH_DTYPE_BF16, S, L, heads, dim_head, HLP) && layers >= 1,
               "xattn_prep_bwd_mfma: unsupported shape S=%d heads=%d dim_head=%d HLP=%d", S, heads, dim_head, HLP);
    PrepBwdMArgs m;
    memset(&m, 0, sizeof(m));
    PrepBwdArgs& a = m.a;
    a.tok_bstride = tok_bstride; a.tok_sstride = tok_sstride; a.B = B; a.S = S; a.L = L; a.heads = heads;
    a.dh = dim_head; a.HLP = HLP; a.scale = scale; a.tok = tok; a.dtok = dtok_accum; a.ln_g = ln_g;
    a.mn = mn; a.mstats = mstats; a.dkq = dkq; a.dvoT = dvoT; a.dk = dk; a.dv = dv;
    a.ln_partial = reinterpret_cast<float*>(workspace);
    a.dtok_part = layers > 1 ? a.ln_partial + (long)layers * S * 64 + 64 : nullptr;
    a.ls_param = param_stride; a.ls_pack = 32L * heads * dim_head;
    m.wq = wq; m.woT = reinterpret_cast<const bf16*>(woT); m.wkT = reinterpret_cast<const bf16*>(wkT);
    m.wvT = reinterpret_cast<const bf16*>(wvT);
    const int inner = heads * dim_head;
    DtokArgs dt;
    dt.part = a.dtok_part; dt.layers = layers; dt.S = S; dt.L = L; dt.B = B; dt.bstride = tok_bstride; dt.sstride = tok_sstride;
    dt.dtok = reinterpret_cast<float*>(dtok_accum);
    static const bool wg_fma_ = getenv("DAHITRA_PREP_WGRAD_FMA") != nullptr;
    if (g_xb.on && !g_xb.paused && dim_head == 64 && S % 8 == 0 && !wg_fma_) {
        // recorded: the four kernels of this stack join the other stacks' in dh_xprep_batch_launch_bwd's four launches
        // (see the forward: flushing here would run before the recorded decoder finalize that produces dkq / dvoT)
        DH_REQUIRE(!(g_xb.bwd.full() || g_xb.dtok.full() || g_xb.wg_mfma.full() || g_xb.wg_ln.full()),
                   "xattn_prep_bwd_stack: more than %d stacks recorded in one round (dh_xprep_batch_launch_bwd first)", XB_MAXJ);
        g_xb.bwd.add(m, S / 4, layers, 1);
        if (layers > 1) g_xb.dtok.add(dt, S, 1, 1);
        PrepWgArgs w;
        w.S = S; w.L = L; w.heads = heads; w.dh = dim_head; w.HLP = HLP; w.scale = scale; w.mn = mn; w.k = k; w.v = v;
        w.dk = dk; w.dv = dv; w.dkq = dkq; w.dvoT = dvoT; w.dwq = dwq; w.dwk = dwk; w.dwv = dwv; w.dwo = dwo;
        w.ln_partial = a.ln_partial; w.dln_g = dln_g; w.dln_b = dln_b;
        w.accumulate = accumulate; w.ls_param = param_stride; w.ln_only = 0;
        g_xb.wg_mfma.add(w, heads, 4, layers);
        w.ln_only = 1;
        g_xb.wg_ln.add(w, 2, 1, layers);
        return 0;
    }
    // launched at once: a recorded stack finalize (the decoder batch holds it back) produces the dkq / dvoT read here
    if (dh_decoder_batch_pending()) { const int rc = dh_decoder_batch_launch(stream); if (rc) return rc; }
    if (dim_head == 64) hipLaunchKernelGGL(xattn_prep_bwd_mfma_kernel<64>, dim3(S / 4, layers), dim3(256), 0, ST(stream), m);
    else hipLaunchKernelGGL(xattn_prep_bwd_mfma_kernel<32>, dim3(S / 4, layers), dim3(256), 0, ST(stream), m);
    if (layers > 1) hipLaunchKernelGGL(dtok_reduce_kernel, dim3(S), dim3(L * 32), 0, ST(stream), dt);
    PrepWgArgs w;
    w.S = S; w.L = L; w.heads = heads; w.dh = dim_head; w.HLP = HLP; w.scale = scale; w.mn = mn; w.k = k; w.v = v;
    w.dk = dk; w.dv = dv; w.dkq = dkq; w.dvoT = dvoT; w.dwq = dwq; w.dwk = dwk; w.dwv = dwv; w.dwo = dwo;
    w.ln_partial = a.ln_partial; w.dln_g = dln_g; w.dln_b = dln_b;
    w.accumulate = accumulate; w.ls_param = param_stride; w.ln_only = 0;
    static const bool wg_fma = getenv("DAHITRA_PREP_WGRAD_FMA") != nullptr;
    if (dim_head == 64 && S % 8 == 0 && !wg_fma) {
        hipLaunchKernelGGL(xattn_prep_wgrad_mfma_kernel, dim3(heads, 4, layers), dim3(256), 0, ST(stream), w);
        w.ln_only = 1;
        hipLaunchKernelGGL(xattn_prep_wgrad_kernel, dim3(2, 1, layers), dim3(256), 0, ST(stream), w);      // LayerNorm dgamma / dbeta only
    } else {
        hipLaunchKernelGGL(xattn_prep_wgrad_kernel, dim3(inner / 8, 5, layers), dim3(256), 0, ST(stream), w);
    }
    DH_CHECK_LAUNCH("xattn_prep_bwd_mfma");
    return 0;
}
extern "C" int dh_xattn_prep_bwd(int dtype, const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride,
                                 int B, int S, int L, int heads, int dim_head, int HLP, float scale,
                                 const float* ln_g, const void* wqT, const float* wk, const float* wv,
                                 const float* wo, const float* mn, const float* mstats, const float* k,
                                 const float* v, const float* dkq, const float* dvoT, float* dk, float* dv,
                                 float* dln_g, float* dln_b, float* dwq, float* dwk, float* dwv, float* dwo,
                                 int accumulate, void* workspace, void* stream) {
    return dh_xattn_prep_bwd_stack(dtype, tok, dtok_accum, tok_bstride, tok_sstride, B, S, L, heads, dim_head, HLP, scale, 1,
                                   0, ln_g, wqT, wk, wv, wo, mn, mstats, k, v, dkq, dvoT, dk, dv, dln_g, dln_b, dwq, dwk,
                                   dwv, dwo, accumulate, workspace, stream);
}
extern "C" long dh_xattn_prep_bwd_workspace_size(int S) { return ((long)S * 64 + 64) * 4; }

extern "C" int dh_softmax_groups_fwd(int dtype, const void* x, void* y, long rows, int heads, int L, int HLP,
                                     void* stream) {
    DH_REQUIRE(L <= 16 && HLP % L == 0, "softmax_groups: L=%d HLP=%d", L, HLP);
    const long n = rows * (HLP / L);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(softmax_groups_fwd_kernel<bf16>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), (const bf16*)x, (bf16*)y, rows, heads, L, HLP);
    else hipLaunchKernelGGL(softmax_groups_fwd_kernel<float>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), (const float*)x, (float*)y, rows, heads, L, HLP);
    DH_CHECK_LAUNCH("softmax_groups_fwd");
    return 0;
}
extern "C" int dh_softmax_groups_bwd(int dtype, const void* y, const void* dy, void* dx, long rows, int heads, int L,
                                     int HLP, void* stream) {
    DH_REQUIRE(L <= 16 && HLP % L == 0, "softmax_groups: L=%d HLP=%d", L, HLP);
    const long n = rows * (HLP / L);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(softmax_groups_bwd_kernel<bf16>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), (const bf16*)y, (const bf16*)dy, (bf16*)dx, rows, heads, L, HLP);
    else hipLaunchKernelGGL(softmax_groups_bwd_kernel<float>, dim3(dh_cdiv(n, 256)), dim3(256), 0, ST(stream), (const float*)y, (const float*)dy, (float*)dx, rows, heads, L, HLP);
    DH_CHECK_LAUNCH("softmax_groups_bwd");
    return 0;
}

extern "C" int dh_self_attn_fwd(int dtype, const void* qkv, void* o, float* attn, int B, int n, int heads,
                                int dim_head, float scale, void* stream) {
    DH_REQUIRE(n <= 16, "self_attn: at most 16 tokens, got %d", n);
    const size_t lds = (size_t)(3 * n * dim_head + n * n) * 4;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(self_attn_fwd_kernel<bf16>, dim3(B * heads), dim3(64), lds, ST(stream), (const bf16*)qkv, (bf16*)o, attn, n, heads, dim_head, scale);
    else hipLaunchKernelGGL(self_attn_fwd_kernel<float>, dim3(B * heads), dim3(64), lds, ST(stream), (const float*)qkv, (float*)o, attn, n, heads, dim_head, scale);
    DH_CHECK_LAUNCH("self_attn_fwd");
    return 0;
}
extern "C" int dh_self_attn_bwd(int dtype, const void* qkv, const float* attn, const void* dout, void* dqkv, int B,
                                int n, int heads, int dim_head, float scale, void* stream) {
    DH_REQUIRE(n <= 16, "self_attn: at most 16 tokens, got %d", n);
    const size_t lds = (size_t)(4 * n * dim_head + 2 * n * n) * 4;
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(self_attn_bwd_kernel<bf16>, dim3(B * heads), dim3(64), lds, ST(stream), (const bf16*)qkv, attn, (const bf16*)dout, (bf16*)dqkv, n, heads, dim_head, scale);
    else hipLaunchKernelGGL(self_attn_bwd_kernel<float>, dim3(B * heads), dim3(64), lds, ST(stream), (const float*)qkv, attn, (const float*)dout, (float*)dqkv, n, heads, dim_head, scale);
    DH_CHECK_LAUNCH("self_attn_bwd");
    return 0;
}

#ifdef XP_TIMING
extern "C" int dh_debug_xpt(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xpt), (size_t)n * 8); }
#endif
