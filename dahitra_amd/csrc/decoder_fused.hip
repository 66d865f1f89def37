// Fused cross-attention decoder layer (bf16, token_len 4, heads*4 <= 32) for gfx950:
//
//   x1 = x + Wo . softmax_l( scale * <Wq LN(x), k_l> ) v_l + bo        (help_funcs.py:66-114, Residual2 / PreNorm2)
//   y  = x1 + W2 gelu(W1 LN(x1) + b1) + b2                             (help_funcs.py:52-63, Residual / PreNorm)
//
// in ONE kernel per direction instead of 7 (forward) / ~25 (backward) launches.  The attention uses the
// per-image re-associated operands Kq = scale*Wq^T k, Vo = Wo v of tokens.hip, so every product is a
// 32-wide MFMA (v_mfma_f32_16x16x32_bf16, fp32 accumulate):
//   dots[hl] = Kq . LN(x)      softmax over the 4 keys of a head = the 4 accumulator registers of a lane
//   o[c]     = VoT . attn      z[m] = W1 . LN(x1)      out[c] = W2 . gelu(z)
// Register layout ("D layout"): lane (pl = lane&15, g = lane>>4) of a 16-pixel sub-tile owns pixel pl and
// channels s*16 + g*4 + j (s = 16-channel half, j = 0..3) -- exactly what an MFMA leaves in its accumulator
// when the weights are the A operand.  The same registers, packed to bf16, ARE the next MFMA's B operand if
// the weight rows are read with the matching k-permutation kappa(g, e) = e < 4 ? g*4+e : 16+g*4+(e-4), so
// activations never round-trip through LDS; LayerNorm reduces over the 4 lane groups with two shuffles.
//
// The backward kernel recomputes the forward from x (nothing but x is saved), chains the data gradients the
// same way, and forms the pixel-reduction gradients (dW1, dW2, per-image dKq, dVoT) with K = 32 pixels per
// MFMA through wave-private LDS tiles read back with ds_read_b64_tr_b16; biases / LayerNorm gradients are
// lane-local sums.  Per-workgroup partials are combined deterministically by dec_bwd_finalize_kernel.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <unordered_map>

namespace {

constexpr int D = 32;
constexpr int DEC_MAXDEPTH = 8;      // layers of a stack launch (the backward keeps every layer's parameter vectors in LDS)
// Weight rows sit in LDS in FRAGMENT ORDER: the eight elements kappa(g, 0..7) a lane multiplies are contiguous, so an A
// fragment is ONE ds_read_b128.  (The first form read them as two 8-byte halves 32 bytes apart; hipcc merged each pair into a
// ds_read2_b64, which is banked modulo 32 dwords over 16 CONTIGUOUS lanes and runs at half the ds_read_b128 rate: at the
// 80-byte pitch chosen for ds_read_b64, rows pl and pl + 8 met on one bank -- SQ_LDS_BANK_CONFLICT was 50 % of the LDS cycles
// of both kernels, profiles/r03q_pmc_mfma_util.json.)  ds_read_b128 is served in four non-contiguous 16-lane groups
// (MI355X_MICROARCH.md, LDS); enumerating them, rows of 32 elements are conflict-free at a pitch of 96 bytes, rows of 64 at 160.
constexpr int WP = 48;           // LDS pitch (elements) of 32-wide weight rows
constexpr int wide_pitch(int cols) { return cols == 32 ? 48 : 80; }      // ... of [32][MLP] rows

struct DecArgs {
    const bf16* x;
    bf16* y;                     // fwd: output; bwd: dx
    const bf16* dy;              // bwd: gradient of the layer output
    const bf16 *kq, *voT, *vo, *kqT;          // per image [S][32][32] bf16 (vo, kqT: backward only)
    const bf16 *w1, *w2, *w1T, *w2T;          // [MLP][32], [32][MLP], [32][MLP], [MLP][32]
    const float *g1, *be1, *bo, *g2, *be2, *fb1, *fb2;
    float* partial;              // bwd: [nblk][PSZ]
    // A workgroup takes `upb` consecutive 64-row UNITS of one image (one 16-pixel sub-tile per wave and unit); an image is cut
    // into bpi = ceil(units per image / upb) blocks, the last of them shorter: any upb serves, not only the powers of two that
    // divide an image, so the host can size the blocks of every job of a launch for the launch as a whole (dec_balance)
    int rows_per_image, upb, bpi;
    long rows;
    float eps;
    // A whole decoder STACK in one launch (depth > 1; dh_decoder_stack_fwd / _bwd).  A pixel row attends to the tokens of its
    // image only and the tokens do not change from layer to layer (help_funcs.py:170-186: x = attn(x, m); x = ff(x) with m
    // fixed), so the stack is a per-row function: a workgroup takes ITS rows through all layers -- re-staging the layer's
    // weights, nothing else synchronises -- instead of `depth` launches that each fill and drain the chip.  Layer l reads
    // (l == 0 ? x : ys + (l - 1) act_ls) and writes ys + l act_ls (every layer's output is kept: the backward recomputes from
    // it); its operands lie at constant strides: kq_ls elements between the per-image attention operands of consecutive
    // layers, w_ls between the packed MLP weights, par_ls floats between the fp32 parameter vectors, part_ls floats between
    // the backward's partial blocks.  Backward: layer l reads its input as above and the gradient from (l == depth - 1 ? dy :
    // dwork), and writes (l == 0 ? y (= dx) : dwork) -- in place: a lane reads and writes only its own rows.
    int depth;
    bf16* ys;
    bf16* dwork;
    long act_ls, kq_ls, w_ls, par_ls, part_ls;
};

union U8 {
    uint4 u;
    uint2 h[2];
    s16x8 v;
};

__device__ __forceinline__ s16x8 pack8(const float (&a)[4], const float (&b)[4]) {
    U8 r;
    r.u.x = f2bf2(a[0], a[1]);
    r.u.y = f2bf2(a[2], a[3]);
    r.u.z = f2bf2(b[0], b[1]);
    r.u.w = f2bf2(b[2], b[3]);
    return r.v;
}
// A fragment of weight row `row`, logical k = koff + kappa(g, e): one 16-byte LDS read (rows staged by stage_ld / stage_st in fragment order)
__device__ __forceinline__ s16x8 lds_a(const unsigned short* base, int pitch, int row, int koff, int g) {
    U8 r;
    r.u = *reinterpret_cast<const uint4*>(base + row * pitch + koff + g * 8);
    return r.v;
}
__device__ __forceinline__ f32x4 mma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// over the 4 lane groups (rows of 16 lanes) holding one pixel's channels.  v_permlane16_swap / v_permlane32_swap (gfx950)
// exchange rows between two registers in the VALU: with both operands = v, the two results are (this row pair's even row,
// its odd row) resp. (lower half, upper half) in every lane -- the xor-16 / xor-32 butterfly without the two ds_bpermute
// round trips through the LDS unit (16 of them sat in the backward chain of every sub-tile).  Same sums, same bits.
__device__ __forceinline__ float group4_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// a wave-uniform pointer, forced into scalar registers
template <typename T> __device__ __forceinline__ const T* uniform_ptr(const T* q) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const T*>(((unsigned long long)hi << 32) | lo);
}

// 8 bytes at (uniform base, 32-bit lane offset), as GLOBAL accesses: a pointer rebuilt from scalar halves is generic to the
// compiler, and a flat load counts on lgkmcnt as well -- every LDS wait of the chain would wait for x / dy of the next sub-tile
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) u32x2 gu32x2;
__device__ __forceinline__ uint2 gld8(const char* base, unsigned off) {
    const u32x2 v = *(const gu32x2*)(base + off);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void gst8(char* base, unsigned off, uint2 v) { *(gu32x2*)(base + off) = u32x2{v.x, v.y}; }

// workgroup barrier that orders LDS traffic only: __syncthreads() is also a fence for global memory, i.e. s_waitcnt vmcnt(0) --
// it would drain the next layer's operands requested just before it (prefetch of dec_fwd_body / dec_bwd_body)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void lds4(const float* p, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}

// cooperative copy of a [ROWS][COLS] bf16 matrix (global, dense) into LDS with pitch, every 32-column block in fragment order:
// logical columns 4q .. 4q+3 (q = 0..7) land at position 8 (q & 3) + 4 (q >> 2), i.e. kappa(g, e) at g * 8 + e.
// 32-column rows: a ds_write_b64 lane group (16 lanes, banks modulo 32 dwords) covers two rows of 16 dwords, and rows r, r + 1
// lie 24 dwords apart (pitch 48 elements): half of their banks coincide.  Rows r, r + 2 (48 dwords apart) do not: the groups
// take rows in the order 0 2 1 3 (bits 0 and 1 of the row index swapped; ROWS is a multiple of 4).
// Two phases with compile-time shapes: every matrix of a layer is REQUESTED (stage_ld, 8-byte pieces into
// registers) before the first is committed (stage_st).  As one run-time-shaped loop per matrix it was load -> wait -> store per
// matrix: seven L2 round trips in a row at the head of every layer of the backward (3.1 k of the ~9 k cycles a workgroup spends
// per layer outside its sub-tile loop, tools/dec_timeline.py), four in the forward.
template <int ROWS, int COLS> struct StageRegs { uint2 v[(ROWS * COLS / 4 + 255) / 256]; };
// (the matrix base is wave-uniform: as scalar base + 32-bit lane offset a load needs ONE address register, and the matrices of one
// shape share it -- as 64-bit lane addresses the seven matrices of the backward held 14 registers across the layer loop)
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_ld(StageRegs<ROWS, COLS>& s, const bf16* src, int tid) {
    constexpr int VEC = COLS / 4, N = (ROWS * VEC + 255) / 256;
    const char* base = reinterpret_cast<const char*>(uniform_ptr(src));
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = tid + k * 256;
        if (ROWS * VEC % 256 == 0 || i < ROWS * VEC) {
            const int r0 = i / VEC, q = i % VEC;
            const int r = VEC == 8 ? ((r0 & ~3) | ((r0 & 1) << 1) | ((r0 >> 1) & 1)) : r0;
            s.v[k] = gld8(base, (unsigned)((r * COLS + q * 4) * 2));
        }
    }
}
template <int ROWS, int COLS>
__device__ __forceinline__ void stage_st(unsigned short* dst, int pitch, const StageRegs<ROWS, COLS>& s, int tid) {
    constexpr int VEC = COLS / 4, N = (ROWS * VEC + 255) / 256;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int i = tid + k * 256;
        if (ROWS * VEC % 256 == 0 || i < ROWS * VEC) {
            const int r0 = i / VEC, q = i % VEC, c = (q & ~7) * 4 + (q & 3) * 8 + ((q >> 2) & 1) * 4;
            const int r = VEC == 8 ? ((r0 & ~3) | ((r0 & 1) << 1) | ((r0 >> 1) & 1)) : r0;
            *reinterpret_cast<uint2*>(dst + r * pitch + c) = s.v[k];
        }
    }
}

// GELU (erf form, help_funcs.py:57 nn.GELU) for the bf16 path: gelu(z) = z Phi(z) with Phi approximated by a logistic function
// of an odd quintic,  Phi(z) ~ sigma(z (k0 + k1 t + k2 t^2)),  t = min(z^2, 36)  (the three coefficients fitted to the erf form
// over |z| <= 12, tools/gelu_fit.py: max |error| 2.7e-5 in gelu, 1.1e-4 in gelu' -- the bf16 result is rounded at 4e-3
// relative; with the textbook tanh constants the same form is 4.7e-4 / 8.7e-4 off).  7 VALU + exp2 + rcp per value, 12 with
// the derivative; the Abramowitz-Stegun erf of the previous build (error 1.5e-7) took 14 + 2 resp. 18 + 2 and was 40 % of the
// forward kernel's VALU work (tools/dec_timeline.py).  The fp32 parity path keeps erff (common.h); the clamp keeps
// the quintic monotone (k2 < 0) where sigma is already 0 / 1 in fp32.
__device__ __forceinline__ float gelu_fast(float z, float* dgelu) {
    constexpr float K0 = 1.5950013121464464f, K1 = 0.07399967641212012f, K2 = -0.0007003036283985342f;
    constexpr float NL2E = -1.4426950408889634f;                       // exp(-a) = exp2(a * NL2E)
    const float t = fminf(z * z, 36.0f);
    const float w = z * ((K0 * NL2E) + t * ((K1 * NL2E) + t * (K2 * NL2E)));
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));      // sigma(a); exp2 -> inf gives rcp -> 0
    if (dgelu) {
        const float ap = K0 + t * (3.0f * K1 + t * (5.0f * K2));       // da/dz (where t is clamped, s (1 - s) is 0 anyway)
        *dgelu = s + (z * ap) * (s - s * s);
    }
    return z * s;
}

struct LNres {
    float mean, rstd;
};
// LayerNorm of one pixel held as v[2][4] across 4 lane groups (gam / bet: this lane's 8 channels); returns xhat in xh and the affine in o
__device__ __forceinline__ LNres layer_norm(const float (&v)[2][4], const float (&gam)[2][4], const float (&bet)[2][4],
                                            float eps, float (&xh)[2][4], float (&o)[2][4]) {
    // sum and sum of squares in ONE pass: the two cross-row reductions run side by side instead of one after the other (these
    // kernels are chains of dependent steps), and x_hat = x * rstd - mean * rstd is one FMA per channel
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s += v[h][j]; q += v[h][j] * v[h][j]; }
    const float mean = group4_sum(s) * (1.f / D);
    const float var = fmaxf(group4_sum(q) * (1.f / D) - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps), nmr = -mean * rstd;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xh[h][j] = v[h][j] * rstd + nmr;
            o[h][j] = xh[h][j] * gam[h][j] + bet[h][j];
        }
    return LNres{mean, rstd};
}

// ------------------------------------------------------------------------------------------------------
// forward: 256 threads, rows_per_block pixel rows per workgroup (16-pixel sub-tiles, wave after wave)
// ------------------------------------------------------------------------------------------------------
// STACK = false: one layer, the loop below folds away (the layer loop costs registers: 98 -> 134 in the forward, and the
// backward, which sits at its 256-register budget, spills -- 35.0 -> 39.5 us on a single large layer)
template <int MLP, bool STACK>
__device__ __forceinline__ void dec_fwd_body(const DecArgs& p, const int bid) {
    constexpr int W2P = wide_pitch(MLP);
    constexpr int PARW = 6 * 32 + MLP;                   // g1 be1 bo g2 be2 fb2 (32 each), fb1 (MLP): one layer's parameter vectors
    __shared__ __attribute__((aligned(16))) unsigned short sKq[32 * WP], sVoT[32 * WP], sW1[MLP * WP], sW2[32 * W2P];
    __shared__ __attribute__((aligned(16))) float sParAll[(STACK ? DEC_MAXDEPTH : 1) * PARW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    const int img = bid / p.bpi, u0 = (bid - img * p.bpi) * p.upb;
    const long row0 = (long)img * p.rows_per_image + (long)u0 * 64;
    const int nsub = min(p.upb, (p.rows_per_image >> 6) - u0);          // 16-pixel sub-tiles per wave
    const int depth = STACK ? p.depth : 1;
    // this lane's 8 bytes of a row, relative to the first row of the workgroup's sub-tile round (the rounds lie 4096 bytes apart)
    const unsigned loff = (unsigned)(((wv * 16 + pl) * D + g * 4) * 2);
    // What a layer needs from memory before its first sub-tile -- the four matrices and x of sub-tile 0 -- is REQUESTED before the
    // LAST sub-tile of the layer before it and committed to LDS after that layer's last barrier (as in the backward).  Requested at
    // the head of the layer, a workgroup waited per layer for an L2 round trip (matrices), then a barrier, then an HBM round trip
    // (x): ~3 us next to 16 us of work in the 512-row blocks of DAHiTra's 64 x 64 level.
    struct {
        StageRegs<32, D> rKq, rVoT;
        StageRegs<MLP, D> rW1;
        StageRegs<D, MLP> rW2;
    } pre;
    uint2 nx[2];
    auto layer_in = [&](int l) { return reinterpret_cast<const char*>(uniform_ptr(((!STACK || l == 0) ? p.x : p.ys + (l - 1) * p.act_ls) + row0 * D)); };
    auto prefetch_w = [&](int l) {
        int ts = tid;
        if constexpr (STACK) asm volatile("" : "+v"(ts));
        const long ko = STACK ? l * p.kq_ls : 0, wo = STACK ? l * p.w_ls : 0;
        stage_ld(pre.rKq, p.kq + ko + (size_t)img * 32 * D, ts);
        stage_ld(pre.rVoT, p.voT + ko + (size_t)img * D * 32, ts);
        stage_ld(pre.rW1, p.w1 + wo, ts);
        stage_ld(pre.rW2, p.w2 + wo, ts);
    };
    auto prefetch_x = [&](int l) {
        const char* xb = layer_in(l);
        nx[0] = gld8(xb, loff);
        nx[1] = gld8(xb, loff + 32);
    };
    prefetch_x(0);
    prefetch_w(0);
    for (int i = tid; i < depth * PARW; i += 256) {
        const int l = i / PARW, k = i - l * PARW, c = k & 31;
        const float* src = k < 32 ? p.g1 : k < 64 ? p.be1 : k < 96 ? p.bo : k < 128 ? p.g2 : k < 160 ? p.be2 : k < 192 ? p.fb2 : p.fb1;
        sParAll[i] = src[(STACK ? l * p.par_ls : 0) + (k < 192 ? c : k - 192)];
    }
#pragma unroll 1
    for (int l = 0; l < depth; ++l) {
        if (l) lds_barrier();                        // every wave is done with the previous layer's weights
        const char* xin = layer_in(l);
        char* yout = const_cast<char*>(reinterpret_cast<const char*>(uniform_ptr((STACK ? p.ys + l * p.act_ls : p.y) + row0 * D)));
        stage_st(sKq, WP, pre.rKq, tid);
        stage_st(sVoT, WP, pre.rVoT, tid);
        stage_st(sW1, WP, pre.rW1, tid);
        stage_st(sW2, W2P, pre.rW2, tid);
        __syncthreads();                             // (the first layer's parameter table included)
        // the seven parameter vectors: this lane's 8 (fb1: MLP / 4) channels, in registers for the whole layer
        float cg1[2][4], cbe1[2][4], cbo[2][4], cg2[2][4], cbe2[2][4], cfb2[2][4], cfb1[MLP / 16][4];
        const float* par = sParAll + l * PARW;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = h * 16 + g * 4;
            lds4(par + c, cg1[h]); lds4(par + 32 + c, cbe1[h]); lds4(par + 64 + c, cbo[h]);
            lds4(par + 96 + c, cg2[h]); lds4(par + 128 + c, cbe2[h]); lds4(par + 160 + c, cfb2[h]);
        }
#pragma unroll
        for (int h = 0; h < MLP / 16; ++h) lds4(par + 192 + h * 16 + g * 4, cfb1[h]);
        // x of the NEXT sub-tile is requested while this one is computed (raw bf16: 4 registers), as in the backward: a load at the
        // head of the chain was waited for at memory latency in every round
        for (int ps = 0; ps < nsub; ++ps) {
            float x[2][4], xh[2][4], xn[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                x[h][0] = __uint_as_float(nx[h].x << 16); x[h][1] = __uint_as_float(nx[h].x & 0xffff0000u);
                x[h][2] = __uint_as_float(nx[h].y << 16); x[h][3] = __uint_as_float(nx[h].y & 0xffff0000u);
            }
            if (ps + 1 < nsub) {
                nx[0] = gld8(xin + (size_t)(ps + 1) * 4096, loff);
                nx[1] = gld8(xin + (size_t)(ps + 1) * 4096, loff + 32);
            } else if (STACK && l + 1 < depth) {
                prefetch_w(l + 1);                   // in flight during the layer's last sub-tile
                if (nsub > 1) prefetch_x(l + 1);     // (sub-tile 0 of this layer's output: stored by this lane rounds ago)
            }
            layer_norm(x, cg1, cbe1, p.eps, xh, xn);
            // dots -> softmax over the 4 keys of each head (lane-local)
            const s16x8 bxn = pack8(xn[0], xn[1]);
            float at[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 d = mma(lds_a(sKq, WP, s * 16 + pl, 0, g), bxn, f32x4{0.f, 0.f, 0.f, 0.f});
                const float m = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]));
                float e[4], sum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) { e[j] = __expf(d[j] - m); sum += e[j]; }
                const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int j = 0; j < 4; ++j) at[s][j] = e[j] * inv;
            }
            const s16x8 bat = pack8(at[0], at[1]);
            float x1[2][4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4 o = mma(lds_a(sVoT, WP, s * 16 + pl, 0, g), bat, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int j = 0; j < 4; ++j) x1[s][j] = o[j] + cbo[s][j] + x[s][j];
            }
            float xh2[2][4], l2[2][4];
            layer_norm(x1, cg2, cbe2, p.eps, xh2, l2);
            const s16x8 bl2 = pack8(l2[0], l2[1]);
            float hh[MLP / 16][4];
#pragma unroll
            for (int s = 0; s < MLP / 16; ++s) {
                f32x4 z = mma(lds_a(sW1, WP, s * 16 + pl, 0, g), bl2, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int j = 0; j < 4; ++j) hh[s][j] = gelu_fast(z[j] + cfb1[s][j], nullptr);
            }
            f32x4 out[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int q = 0; q < MLP / 32; ++q) {
                const s16x8 bh = pack8(hh[2 * q], hh[2 * q + 1]);
#pragma unroll
                for (int s = 0; s < 2; ++s) out[s] = mma(lds_a(sW2, W2P, s * 16 + pl, 32 * q, g), bh, out[s]);
            }
            char* yr = yout + (size_t)ps * 4096;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) r[j] = out[s][j] + cfb2[s][j] + x1[s][j];
                gst8(yr, loff + s * 32, make_uint2(f2bf2(r[0], r[1]), f2bf2(r[2], r[3])));
            }
        }
        if (STACK && l + 1 < depth && nsub == 1) prefetch_x(l + 1);      // (one sub-tile per wave: its output was stored just now)
    }
}
template <int MLP, bool STACK = false>
__global__ __launch_bounds__(256) void dec_fwd_kernel(DecArgs p) { dec_fwd_body<MLP, STACK>(p, blockIdx.x); }
// Layers of several decoder stacks in one launch (dh_decoder_batch_*): DAHiTra's levels are independent, the launches of the small
// ones (16 x 16 and 32 x 32 maps: 128 - 512 workgroups, 3.5 - 16 us) ride with the large one's.  Arguments by value.
constexpr int DEC_MAXJ = 4;
struct DecMulti {
    int n;
    int first[DEC_MAXJ + 1];
    DecArgs a[DEC_MAXJ];
};
template <int MLP, bool STACK = false>
__global__ __launch_bounds__(256) void dec_fwd_multi_kernel(DecMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const DecArgs p = m.a[j];
    dec_fwd_body<MLP, STACK>(p, (int)blockIdx.x - m.first[j]);
}

// ------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------
// partial layout per workgroup (floats)
template <int MLP> struct PL {
    static constexpr int W1 = 0, W2 = MLP * D, B1 = 2 * MLP * D, B2 = B1 + MLP, BO = B2 + D, G1 = BO + D, BE1 = G1 + D,
                         G2 = BE1 + D, BE2 = G2 + D, KQ = BE2 + D, VOT = KQ + 32 * D, SIZE = VOT + D * 32;
    // The four matrix regions hold their 16 x 16 accumulator blocks as the backward's lanes own them (dec_bwd_body): float n of
    // a region = accumulator n >> 8, lane (n >> 2) & 63 (pixel-column pl = lane & 15, row group g = lane >> 4), register j = n & 3,
    // i.e. element (row block * 16 + g * 4 + j, column block * 16 + pl).  Accumulator order: W1 [MLP][D] and KQ / VOT [32][32]
    // (row block, column block) row-major over 2 column blocks; W2 [D][MLP] over MLP / 16 column blocks.
    // Returns the row-major index of float n of a region whose matrix has `cols` columns.
    __host__ __device__ static int matrix_index(int n, int cols) {
        const int acc = n >> 8, ln = (n >> 2) & 63, j = n & 3, nb = cols / 16;
        return ((acc / nb) * 16 + (ln >> 4) * 4 + j) * cols + (acc % nb) * 16 + (ln & 15);
    }
};
constexpr int lds_pitch(int row_bytes) { return ((row_bytes / 32) & 1) ? row_bytes : row_bytes + 32; }

// write a packed (kappa-ordered) 32-channel operand of pixel `pp` into a tile at channel offset c0.
// The 8-byte units of a pixel row (4 channels each) are XOR-swizzled by (pixel >> 2) & 3 in their low two bits: the tile
// pitches (96 / 160 bytes: odd multiples of 32, what ds_read_b64_tr_b16 wants) put the 16 pixel rows a ds_write_b64 lane group
// writes (same g, pl = 0..15; banked modulo 32 dwords) on 4 bank groups, a 4-way conflict on every tile write -- 16 of them per
// sub-tile; with the swizzle the four rows of one bank group take four different units: conflict-free.  The transpose reads
// undo it in their address (a lane reads ONE unit of ONE row) and keep their own conflict-free pattern (a row's four units are
// only permuted among the four lanes that read them).
__device__ __forceinline__ void tile_put(unsigned char* tile, int pitch, int pp, int c0, int g, s16x8 v) {
    U8 r;
    r.v = v;
    unsigned char* dst = tile + pp * pitch + (c0 + (g ^ ((pp >> 2) & 3)) * 4) * 2;
    *reinterpret_cast<uint2*>(dst) = r.h[0];
    *reinterpret_cast<uint2*>(dst + 32) = r.h[1];
}

// transpose read, K = 16: channel-sub `cs` of a wave-private [16 px][ch] bf16 tile -> fragment with k = the 16 pixels
// (lane (pl, g): pixels g*4 .. g*4+3 of channel cs*16 + pl -- the operand layout of v_mfma_f32_16x16x16_bf16)
__device__ __forceinline__ s16x4 tile_frag16(const unsigned char* tile, int pitch, int cs, int pl, int g) {
    // pixel row g * 4 + (pl >> 2): its swizzle key (row >> 2) & 3 is g (tile_put)
    const unsigned char* base = tile + (g * 4 + (pl >> 2)) * pitch + (cs * 16 + ((pl & 3) ^ g) * 4) * 2;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
}
__device__ __forceinline__ f32x4 mma16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// row16_sum (common.h) of four values at once, as v_add_f32_dpp: through __builtin_amdgcn_update_dpp every step came out as
// v_mov_b32 (the `old` operand) + v_mov_b32_dpp + v_add -- 2.5 instructions where one does -- and its zero register, reloaded from
// scratch, drew an s_waitcnt vmcnt(0) in front of the sums that drained the next layer's operands in flight.  The four chains are
// interleaved step by step: a DPP read wants two wait states after the write of its source, the three other chains provide them
// (the s_nop covers the producers of the inputs, which the assembler cannot see).
__device__ __forceinline__ void row16_sum4(float (&v)[4]) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// Backward, TWO workgroups per CU (<= 256 registers).  The first form of this kernel carried two 16-pixel sub-tiles per wave
// through the whole chain so that the pixel-reduction products could use K = 32 MFMAs: 367 registers, one wave per SIMD, and a
// chain of ~60 dependent stages (LayerNorm shuffles, MFMA results, LDS fragments, exp / rcp, parameter vectors fetched from
// global memory inside the loop, eight workgroup barriers per iteration around WAVE-PRIVATE tiles) with nothing to overlap
// it: 1.6k instructions took 13.9k cycles per 32 pixels (57.9 us for 64 x 4096 rows = 0.87 TB/s).  Here a wave carries ONE
// sub-tile at a time, the pixel reductions use the K = 16 MFMA (v_mfma_f32_16x16x16_bf16: 16 pixels = one sub-tile) right
// after the chain, the parameter vectors sit in LDS, and no workgroup barrier is left in the loop (a wave's LDS operations
// execute in order: write tile, read it back transposed); the second wave of each SIMD hides the chain's latencies.
// -DDEC_TIMING: shader-cycle sums per phase of the sub-tile loop (wave 0 of every workgroup) for tools/dec_timeline.py
#ifdef DEC_TIMING
__device__ long long g_dect[4096 * 20];
#define DEC_T(k) do { const long long now_ = (long long)clock64(); t_acc[k] += now_ - t_last; t_last = now_; } while (0)
#else
#define DEC_T(k) do { } while (0)
#endif

template <int MLP, bool STACK>
__device__ __forceinline__ void dec_bwd_body(const DecArgs& p, const int bid, unsigned char* smem) {
#ifdef DEC_TIMING
    long long t_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = (long long)clock64();
    const long long t_begin = t_last;
#endif
    constexpr int NM = MLP / 16, NQ = MLP / 32;
    constexpr int TP32 = lds_pitch(64), TPM = lds_pitch(MLP * 2);       // tile pitches (bytes)
    constexpr int TILE = 16 * (TPM > TP32 ? TPM : TP32);                // bytes per wave-private 16-pixel tile
    using P = PL<MLP>;
    unsigned short* sKq = reinterpret_cast<unsigned short*>(smem);
    unsigned short* sVoT = sKq + 32 * WP;
    unsigned short* sVo = sVoT + 32 * WP;
    unsigned short* sKqT = sVo + 32 * WP;
    unsigned short* sW1 = sKqT + 32 * WP;                   // [MLP][32]
    unsigned short* sW2T = sW1 + MLP * WP;                  // [MLP][32]
    unsigned short* sW1T = sW2T + MLP * WP;                 // [32][MLP]
    constexpr int W1TP = wide_pitch(MLP);
    unsigned char* tiles = reinterpret_cast<unsigned char*>(sW1T + 32 * W1TP);
    // g1, be1, bo, g2, be2, fb1[MLP] of EVERY layer (PARW floats each), staged once, behind the area the final combine reuses
    constexpr int PARW = 5 * 32 + MLP;
    float* sParAll = reinterpret_cast<float*>(smem + (size_t)P::SIZE * 4 * 4);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    unsigned char* tA0 = tiles + (wv * 4 + 0) * TILE;
    unsigned char* tB0 = tiles + (wv * 4 + 1) * TILE;
    unsigned char* tA1 = tiles + (wv * 4 + 2) * TILE;
    unsigned char* tB1 = tiles + (wv * 4 + 3) * TILE;
    const int img = bid / p.bpi, u0 = (bid - img * p.bpi) * p.upb;
    const long row0 = (long)img * p.rows_per_image + (long)u0 * 64;
    const int nsub = min(p.upb, (p.rows_per_image >> 6) - u0);         // 16-pixel sub-tiles per wave
    const int depth = STACK ? p.depth : 1;
    // Everything a layer needs from memory before its first sub-tile -- the seven matrices, the parameter vectors, x / dy of
    // sub-tile 0 -- is REQUESTED while the layer before it (in backward order) parks and reduces its parameter-gradient partials,
    // and committed to LDS after that layer's last barrier.  (Requested at the head of the layer the wait was exposed in every
    // layer of every workgroup: ~6 us per layer and workgroup outside the sub-tile loop against 3.5 us per sub-tile,
    // tools/dec_stack_bench.py at 128 / 256 / 512 rows per workgroup; profiles/r06a.)
    struct {
        StageRegs<32, D> rKq, rVoT, rVo, rKqT;
        StageRegs<MLP, D> rW1, rW2T;
        StageRegs<D, MLP> rW1T;
    } pre;
    uint2 nx[2], ndy[2];
    // this lane's 8 bytes of a row, relative to the first row of the workgroup's sub-tile round (byte offset; the rounds of a
    // wave lie 4 x 16 rows = 4096 bytes apart: a scalar step)
    const unsigned loff = (unsigned)(((wv * 16 + pl) * D + g * 4) * 2);
    auto prefetch = [&](int l, int ll) {
        // (the staging addresses are derived from a value the optimiser cannot see through: hoisted out of the layer loop they
        // would sit in -- or be spilled from -- registers the sub-tile loop needs)
        int ts = tid;
        if constexpr (STACK) asm volatile("" : "+v"(ts));
        const long ko = STACK ? l * p.kq_ls : 0, wo = STACK ? l * p.w_ls : 0;
        // x / dy of this wave's first sub-tile.  (dy of a lower layer is what this very lane stored to dwork in the layer
        // above: program order, same address.)
        {
            const char* xb = reinterpret_cast<const char*>(uniform_ptr((STACK ? (l == 0 ? p.x : p.ys + (l - 1) * p.act_ls) : p.x) + row0 * D));
            const char* gb = reinterpret_cast<const char*>(uniform_ptr((STACK ? (ll == 0 ? p.dy : p.dwork) : p.dy) + row0 * D));
            nx[0] = gld8(xb, loff);
            nx[1] = gld8(xb, loff + 32);
            ndy[0] = gld8(gb, loff);
            ndy[1] = gld8(gb, loff + 32);
        }
        stage_ld(pre.rKq, p.kq + ko + (size_t)img * 32 * D, ts);
        stage_ld(pre.rVoT, p.voT + ko + (size_t)img * D * 32, ts);
        stage_ld(pre.rVo, p.vo + ko + (size_t)img * 32 * D, ts);
        stage_ld(pre.rKqT, p.kqT + ko + (size_t)img * D * 32, ts);
        stage_ld(pre.rW1, p.w1 + wo, ts);
        stage_ld(pre.rW2T, p.w2T + wo, ts);
        stage_ld(pre.rW1T, p.w1T + wo, ts);
    };
    prefetch(depth - 1, 0);
    for (int i = tid; i < depth * PARW; i += 256) {
        const int l = i / PARW, k = i - l * PARW, c = k & 31;
        const float* src = k < 32 ? p.g1 : k < 64 ? p.be1 : k < 96 ? p.bo : k < 128 ? p.g2 : k < 160 ? p.be2 : p.fb1;
        sParAll[i] = src[(STACK ? l * p.par_ls : 0) + (k < 160 ? c : k - 160)];
    }
    // the layers of a stack, last to first (DecArgs::depth): this workgroup's rows only, no synchronisation with any other
#pragma unroll 1
    for (int ll = 0; ll < depth; ++ll) {
    const int l = depth - 1 - ll;
    // (uniform values, pinned to scalar registers: the kernel sits at its 256-register budget)
    const char* x_in = reinterpret_cast<const char*>(uniform_ptr((STACK ? (l == 0 ? p.x : p.ys + (l - 1) * p.act_ls) : p.x) + row0 * D));
    const char* dy_in = reinterpret_cast<const char*>(uniform_ptr((STACK ? (ll == 0 ? p.dy : p.dwork) : p.dy) + row0 * D));
    char* dx_out = const_cast<char*>(reinterpret_cast<const char*>(uniform_ptr((STACK ? (l == 0 ? p.y : p.dwork) : p.y) + row0 * D)));
    if (ll) lds_barrier();                                   // the partial sums of the layer before have left the LDS
    {
        stage_st(sKq, WP, pre.rKq, tid);
        stage_st(sVoT, WP, pre.rVoT, tid);
        stage_st(sVo, WP, pre.rVo, tid);
        stage_st(sKqT, WP, pre.rKqT, tid);
        stage_st(sW1, WP, pre.rW1, tid);
        stage_st(sW2T, WP, pre.rW2T, tid);
        stage_st(sW1T, W1TP, pre.rW1T, tid);
    }
    __syncthreads();
    const float* sPar = sParAll + l * PARW;
    const float *cG1 = sPar, *cBe1 = sPar + 32, *cBo = sPar + 64, *cG2 = sPar + 96, *cBe2 = sPar + 128, *cFb1 = sPar + 160;

    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 aW1[NM][2], aW2[2][NM], aKq[2][2], aVoT[2][2];
#pragma unroll
    for (int a = 0; a < NM; ++a) { aW1[a][0] = aW1[a][1] = zero4; aW2[0][a] = aW2[1][a] = zero4; }
#pragma unroll
    for (int a = 0; a < 2; ++a) { aKq[a][0] = aKq[a][1] = zero4; aVoT[a][0] = aVoT[a][1] = zero4; }
    // The bias gradients db1 = sum_p dz, db2 = sum_p dy, dbo = sum_p dx1 are sums over pixels of operands that sit in the
    // transposed tiles anyway: one more K = 16 MFMA per fragment against a ones matrix leaves them in an accumulator (every
    // column the same; the lanes of column 0 write them) -- no vector add per sub-tile, no cross-lane reduction at the end of
    // the layer (24 of the loop's ~460 vector instructions, 96 of the 224 row-sum steps).  The LayerNorm gradients are sums of
    // fp32 products that no tile holds: lane-local as before.
    f32x4 sb1[NM], sb2[2], sbo[2];
#pragma unroll
    for (int a = 0; a < NM; ++a) sb1[a] = zero4;
    sb2[0] = sb2[1] = sbo[0] = sbo[1] = zero4;
    const s16x4 ones4 = s16x4{(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
    float sg1[2][4], sbe1[2][4], sg2[2][4], sbe2[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) { sg1[a][j] = sbe1[a][j] = sg2[a][j] = sbe2[a][j] = 0.f; }

    // x / dy of the NEXT sub-tile are requested while this one is computed (raw bf16: 8 registers): the loop is one
    // dependent chain, a load at its head would be waited for at HBM latency in every round
    auto request = [&](int it) {
        const char* xr = x_in + (size_t)it * 4096;
        const char* gr = dy_in + (size_t)it * 4096;
        nx[0] = gld8(xr, loff);
        nx[1] = gld8(xr, loff + 32);
        ndy[0] = gld8(gr, loff);
        ndy[1] = gld8(gr, loff + 32);
    };
    auto widen = [](const uint2& u, float (&o)[4]) {
        o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
        o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    };
    DEC_T(10);
#pragma unroll 1
    for (int it = 0; it < nsub; ++it) {
        float x[2][4], dy[2][4];
        widen(nx[0], x[0]);
        widen(nx[1], x[1]);
        widen(ndy[0], dy[0]);
        widen(ndy[1], dy[1]);
        if (it + 1 < nsub) request(it + 1);
        // ---- forward, recomputed from x ----
        float xh1[2][4], xn[2][4], gam[2][4], bet[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) { lds4(cG1 + s * 16 + g * 4, gam[s]); lds4(cBe1 + s * 16 + g * 4, bet[s]); }
        LNres n1;
        {
            float sm = 0.f, q = 0.f;                   // (one pass: see layer_norm)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) { sm += x[h][j]; q += x[h][j] * x[h][j]; }
            n1.mean = group4_sum(sm) * (1.f / D);
            n1.rstd = rsqrtf(fmaxf(group4_sum(q) * (1.f / D) - n1.mean * n1.mean, 0.f) + p.eps);
            const float nmr = -n1.mean * n1.rstd;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh1[h][j] = x[h][j] * n1.rstd + nmr;
                    xn[h][j] = xh1[h][j] * gam[h][j] + bet[h][j];
                }
        }
        const s16x8 kxn = pack8(xn[0], xn[1]);
        DEC_T(0);
        float at[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 d = mma(lds_a(sKq, WP, s * 16 + pl, 0, g), kxn, zero4);
            const float m = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]));
            float e[4], sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { e[j] = __expf(d[j] - m); sum += e[j]; }
            const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int j = 0; j < 4; ++j) at[s][j] = e[j] * inv;
        }
        const s16x8 kat = pack8(at[0], at[1]);
        DEC_T(1);
        float x1[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 o = mma(lds_a(sVoT, WP, s * 16 + pl, 0, g), kat, zero4);
            float bo4[4];
            lds4(cBo + s * 16 + g * 4, bo4);
#pragma unroll
            for (int j = 0; j < 4; ++j) x1[s][j] = o[j] + bo4[j] + x[s][j];
        }
        float xh2[2][4], l2[2][4], gam2[2][4];
        LNres n2;
        {
            float sm = 0.f, q = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) { sm += x1[h][j]; q += x1[h][j] * x1[h][j]; }
            n2.mean = group4_sum(sm) * (1.f / D);
            n2.rstd = rsqrtf(fmaxf(group4_sum(q) * (1.f / D) - n2.mean * n2.mean, 0.f) + p.eps);
            const float nmr = -n2.mean * n2.rstd;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float b4[4];
                lds4(cG2 + h * 16 + g * 4, gam2[h]);
                lds4(cBe2 + h * 16 + g * 4, b4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh2[h][j] = x1[h][j] * n2.rstd + nmr;
                    l2[h][j] = xh2[h][j] * gam2[h][j] + b4[j];
                }
            }
        }
        const s16x8 kl2 = pack8(l2[0], l2[1]);
        DEC_T(2);
        float dg[NM][4], hh[NM][4];                  // gelu'(z), gelu(z)
#pragma unroll
        for (int s = 0; s < NM; ++s) {
            f32x4 zz = mma(lds_a(sW1, WP, s * 16 + pl, 0, g), kl2, zero4);
            float b4[4];
            lds4(cFb1 + s * 16 + g * 4, b4);
#pragma unroll
            for (int j = 0; j < 4; ++j) hh[s][j] = gelu_fast(zz[j] + b4[j], &dg[s][j]);
        }
        s16x8 kh[NQ], kdz[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) kh[q] = pack8(hh[2 * q], hh[2 * q + 1]);
        DEC_T(3);
        // ---- backward chain ----
        const s16x8 kdy = pack8(dy[0], dy[1]);
        // dW2[c][m] += dy^T h  (tiles A0 / B0)
        tile_put(tA0, TP32, pl, 0, g, kdy);
#pragma unroll
        for (int q = 0; q < NQ; ++q) tile_put(tB0, TPM, pl, 32 * q, g, kh[q]);
        float dz[NM][4];
#pragma unroll
        for (int s = 0; s < NM; ++s) {
            f32x4 dh = mma(lds_a(sW2T, WP, s * 16 + pl, 0, g), kdy, zero4);
#pragma unroll
            for (int j = 0; j < 4; ++j) dz[s][j] = dh[j] * dg[s][j];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) kdz[q] = pack8(dz[2 * q], dz[2 * q + 1]);
        // dW1[m][c] += dz^T l2  (tiles A1 / B1)
#pragma unroll
        for (int q = 0; q < NQ; ++q) tile_put(tA1, TPM, pl, 32 * q, g, kdz[q]);
        tile_put(tB1, TP32, pl, 0, g, kl2);
        DEC_T(4);
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const s16x4 fa = tile_frag16(tA0, TP32, sc, pl, g);
#pragma unroll
            for (int sm = 0; sm < NM; ++sm) aW2[sc][sm] = mma16(fa, tile_frag16(tB0, TPM, sm, pl, g), aW2[sc][sm]);
            sb2[sc] = mma16(fa, ones4, sb2[sc]);
        }
        f32x4 dl2[2] = {zero4, zero4};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int s = 0; s < 2; ++s) dl2[s] = mma(lds_a(sW1T, W1TP, s * 16 + pl, 32 * q, g), kdz[q], dl2[s]);
        DEC_T(5);
        // LayerNorm-2 backward (+ residual)
        float gh[2][4], sa = 0.f, sbb = 0.f, dx1[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sg2[s][j] += dl2[s][j] * xh2[s][j];
                sbe2[s][j] += dl2[s][j];
                gh[s][j] = dl2[s][j] * gam2[s][j];
                sa += gh[s][j];
                sbb += gh[s][j] * xh2[s][j];
            }
        sa = group4_sum(sa);
        sbb = group4_sum(sbb);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dx1[s][j] = n2.rstd * (gh[s][j] - (sa + xh2[s][j] * sbb) * (1.f / D)) + dy[s][j];
            }
        const s16x8 kdx1 = pack8(dx1[0], dx1[1]);
        DEC_T(6);
#pragma unroll
        for (int sm = 0; sm < NM; ++sm) {
            const s16x4 fa = tile_frag16(tA1, TPM, sm, pl, g);
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) aW1[sm][sc] = mma16(fa, tile_frag16(tB1, TP32, sc, pl, g), aW1[sm][sc]);
            sb1[sm] = mma16(fa, ones4, sb1[sm]);
        }
        // dVoT[c][hl] += dx1^T attn  (tiles A0 / B0 again: the reads of dW2 were issued above, LDS runs a wave's operations in order)
        tile_put(tA0, TP32, pl, 0, g, kdx1);
        tile_put(tB0, TP32, pl, 0, g, kat);
        // attention backward
        float dd[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 da = mma(lds_a(sVo, WP, s * 16 + pl, 0, g), kdx1, zero4);
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dot += at[s][j] * da[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) dd[s][j] = at[s][j] * (da[j] - dot);
        }
        const s16x8 kdd = pack8(dd[0], dd[1]);
        DEC_T(7);
        // dKq[hl][c] += dd^T xn  (tiles A1 / B1 again)
        tile_put(tA1, TP32, pl, 0, g, kdd);
        tile_put(tB1, TP32, pl, 0, g, kxn);
#pragma unroll
        for (int sr = 0; sr < 2; ++sr) {
            const s16x4 fa = tile_frag16(tA0, TP32, sr, pl, g);
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) aVoT[sr][sc] = mma16(fa, tile_frag16(tB0, TP32, sc, pl, g), aVoT[sr][sc]);
            sbo[sr] = mma16(fa, ones4, sbo[sr]);
        }
        float dxn[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 t = mma(lds_a(sKqT, WP, s * 16 + pl, 0, g), kdd, zero4);
#pragma unroll
            for (int j = 0; j < 4; ++j) dxn[s][j] = t[j];
        }
        sa = 0.f; sbb = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sg1[s][j] += dxn[s][j] * xh1[s][j];
                sbe1[s][j] += dxn[s][j];
                gh[s][j] = dxn[s][j] * gam[s][j];
                sa += gh[s][j];
                sbb += gh[s][j] * xh1[s][j];
            }
        sa = group4_sum(sa);
        sbb = group4_sum(sbb);
        char* dxr = dx_out + (size_t)it * 4096;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = n1.rstd * (gh[s][j] - (sa + xh1[s][j] * sbb) * (1.f / D)) + dx1[s][j];
            gst8(dxr, loff + s * 32, make_uint2(f2bf2(r[0], r[1]), f2bf2(r[2], r[3])));
        }
        DEC_T(8);
#pragma unroll
        for (int sr = 0; sr < 2; ++sr) {
            const s16x4 fa = tile_frag16(tA1, TP32, sr, pl, g);
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) aKq[sr][sc] = mma16(fa, tile_frag16(tB1, TP32, sc, pl, g), aKq[sr][sc]);
        }
        DEC_T(9);
    }

    // ---- combine the 4 wavefronts deterministically, then write this workgroup's partial ----
    // Every wave parks its sums in its OWN slot (all four at once; the pixel-lane reductions are DPP row sums) and the slots
    // are added in wave order.  (Measured alternatives: turn-taking through ONE slot with ds_bpermute butterflies inside each
    // wave's turn took 40 of the first kernel's 70 us; two slots with waves 2 / 3 ADDING onto waves 0 / 1 -- 130 dependent
    // LDS read-modify-writes the compiler may not reorder -- 17k of this kernel's 74k cycles, tools/dec_timeline.py.)
    // The matrices are parked AS THE ACCUMULATORS HOLD THEM: accumulator a of a matrix at floats [a * 256 + lane * 4 + j] of its
    // region -- one conflict-free ds_write_b128 per accumulator and lane (in row-major order they were four 4-byte writes D
    // floats apart each) -- and the partial keeps that order in memory; dec_bwd_finalize_kernel decodes it (PL::matrix_index).
    __syncthreads();                                         // every wave is done with the weights / tiles
    DEC_T(12);
    float* red = reinterpret_cast<float*>(smem);             // [4][P::SIZE] floats (host sized the LDS)
    float* mine = red + (size_t)wv * P::SIZE;
    {
        f32x4* m4 = reinterpret_cast<f32x4*>(mine);
#pragma unroll
        for (int sm = 0; sm < NM; ++sm)
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                m4[P::W1 / 4 + (sm * 2 + sc) * 64 + lane] = aW1[sm][sc];
                m4[P::W2 / 4 + (sc * NM + sm) * 64 + lane] = aW2[sc][sm];
            }
#pragma unroll
        for (int sr = 0; sr < 2; ++sr)
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                m4[P::KQ / 4 + (sr * 2 + sc) * 64 + lane] = aKq[sr][sc];
                m4[P::VOT / 4 + (sr * 2 + sc) * 64 + lane] = aVoT[sr][sc];
            }
    }
    DEC_T(13);
    // (here: the 64 accumulator registers are free, and from here to the head of the next layer only LDS-ordering barriers)
    if (ll + 1 < depth) prefetch(l - 1, ll + 1);             // in flight while the column sums are reduced, the partial written
    // lane-local column sums: over the 16 pixel lanes of the row.  All the DPP row sums first (independent chains, full
    // exec), then ONE masked block of 16-byte stores: a masked 4-byte store after every sum toggled exec 56 times and took
    // 3.9k of the workgroup's 65k cycles (tools/dec_timeline.py)
#pragma unroll
    for (int s = 0; s < 2; ++s) { row16_sum4(sg1[s]); row16_sum4(sbe1[s]); row16_sum4(sg2[s]); row16_sum4(sbe2[s]); }
    if (pl == 0) {
        auto put4 = [&](int off, const float (&v)[4]) { *reinterpret_cast<float4*>(mine + off + g * 4) = make_float4(v[0], v[1], v[2], v[3]); };
        auto putv = [&](int off, const f32x4& v) { *reinterpret_cast<f32x4*>(mine + off + g * 4) = v; };
#pragma unroll
        for (int s = 0; s < NM; ++s) putv(P::B1 + s * 16, sb1[s]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            putv(P::B2 + s * 16, sb2[s]); putv(P::BO + s * 16, sbo[s]); put4(P::G1 + s * 16, sg1[s]);
            put4(P::BE1 + s * 16, sbe1[s]); put4(P::G2 + s * 16, sg2[s]); put4(P::BE2 + s * 16, sbe2[s]);
        }
    }
    DEC_T(14);
    lds_barrier();
    DEC_T(15);
    f32x4* out = reinterpret_cast<f32x4*>(p.partial + (STACK ? l * p.part_ls : 0) + (size_t)bid * P::SIZE);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(red);
    for (int i = tid; i < P::SIZE / 4; i += 256)
        out[i] = ((r4[i] + r4[P::SIZE / 4 + i]) + r4[2 * (P::SIZE / 4) + i]) + r4[3 * (P::SIZE / 4) + i];
    }   // layers
#ifdef DEC_TIMING
    DEC_T(11);
    if (tid == 0 && blockIdx.x < 4096) {
        for (int k = 0; k < 16; ++k) g_dect[blockIdx.x * 20 + k] = t_acc[k];
        g_dect[blockIdx.x * 20 + 16] = p.upb;
        g_dect[blockIdx.x * 20 + 17] = t_last - t_begin;
    }
#endif
}
#ifdef DEC_TIMING
extern "C" int dh_debug_dect(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_dect), (size_t)n * 8); }
#endif

template <int MLP, bool STACK = false>
__global__ __launch_bounds__(256, MLP == 32 ? 2 : 1) void dec_bwd_kernel(DecArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    dec_bwd_body<MLP, STACK>(p, blockIdx.x, smem);
}
template <int MLP, bool STACK = false>
__global__ __launch_bounds__(256, MLP == 32 ? 2 : 1) void dec_bwd_multi_kernel(DecMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const DecArgs p = m.a[j];
    dec_bwd_body<MLP, STACK>(p, (int)blockIdx.x - m.first[j], smem);
}

// sums the workgroup partials: shared parameters over all workgroups (accumulated into the gradient arena),
// per-image dKq / dVoT over the workgroups of that image (assigned)
// blockIdx.z = layer of a decoder stack whose backward launches left their partials `pstride` floats apart (layer 0 .. depth-1
// of the SAME shapes): the layers' parameters sit `gstride` floats apart in the gradient arena, dkq / dvoT `kstride` apart.
struct FinArgs {
    const float* partial;
    int nblk, bpi;
    float *dw1, *dw2, *db1, *db2, *dbo, *dg1, *dbe1, *dg2, *dbe2, *dkq, *dvoT;
    long pstride, gstride, kstride;
};
template <int MLP>
__device__ __forceinline__ void dec_bwd_finalize_body(const FinArgs& fa, const uint3 bid) {
    using P = PL<MLP>;
    __shared__ double red[8][32];
    const float* partial = fa.partial;
    const int nblk = fa.nblk, bpi = fa.bpi;
    float *dw1 = fa.dw1, *dw2 = fa.dw2, *db1 = fa.db1, *db2 = fa.db2, *dbo = fa.dbo, *dg1 = fa.dg1, *dbe1 = fa.dbe1,
          *dg2 = fa.dg2, *dbe2 = fa.dbe2, *dkq = fa.dkq, *dvoT = fa.dvoT;
    const long pstride = fa.pstride, gstride = fa.gstride, kstride = fa.kstride;
    {
        const long z = bid.z;
        partial += z * pstride;
        dw1 += z * gstride; dw2 += z * gstride; db1 += z * gstride; db2 += z * gstride; dbo += z * gstride;
        dg1 += z * gstride; dbe1 += z * gstride; dg2 += z * gstride; dbe2 += z * gstride;
        dkq += z * kstride; dvoT += z * kstride;
    }
    const int lane = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const int i = bid.x * 32 + lane;                 // element of the shared-parameter part
    if (bid.y == 0) {
        double s = 0.0;
        if (i < P::KQ) {
            // four independent partial sums: the loop is otherwise one chain of dependent-latency loads
            double s1 = 0.0, s2 = 0.0, s3 = 0.0;
            const float* src = partial + i;
            int b = ph;
            for (; b + 24 < nblk; b += 32) {
                s += (double)src[(size_t)b * P::SIZE];
                s1 += (double)src[(size_t)(b + 8) * P::SIZE];
                s2 += (double)src[(size_t)(b + 16) * P::SIZE];
                s3 += (double)src[(size_t)(b + 24) * P::SIZE];
            }
            for (; b < nblk; b += 8) s += (double)src[(size_t)b * P::SIZE];
            s = (s + s1) + (s2 + s3);
        }
        red[ph][lane] = s;
        __syncthreads();
        if (ph == 0 && i < P::KQ) {
            double t = 0.0;
            for (int r = 0; r < 8; ++r) t += red[r][lane];
            float* dst;
            int o;
            if (i < P::W2) { dst = dw1; o = P::matrix_index(i - P::W1, D); }
            else if (i < P::B1) { dst = dw2; o = P::matrix_index(i - P::W2, MLP); }
            else if (i < P::B2) { dst = db1; o = i - P::B1; }
            else if (i < P::BO) { dst = db2; o = i - P::B2; }
            else if (i < P::G1) { dst = dbo; o = i - P::BO; }
            else if (i < P::BE1) { dst = dg1; o = i - P::G1; }
            else if (i < P::G2) { dst = dbe1; o = i - P::BE1; }
            else if (i < P::BE2) { dst = dg2; o = i - P::G2; }
            else { dst = dbe2; o = i - P::BE2; }
            dst[o] += (float)t;
        }
    } else {
        const int img = bid.y - 1;
        const int e = bid.x * 256 + threadIdx.x;     // element of [dKq | dVoT] (2048)
        if (e < 2048) {
            float s = 0.f;
            const float* src = partial + (size_t)img * bpi * P::SIZE + P::KQ + e;
#pragma unroll 4
            for (int b = 0; b < bpi; ++b) s += src[(size_t)b * P::SIZE];       // (order kept: loads issue ahead, adds in order)
            if (e < 1024) dkq[(size_t)img * 1024 + P::matrix_index(e, D)] = s;
            else dvoT[(size_t)img * 1024 + P::matrix_index(e - 1024, 32)] = s;
        }
    }
}
// The finalize's workgroups as ONE line: first the gx x depth column sums of the shared parameters (bid.y = 0), then the 8 x images
// x depth workgroups of the per-image dKq / dVoT sums.  (As a gx x (1 + images) x depth grid 63 of the 71 workgroups of every image
// row had nothing to do: 73 840 workgroups for the three levels of a DAHiTra pass, 8 800 of them with work.)
__host__ __device__ inline int fin_blocks(int gx, int images, int depth) { return (gx + 8 * images) * depth; }
__device__ __forceinline__ uint3 fin_locate(int local, int gx, int images, int depth) {
    uint3 bid;
    if (local < gx * depth) { bid.x = local % gx; bid.y = 0; bid.z = local / gx; }
    else {
        local -= gx * depth;
        bid.x = local % 8; local /= 8;
        bid.y = 1 + local % images;
        bid.z = local / images;
    }
    return bid;
}
template <int MLP>
__global__ __launch_bounds__(256) void dec_bwd_finalize_kernel(FinArgs fa, int gx, int images, int depth) {
    dec_bwd_finalize_body<MLP>(fa, fin_locate((int)blockIdx.x, gx, images, depth));
}
// the finalizes of several independent stacks in one launch (dh_decoder_batch_*): workgroups [first[j], first[j + 1]) run job j's
// gx x gy[j] x depth grid
struct FinMulti {
    int n;
    int first[DEC_MAXJ + 1], gx[DEC_MAXJ], gy[DEC_MAXJ], depth[DEC_MAXJ];
    FinArgs a[DEC_MAXJ];
};
template <int MLP>
__global__ __launch_bounds__(256) void dec_bwd_finalize_multi_kernel(FinMulti m) {
    int j = 0;
    while (j + 1 < m.n && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const int local = (int)blockIdx.x - m.first[j];
    dec_bwd_finalize_body<MLP>(m.a[j], fin_locate(local, m.gx[j], m.gy[j] - 1, m.depth[j]));
}

template <int MLP> size_t bwd_lds_bytes() {
    const size_t w = (size_t)(4 * 32 * WP + 2 * MLP * WP + 32 * wide_pitch(MLP)) * 2;
    const size_t tile = 16 * (size_t)(lds_pitch(MLP * 2) > lds_pitch(64) ? lds_pitch(MLP * 2) : lds_pitch(64));
    size_t t = w + 16 * tile;                                 // four tiles per wave
    static_assert((size_t)(4 * 32 * WP + 2 * MLP * WP + 32 * wide_pitch(MLP)) * 2 + 16 * 16 * (size_t)(lds_pitch(MLP * 2) > lds_pitch(64) ? lds_pitch(MLP * 2) : lds_pitch(64))
                  <= (size_t)PL<MLP>::SIZE * 4 * 4, "the parameter table sits behind the four wave slots of the final combine");
    t = (size_t)PL<MLP>::SIZE * 4 * 4;                        // the four wave slots of the final combine (>= weights + tiles)
    return t + (size_t)DEC_MAXDEPTH * (5 * 32 + MLP) * 4;     // + the parameter vectors of every layer
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)


// ---- block sizes (DecArgs::upb / bpi) ---------------------------------------------------------------------------------
// A plan cuts every image of a job into blocks of `upb` 64-row units.  The DEFAULT plans are the fixed rules of the first builds
// (powers of two that divide an image); launches that hold several jobs (dh_decoder_batch_*: DAHiTra's three levels) are
// re-planned as a whole by dec_balance when they are issued.
struct DecPlan { int upb, bpi; };
static inline DecPlan dec_plan(int rows_per_image, int upb) {
    const int upi = rows_per_image / 64;
    upb = std::max(1, std::min(upb, upi));
    return DecPlan{upb, (upi + upb - 1) / upb};
}
// rows per forward workgroup: every workgroup stages the image's kq / voT and the two weight matrices first (a quarter of the
// instructions of a 128-row workgroup), so large launches take more rows per workgroup while >= 1024 workgroups remain
static inline DecPlan dec_fwd_default(long rows, int rows_per_image) {
    static const long minblk = getenv("DAHITRA_DEC_FWD_MINBLK") ? atol(getenv("DAHITRA_DEC_FWD_MINBLK")) : 1024;
    int rpb = 512;
    while (rpb > 64 && (rows_per_image % rpb || rows / rpb < minblk)) rpb >>= 1;
    return dec_plan(rows_per_image, rpb / 64);
}
// rows per backward workgroup: 512 at most (every workgroup writes a PL::SIZE partial), fewer while that leaves less than 256
// workgroups; per layer a workgroup spends several microseconds outside its sub-tile loop (parking and reducing the
// parameter-gradient partials, writing them: tools/dec_timeline.py), so the small levels of a staged launch want rows, not
// workgroups.  MLP = 64 (one workgroup per CU at 344 registers: 256 resident) takes up to 1024 rows: 256 workgroups in ONE
// round instead of 512 in two (+0.3 % on the s4 step; DAHITRA_DEC_BWD_MAXRPB64=512 restores the old rule)
static inline DecPlan dec_bwd_default(long rows, int rows_per_image, int mlp) {
    static const long minblk = getenv("DAHITRA_DEC_BWD_MINBLK") ? atol(getenv("DAHITRA_DEC_BWD_MINBLK")) : 256;
    static const int max64 = getenv("DAHITRA_DEC_BWD_MAXRPB64") ? atoi(getenv("DAHITRA_DEC_BWD_MAXRPB64")) : 1024;
    int rpb = mlp == 64 ? max64 : 512;
    while (rpb > 64 && (rows_per_image % rpb || rows / rpb < minblk)) rpb >>= 1;
    return dec_plan(rows_per_image, rpb / 64);
}
// the smallest backward block a re-plan may choose: the partial workspace (dh_decoder_layer_bwd_workspace_size) is sized for it
static inline int dec_bwd_min_upb(long rows, int rows_per_image, int mlp) {
    return std::max(1, dec_bwd_default(rows, rows_per_image, mlp).upb / 2);
}
// The plan a backward launch used, by the address of its partial workspace: the finalize (a separate call, possibly issued in a
// later round of the batch) sums exactly the blocks that launch wrote.
static thread_local std::unordered_map<const void*, DecPlan> g_bwd_plan;
static void dec_remember_plan(const void* partial, DecPlan pl) {
    if (g_bwd_plan.size() > 512) g_bwd_plan.clear();
    g_bwd_plan[partial] = pl;
}
static DecPlan dec_recall_plan(const void* partial, long rows, int rows_per_image, int mlp) {
    auto it = g_bwd_plan.find(partial);
    return it != g_bwd_plan.end() ? it->second : dec_bwd_default(rows, rows_per_image, mlp);
}

// dec_balance: block sizes for ALL jobs of one launch (dh_decoder_batch_*: DAHiTra's three levels).  These kernels are bound by
// the vector ALU of a CU, not by resident workgroups: one backward workgroup alone on a CU takes ~1.75 us per 64-row unit, two
// share it at ~3.5 us each -- the same throughput -- while every (workgroup, layer) pays for staging the layer's matrices and for
// parking, reducing and writing 17 KB of parameter-gradient partials.  So: as FEW, LARGE blocks as keep every CU busy.  The job
// with the most work gets the largest upb <= 16 that still leaves `per_cu` blocks per CU (backward 1, forward 2: its
// workgroups are short and three are resident); the smaller jobs ride along with blocks no smaller than that.  Measured on the
// three levels of a DAHiTra pass (tools/dec_stack_bench.py, profiles/r06a_dec_stack.txt): backward 293 -> 289 us (64 images) and
// 166 -> 160 us (32 images) against the per-job rules, forward 146 -> 139 and 76 -> 69 us; a list-scheduling model over
// resident-workgroup slots (the first form of this function) predicted gains it did not deliver, because a slot is not a
// processor.  DAHITRA_DEC_BALANCE=0: the per-job default plans.  DAHITRA_DEC_UPB_FWD / _BWD=<n>: that block size for every job.
static void dec_balance(DecMulti& m, int n, bool bwd, int mlp, int cus) {
    static const bool on = !(getenv("DAHITRA_DEC_BALANCE") && atoi(getenv("DAHITRA_DEC_BALANCE")) == 0);
    static const int force_f = getenv("DAHITRA_DEC_UPB_FWD") ? atoi(getenv("DAHITRA_DEC_UPB_FWD")) : 0;
    static const int force_b = getenv("DAHITRA_DEC_UPB_BWD") ? atoi(getenv("DAHITRA_DEC_UPB_BWD")) : 0;
    int upb[DEC_MAXJ], lo[DEC_MAXJ];
    int big = 0;
    for (int j = 0; j < n; ++j) {
        const DecArgs& a = m.a[j];
        upb[j] = a.upb;
        lo[j] = bwd ? dec_bwd_min_upb(a.rows, a.rows_per_image, mlp) : 1;
        if ((a.depth > 1 ? a.depth : 1) * a.rows > (m.a[big].depth > 1 ? m.a[big].depth : 1) * m.a[big].rows) big = j;
    }
    const int force = bwd ? force_b : force_f;
    if (force > 0) {
        for (int j = 0; j < n; ++j) upb[j] = std::max(lo[j], std::min(force, m.a[j].rows_per_image / 64));
    } else if (on && cus > 0 && n > 1) {
        const DecArgs& a = m.a[big];
        const int upi = a.rows_per_image / 64, images = (int)(a.rows / a.rows_per_image), want = (bwd ? 1 : 2) * cus;
        // (block sizes that divide an image: a ragged last block is a short workgroup next to long ones)
        int u = std::min(16, upi);
        while (u > lo[big] && (upi % u || images * (upi / u) < want)) --u;
        if (upi % u) u = upb[big];
        upb[big] = u;
        for (int j = 0; j < n; ++j)
            if (j != big) {
                int v = std::min(std::min(16, m.a[j].rows_per_image / 64), std::max(u, 4));
                while (v > lo[j] && (m.a[j].rows_per_image / 64) % v) --v;
                upb[j] = std::max(lo[j], v);
            }
    }
    if (getenv("DAHITRA_DEC_BALANCE_LOG")) {
        fprintf(stderr, "[dec_balance] %s mlp %d cus %d:", bwd ? "bwd" : "fwd", mlp, cus);
        for (int j = 0; j < n; ++j)
            fprintf(stderr, "  (%ld x %d rows, depth %d) upb %d -> %d", m.a[j].rows / m.a[j].rows_per_image, m.a[j].rows_per_image, m.a[j].depth,
                    m.a[j].upb, upb[j]);
        fprintf(stderr, "\n");
    }
    m.first[0] = 0;
    for (int j = 0; j < n; ++j) {
        const DecPlan pl = dec_plan(m.a[j].rows_per_image, upb[j]);
        m.a[j].upb = pl.upb;
        m.a[j].bpi = pl.bpi;
        m.first[j + 1] = m.first[j] + (int)(m.a[j].rows / m.a[j].rows_per_image) * pl.bpi;
        if (bwd) dec_remember_plan(m.a[j].partial, pl);
    }
}
static int dec_cus() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 0;
        if (!cus) (void)hipGetLastError();
    }
    return cus;
}

// ---- batched launches (dh_decoder_batch_*): index 0 = MLP 32, 1 = MLP 64; forward and backward each
struct DecBatch {      // index = (MLP == 64) + 2 * (layer-fused stack)
    bool on = false;
    int nf[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0};
    DecMulti f[4], b[4];
    int nfin[2] = {0, 0};      // recorded stack finalizes (MLP 32 / 64): issued after the backward launches
    FinMulti fin[2];
};
static thread_local DecBatch g_db;
template <int MLP> static int dec_set_bwd_lds(const void* kern, bool& done) {
    if (!done) {
        done = true;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bwd_lds_bytes<MLP>()) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("decoder_layer_bwd: cannot raise dynamic LDS to %zu", bwd_lds_bytes<MLP>());
        }
    }
    return 0;
}
// Longest workgroups FIRST: the workgroups of a launch are dispatched in index order, and a workgroup of a layer-fused stack
// runs depth x (rows per workgroup / 64) sub-tile rounds back to back -- DAHiTra's 64 x 64 level: 4 layers x 8 sub-tiles, ~140 us
// of the backward, against 8 rounds for the other two levels.  Recorded in the order the levels run (16 x 16 first) the long
// workgroups started only when the short ones had drained and set the launch's length by themselves: 285 us for ~190 us of work.
static void dec_sort_jobs(DecMulti& m, int n) {
    if (getenv("DAHITRA_DEC_NO_SORT")) return;
    int nblk[DEC_MAXJ];
    for (int j = 0; j < n; ++j) nblk[j] = m.first[j + 1] - m.first[j];
    for (int i = 1; i < n; ++i)                      // insertion sort, stable: n <= 4
        for (int j = i; j > 0; --j) {
            const long wa = (long)(m.a[j].depth > 1 ? m.a[j].depth : 1) * m.a[j].upb;
            const long wb = (long)(m.a[j - 1].depth > 1 ? m.a[j - 1].depth : 1) * m.a[j - 1].upb;
            if (wa <= wb) break;
            const DecArgs t = m.a[j]; m.a[j] = m.a[j - 1]; m.a[j - 1] = t;
            const int tb = nblk[j]; nblk[j] = nblk[j - 1]; nblk[j - 1] = tb;
        }
    m.first[0] = 0;
    for (int j = 0; j < n; ++j) m.first[j + 1] = m.first[j] + nblk[j];
}
static int dec_batch_flush(hipStream_t st) {
    DecBatch& d = g_db;
    static bool m32 = false, m64 = false, s32 = false, s64 = false;
    for (int k = 0; k < 4; ++k) {
        if (d.nf[k]) dec_balance(d.f[k], d.nf[k], false, k & 1 ? 64 : 32, dec_cus());
        if (d.nb[k]) dec_balance(d.b[k], d.nb[k], true, k & 1 ? 64 : 32, dec_cus());
        if (d.nf[k] > 1) dec_sort_jobs(d.f[k], d.nf[k]);
        if (d.nb[k] > 1) dec_sort_jobs(d.b[k], d.nb[k]);
        if (d.nf[k]) {
            d.f[k].n = d.nf[k];
            const int total = d.f[k].first[d.nf[k]];
            if (k == 0) hipLaunchKernelGGL((dec_fwd_multi_kernel<32, false>), dim3(total), dim3(256), 0, st, d.f[k]);
            else if (k == 1) hipLaunchKernelGGL((dec_fwd_multi_kernel<64, false>), dim3(total), dim3(256), 0, st, d.f[k]);
            else if (k == 2) hipLaunchKernelGGL((dec_fwd_multi_kernel<32, true>), dim3(total), dim3(256), 0, st, d.f[k]);
            else hipLaunchKernelGGL((dec_fwd_multi_kernel<64, true>), dim3(total), dim3(256), 0, st, d.f[k]);
            d.nf[k] = 0;
        }
        if (d.nb[k]) {
            d.b[k].n = d.nb[k];
            const int total = d.b[k].first[d.nb[k]];
            if (k == 0) {
                if (dec_set_bwd_lds<32>(reinterpret_cast<const void*>(dec_bwd_multi_kernel<32, false>), m32)) return 1;
                hipLaunchKernelGGL((dec_bwd_multi_kernel<32, false>), dim3(total), dim3(256), bwd_lds_bytes<32>(), st, d.b[k]);
            } else if (k == 1) {
                if (dec_set_bwd_lds<64>(reinterpret_cast<const void*>(dec_bwd_multi_kernel<64, false>), m64)) return 1;
                hipLaunchKernelGGL((dec_bwd_multi_kernel<64, false>), dim3(total), dim3(256), bwd_lds_bytes<64>(), st, d.b[k]);
            } else if (k == 2) {
                if (dec_set_bwd_lds<32>(reinterpret_cast<const void*>(dec_bwd_multi_kernel<32, true>), s32)) return 1;
                hipLaunchKernelGGL((dec_bwd_multi_kernel<32, true>), dim3(total), dim3(256), bwd_lds_bytes<32>(), st, d.b[k]);
            } else {
                if (dec_set_bwd_lds<64>(reinterpret_cast<const void*>(dec_bwd_multi_kernel<64, true>), s64)) return 1;
                hipLaunchKernelGGL((dec_bwd_multi_kernel<64, true>), dim3(total), dim3(256), bwd_lds_bytes<64>(), st, d.b[k]);
            }
            d.nb[k] = 0;
        }
    }
    for (int k = 0; k < 2; ++k)
        if (d.nfin[k]) {
            d.fin[k].n = d.nfin[k];
            for (int j = 0; j < d.nfin[k]; ++j) {        // the blocks the backward launch of this workspace wrote (dec_balance)
                FinArgs& fa = d.fin[k].a[j];
                const int images = d.fin[k].gy[j] - 1;
                const auto it = g_bwd_plan.find(fa.partial);
                if (it != g_bwd_plan.end()) { fa.bpi = it->second.bpi; fa.nblk = images * fa.bpi; }
            }
            const int total = d.fin[k].first[d.nfin[k]];
            if (k == 0) hipLaunchKernelGGL(dec_bwd_finalize_multi_kernel<32>, dim3(total), dim3(256), 0, st, d.fin[k]);
            else hipLaunchKernelGGL(dec_bwd_finalize_multi_kernel<64>, dim3(total), dim3(256), 0, st, d.fin[k]);
            d.nfin[k] = 0;
        }
    DH_CHECK_LAUNCH("decoder_batch");
    return 0;
}
static int dec_batch_record(DecMulti* m, int* n, const DecArgs& a, int nblk, hipStream_t st) {
    // (no flush on overflow: a recorded stack launch may depend on a recorded operand preparation of tokens.hip that only
    // dh_xprep_batch_launch_fwd issues first -- more than DEC_MAXJ independent stacks per round is an error, not a reorder)
    DH_REQUIRE(*n < DEC_MAXJ, "decoder_batch: more than %d stacks recorded in one round (dh_decoder_batch_launch first)", DEC_MAXJ);
    if (*n == 0) m->first[0] = 0;
    m->a[*n] = a;
    m->first[*n + 1] = m->first[*n] + nblk;
    ++*n;
    return 0;
}

static int check_common(long rows, int rows_per_image, int mlp) {
    DH_REQUIRE(mlp == 32 || mlp == 64, "decoder_fused: mlp_dim must be 32 or 64, got %d", mlp);
    DH_REQUIRE(rows_per_image % 128 == 0 && rows % rows_per_image == 0, "decoder_fused: rows per image (%d) must be a multiple of 128", rows_per_image);
    return 0;
}

// x, y: [rows][32] bf16; kq, voT: [images][32][32] bf16 (dh_xattn_prep_fwd); w1: [mlp][32], w2: [32][mlp] bf16
extern "C" int dh_decoder_layer_fwd(const void* x, void* y, const void* kq, const void* voT, const float* ln1_g,
                                    const float* ln1_b, const float* bo, const float* ln2_g, const float* ln2_b,
                                    const void* w1, const float* b1, const void* w2, const float* b2, long rows,
                                    int rows_per_image, int mlp, float eps, void* stream) {
    if (check_common(rows, rows_per_image, mlp)) return 1;
    DecArgs a = {};
    a.x = (const bf16*)x; a.y = (bf16*)y; a.kq = (const bf16*)kq; a.voT = (const bf16*)voT;
    a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2;
    a.g1 = ln1_g; a.be1 = ln1_b; a.bo = bo; a.g2 = ln2_g; a.be2 = ln2_b; a.fb1 = b1; a.fb2 = b2;
    a.rows_per_image = rows_per_image; a.rows = rows; a.eps = eps;
    const DecPlan pl = dec_fwd_default(rows, rows_per_image);
    a.upb = pl.upb; a.bpi = pl.bpi;
    const int grid = (int)(rows / rows_per_image) * pl.bpi;
    if (g_db.on) return dec_batch_record(&g_db.f[mlp == 64], &g_db.nf[mlp == 64], a, grid, ST(stream));
    if (mlp == 64) hipLaunchKernelGGL(dec_fwd_kernel<64>, dim3(grid), dim3(256), 0, ST(stream), a);
    else hipLaunchKernelGGL(dec_fwd_kernel<32>, dim3(grid), dim3(256), 0, ST(stream), a);
    DH_CHECK_LAUNCH("decoder_layer_fwd");
    return 0;
}

// Batched decoder layers: between dh_decoder_batch_begin() and _end(), dh_decoder_layer_fwd and the data-gradient-only form of
// dh_decoder_layer_bwd (dw1 == NULL: partials left for dh_decoder_stack_bwd_finalize) only RECORD their launch (up to 4 per
// direction and MLP width; a fifth issues the first four); dh_decoder_batch_launch(stream) issues the recorded layers of
// INDEPENDENT stacks as one launch per direction and width.  Every buffer of a recorded call stays alive and unchanged until
// then.  Per host thread; _abort drops the recorded calls.
static void dec_batch_clear() { for (int k = 0; k < 4; ++k) g_db.nf[k] = g_db.nb[k] = 0; g_db.nfin[0] = g_db.nfin[1] = 0; }
extern "C" int dh_decoder_batch_begin() { g_db.on = true; dec_batch_clear(); return 0; }
extern "C" int dh_decoder_batch_pending() { int n = g_db.nfin[0] + g_db.nfin[1]; for (int k = 0; k < 4; ++k) n += g_db.nf[k] + g_db.nb[k]; return n; }
extern "C" int dh_decoder_batch_launch(void* stream) { return dec_batch_flush(ST(stream)); }
extern "C" int dh_decoder_batch_end(void* stream) { const int rc = dec_batch_flush(ST(stream)); g_db.on = false; return rc; }
extern "C" int dh_decoder_batch_abort() { g_db.on = false; dec_batch_clear(); return 0; }

extern "C" long dh_decoder_layer_bwd_workspace_size(long rows, int rows_per_image, int mlp) {
    // (sized for the smallest block a re-planned launch may use, dec_balance: up to twice the default plan's blocks)
    const long nblk = (rows / rows_per_image) * dec_plan(rows_per_image, dec_bwd_min_upb(rows, rows_per_image, mlp)).bpi;
    return nblk * (mlp == 64 ? PL<64>::SIZE : PL<32>::SIZE) * 4;
}

// dw1 == NULL: only the data-gradient launch runs and the per-workgroup partials stay in `workspace` -- the caller sums the
// partials of all layers of a decoder stack with ONE dh_decoder_stack_bwd_finalize launch (32 finalize launches of 5.6 us sat on
// the critical path of the DAHiTra step).
// dx: [rows][32] bf16.  Shared-parameter gradients are ACCUMULATED into dw1 [mlp][32], dw2 [32][mlp], db1, db2,
// dbo, dln1_g/b, dln2_g/b (fp32); per-image dkq [images][32][32], dvoT [images][32][32] are assigned.
extern "C" int dh_decoder_layer_bwd(const void* x, const void* dy, void* dx, const void* kq, const void* voT,
                                    const void* vo, const void* kqT, const float* ln1_g, const float* ln1_b,
                                    const float* bo, const float* ln2_g, const float* ln2_b, const void* w1,
                                    const void* w1T, const float* b1, const void* w2, const void* w2T, const float* b2,
                                    float* dw1, float* dw2, float* db1, float* db2, float* dbo, float* dln1_g,
                                    float* dln1_b, float* dln2_g, float* dln2_b, float* dkq, float* dvoT, long rows,
                                    int rows_per_image, int mlp, float eps, void* workspace, void* stream) {
    if (check_common(rows, rows_per_image, mlp)) return 1;
    DecArgs a = {};
    a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.y = (bf16*)dx;
    a.kq = (const bf16*)kq; a.voT = (const bf16*)voT; a.vo = (const bf16*)vo; a.kqT = (const bf16*)kqT;
    a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.w1T = (const bf16*)w1T; a.w2T = (const bf16*)w2T;
    a.g1 = ln1_g; a.be1 = ln1_b; a.bo = bo; a.g2 = ln2_g; a.be2 = ln2_b; a.fb1 = b1; a.fb2 = b2;
    a.partial = reinterpret_cast<float*>(workspace);
    a.rows_per_image = rows_per_image; a.rows = rows; a.eps = eps;
    const DecPlan pl = dec_bwd_default(rows, rows_per_image, mlp);
    a.upb = pl.upb; a.bpi = pl.bpi;
    const int images = (int)(rows / rows_per_image);
    const int bpi = pl.bpi, nblk = images * bpi;
    if (g_db.on && !dw1) return dec_batch_record(&g_db.b[mlp == 64], &g_db.nb[mlp == 64], a, nblk, ST(stream));
    dec_remember_plan(a.partial, pl);
    static bool attr64 = false, attr32 = false;
    if (mlp == 64) {
        const size_t lds = bwd_lds_bytes<64>();
        if (!attr64) {
            attr64 = true;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                DH_FAIL("decoder_layer_bwd: cannot raise dynamic LDS to %zu", lds);
            }
        }
        hipLaunchKernelGGL(dec_bwd_kernel<64>, dim3(nblk), dim3(256), lds, ST(stream), a);
        if (dw1) {
            const FinArgs fa = {a.partial, nblk, bpi, dw1, dw2, db1, db2, dbo, dln1_g, dln1_b, dln2_g, dln2_b, dkq, dvoT, 0L, 0L, 0L};
            const int gx = dh_cdiv(PL<64>::KQ, 32);
            hipLaunchKernelGGL(dec_bwd_finalize_kernel<64>, dim3(fin_blocks(gx, images, 1)), dim3(256), 0, ST(stream), fa, gx, images, 1);
        }
    } else {
        const size_t lds = bwd_lds_bytes<32>();
        if (!attr32) {
            attr32 = true;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_bwd_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                DH_FAIL("decoder_layer_bwd: cannot raise dynamic LDS to %zu", lds);
            }
        }
        hipLaunchKernelGGL(dec_bwd_kernel<32>, dim3(nblk), dim3(256), lds, ST(stream), a);
        if (dw1) {
            const FinArgs fa = {a.partial, nblk, bpi, dw1, dw2, db1, db2, dbo, dln1_g, dln1_b, dln2_g, dln2_b, dkq, dvoT, 0L, 0L, 0L};
            const int gx = dh_cdiv(PL<32>::KQ, 32);
            hipLaunchKernelGGL(dec_bwd_finalize_kernel<32>, dim3(fin_blocks(gx, images, 1)), dim3(256), 0, ST(stream), fa, gx, images, 1);
        }
    }
    DH_CHECK_LAUNCH("decoder_layer_bwd");
    return 0;
}

// A whole decoder stack (`depth` layers of the same shapes, one set of tokens) in ONE launch per direction: see DecArgs::depth.
// x [rows][32] bf16 = the stack's input; ys [depth][rows][32] bf16 receives every layer's output (ys[depth - 1] = the stack's
// output; the backward recomputes each layer from its input).  Operands of layer l: kq / voT (/ vo / kqT) at + l * kq_lstride
// elements (the stacked dh_xattn_prep_fwd_stack outputs: images * 32 * 32), the packed MLP weights at + l * w_lstride
// elements, the fp32 parameter vectors at + l * par_lstride floats (the net's flat arena).  Same arithmetic as `depth` calls of
// dh_decoder_layer_fwd / _bwd, bit for bit.  Both join an open decoder batch (dh_decoder_batch_begin).
extern "C" int dh_decoder_stack_fwd(const void* x, void* ys, const void* kq, const void* voT, const float* ln1_g, const float* ln1_b,
                                    const float* bo, const float* ln2_g, const float* ln2_b, const void* w1, const float* b1,
                                    const void* w2, const float* b2, int depth, long kq_lstride, long w_lstride, long par_lstride,
                                    long rows, int rows_per_image, int mlp, float eps, void* stream) {
    if (check_common(rows, rows_per_image, mlp)) return 1;
    DH_REQUIRE(depth >= 1 && depth <= DEC_MAXDEPTH && x && ys, "decoder_stack_fwd: bad arguments (depth %d, 1 .. %d supported)", depth, DEC_MAXDEPTH);
    DecArgs a = {};
    a.x = (const bf16*)x; a.ys = (bf16*)ys; a.y = (bf16*)ys; a.kq = (const bf16*)kq; a.voT = (const bf16*)voT;
    a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2;
    a.g1 = ln1_g; a.be1 = ln1_b; a.bo = bo; a.g2 = ln2_g; a.be2 = ln2_b; a.fb1 = b1; a.fb2 = b2;
    a.rows_per_image = rows_per_image; a.rows = rows; a.eps = eps;
    a.depth = depth; a.act_ls = rows * D; a.kq_ls = kq_lstride; a.w_ls = w_lstride; a.par_ls = par_lstride;
    const DecPlan pl = dec_fwd_default(rows, rows_per_image);
    a.upb = pl.upb; a.bpi = pl.bpi;
    const int grid = (int)(rows / rows_per_image) * pl.bpi;
    if (g_db.on) return dec_batch_record(&g_db.f[2 + (mlp == 64)], &g_db.nf[2 + (mlp == 64)], a, grid, ST(stream));
    if (mlp == 64) hipLaunchKernelGGL((dec_fwd_kernel<64, true>), dim3(grid), dim3(256), 0, ST(stream), a);
    else hipLaunchKernelGGL((dec_fwd_kernel<32, true>), dim3(grid), dim3(256), 0, ST(stream), a);
    DH_CHECK_LAUNCH("decoder_stack_fwd");
    return 0;
}
// Data gradient of the stack: dy [rows][32] = gradient of ys[depth - 1], dx [rows][32] = gradient of x, dwork [rows][32] bf16
// scratch (the gradient between layers, rewritten in place); the per-workgroup partial sums of layer l land in workspace +
// l * dh_decoder_layer_bwd_workspace_size(rows, rows_per_image, mlp) bytes for dh_decoder_stack_bwd_finalize.
extern "C" int dh_decoder_stack_bwd(const void* x, const void* ys, const void* dy, void* dx, void* dwork, const void* kq,
                                    const void* voT, const void* vo, const void* kqT, const float* ln1_g, const float* ln1_b,
                                    const float* bo, const float* ln2_g, const float* ln2_b, const void* w1, const void* w1T,
                                    const float* b1, const void* w2, const void* w2T, const float* b2, int depth, long kq_lstride,
                                    long w_lstride, long par_lstride, long rows, int rows_per_image, int mlp, float eps,
                                    void* workspace, void* stream) {
    if (check_common(rows, rows_per_image, mlp)) return 1;
    DH_REQUIRE(depth >= 1 && depth <= DEC_MAXDEPTH && x && (ys || depth == 1) && dy && dx && (dwork || depth == 1) && workspace,
               "decoder_stack_bwd: bad arguments (depth %d, 1 .. %d supported)", depth, DEC_MAXDEPTH);
    DecArgs a = {};
    a.x = (const bf16*)x; a.ys = (bf16*)const_cast<void*>(ys); a.dy = (const bf16*)dy; a.y = (bf16*)dx; a.dwork = (bf16*)dwork;
    a.kq = (const bf16*)kq; a.voT = (const bf16*)voT; a.vo = (const bf16*)vo; a.kqT = (const bf16*)kqT;
    a.w1 = (const bf16*)w1; a.w2 = (const bf16*)w2; a.w1T = (const bf16*)w1T; a.w2T = (const bf16*)w2T;
    a.g1 = ln1_g; a.be1 = ln1_b; a.bo = bo; a.g2 = ln2_g; a.be2 = ln2_b; a.fb1 = b1; a.fb2 = b2;
    a.partial = reinterpret_cast<float*>(workspace);
    a.rows_per_image = rows_per_image; a.rows = rows; a.eps = eps;
    const DecPlan pl = dec_bwd_default(rows, rows_per_image, mlp);
    a.upb = pl.upb; a.bpi = pl.bpi;
    a.depth = depth; a.act_ls = rows * D; a.kq_ls = kq_lstride; a.w_ls = w_lstride; a.par_ls = par_lstride;
    a.part_ls = dh_decoder_layer_bwd_workspace_size(rows, rows_per_image, mlp) / 4;
    const int nblk = (int)(rows / rows_per_image) * pl.bpi;
    if (g_db.on) return dec_batch_record(&g_db.b[2 + (mlp == 64)], &g_db.nb[2 + (mlp == 64)], a, nblk, ST(stream));
    dec_remember_plan(a.partial, pl);
    static bool m32 = false, m64 = false;
    if (mlp == 64) {
        if (dec_set_bwd_lds<64>(reinterpret_cast<const void*>(dec_bwd_kernel<64, true>), m64)) return 1;
        hipLaunchKernelGGL((dec_bwd_kernel<64, true>), dim3(nblk), dim3(256), bwd_lds_bytes<64>(), ST(stream), a);
    } else {
        if (dec_set_bwd_lds<32>(reinterpret_cast<const void*>(dec_bwd_kernel<32, true>), m32)) return 1;
        hipLaunchKernelGGL((dec_bwd_kernel<32, true>), dim3(nblk), dim3(256), bwd_lds_bytes<32>(), ST(stream), a);
    }
    DH_CHECK_LAUNCH("decoder_stack_bwd");
    return 0;
}

// The finalize of `depth` layers at once: partials of layer l at workspace + l * dh_decoder_layer_bwd_workspace_size bytes,
// gradients of layer l at (pointer of layer 0) + l * grad_stride floats, dkq / dvoT of layer l at + l * images * 1024 floats.
extern "C" int dh_decoder_stack_bwd_finalize(const void* workspace, int depth, long rows, int rows_per_image, int mlp, float* dw1,
                                             float* dw2, float* db1, float* db2, float* dbo, float* dln1_g, float* dln1_b,
                                             float* dln2_g, float* dln2_b, long grad_stride, float* dkq, float* dvoT, void* stream) {
    if (check_common(rows, rows_per_image, mlp)) return 1;
    DH_REQUIRE(depth >= 1 && workspace && dw1 && dkq && dvoT, "decoder_stack_bwd_finalize: bad arguments (depth %d)", depth);
    const float* partial = reinterpret_cast<const float*>(workspace);
    const int images = (int)(rows / rows_per_image);
    const int bpi = dec_recall_plan(partial, rows, rows_per_image, mlp).bpi, nblk = images * bpi;
    const long pstride = dh_decoder_layer_bwd_workspace_size(rows, rows_per_image, mlp) / 4, kstride = (long)images * 1024;
    const FinArgs fa = {partial, nblk, bpi, dw1, dw2, db1, db2, dbo, dln1_g, dln1_b, dln2_g, dln2_b, dkq, dvoT, pstride, grad_stride, kstride};
    if (g_db.on) {          // recorded: issued with the other stacks' finalizes, after the recorded backward launches
        const int k = mlp == 64, gx = dh_cdiv(mlp == 64 ? PL<64>::KQ : PL<32>::KQ, 32);
        DH_REQUIRE(g_db.nfin[k] < DEC_MAXJ, "decoder_batch: more than %d stack finalizes recorded in one round", DEC_MAXJ);
        FinMulti& fm = g_db.fin[k];
        int& n = g_db.nfin[k];
        if (n == 0) fm.first[0] = 0;
        fm.a[n] = fa; fm.gx[n] = gx; fm.gy[n] = 1 + images; fm.depth[n] = depth;
        fm.first[n + 1] = fm.first[n] + fin_blocks(gx, images, depth);
        ++n;
        return 0;
    }
    if (mlp == 64)
        hipLaunchKernelGGL(dec_bwd_finalize_kernel<64>, dim3(fin_blocks(dh_cdiv(PL<64>::KQ, 32), images, depth)), dim3(256), 0, ST(stream), fa,
                           dh_cdiv(PL<64>::KQ, 32), images, depth);
    else
        hipLaunchKernelGGL(dec_bwd_finalize_kernel<32>, dim3(fin_blocks(dh_cdiv(PL<32>::KQ, 32), images, depth)), dim3(256), 0, ST(stream), fa,
                           dh_cdiv(PL<32>::KQ, 32), images, depth);
    DH_CHECK_LAUNCH("decoder_stack_bwd_finalize");
    return 0;
}
