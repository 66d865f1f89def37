// The class head's forward: 3x3 / pad 1 convolution 32 -> n_class <= 2 channels at full resolution, fp32 NCHW logits
// (models/help_funcs.py:13-14 `nn.Conv2d(32, out, 3, padding=1)` behind BatchNorm + ReLU; networks.py:1247 / 1355 `classifier`).
//
// As a tile convolution it pads 2 output channels to 16, stages a haloed input tile and runs nine taps of MFMAs of which 7/8
// are padding; with BatchNorm + ReLU applied in LDS it ran at 2 TB/s (66 us for the 134 MB it reads at the bench size).  Here
// the contraction over the 32 INPUT channels is done once per input pixel for all nine taps:
//   P[pixel][tap][class] = sum_ci W[class][ci][tap] * h[pixel][ci]          (M = (tap, class) = 18, N = 16 pixels, K = 32)
// -- the B operand of that MFMA (k = 8 g + e, n = pixel) is what a lane gets from ONE 16-byte load of its pixel (channels 8 g ..
// 8 g + 7), BatchNorm + ReLU applied to those eight values in registers, no halo, no staging of the input -- and the convolution
// is the nine-term gather  out[y][x] = sum_(kh, kw) P[y + kh - 1][x + kw - 1][(kh, kw)]  from a ring of four P rows in LDS
// (18 floats per pixel; a zero pixel either side, zero rows outside the image).  A workgroup walks a strip of rows of one image:
// P of row y + 1, one barrier, the gather of row y (coalesced stores of both class planes).
#include "common.h"

namespace {

constexpr int HF_KP = 18;        // floats per pixel of a P row: index tap * 2 + class.  Pitch 18 dwords: the 16 lanes of a
                                 // ds_write_b64 group and the 32 of a ds_read_b64 group fall on distinct bank pairs
constexpr int HF_SLOTS = 4;      // rows y - 1, y, y + 1 are read while y + 2 is written: one barrier per row

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 hf_f16x8;
constexpr int HF_GPW = 5;       // 16-pixel groups of a row per wave: W <= 4 * 5 * 16 = 320 (the LDS rule allows 282)

// T = float: the fp32 pipeline in its split-product mode 3 (dh_set_f32_mma_mode: the forward's form) -- x fp32, every product as
// three v_mfma_f32_16x16x32_f16 on fp16 planes (x = hi + lo; the weights' planes of W * 2^8, the sums scaled back), ~2^-21.
template <typename T, bool INBN>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w_oihw,
                                                       const float* __restrict__ bias, int NC, const float* __restrict__ in_scale,
                                                       const float* __restrict__ in_shift, int in_groups, float* __restrict__ out,
                                                       int N, int H, int W, int rpb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int RS = (W + 2) * HF_KP;                          // floats per slot
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    const int bpi = (H + rpb - 1) / rpb, n = blockIdx.x / bpi, r0 = (blockIdx.x - n * bpi) * rpb, r1 = min(r0 + rpb, H);
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int PL = F32 ? 2 : 1, PXB = 32 * (int)sizeof(T);
    // A fragments (planes): row m = (tap, class) of the weights, k = ci = 8 g + e.  a0: m = pl (taps 0 .. 7), a1: m = 16 + pl (tap 8)
    uint4 a0[PL], a1[PL];
    {
        float v0[8], v1[8];
        const int tap = pl >> 1, co = pl & 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = 8 * g + e;
            v0[e] = co < NC ? w_oihw[((size_t)co * 32 + ci) * 9 + tap] : 0.f;
            v1[e] = (pl < 2 && pl < NC) ? w_oihw[((size_t)pl * 32 + ci) * 9 + 8] : 0.f;
            if constexpr (F32) { v0[e] *= F32H3_WSCALE; v1[e] *= F32H3_WSCALE; }
        }
        if constexpr (F32) { split_f16_planes(v0, a0); split_f16_planes(v1, a1); }
        else { a0[0] = pack16<bf16>(v0); a1[0] = pack16<bf16>(v1); }
    }
    auto mma = [](const uint4& a, const uint4& b, f32x4 c) {
        if constexpr (F32) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hf_f16x8, a), __builtin_bit_cast(hf_f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
    };
    float ms[8], mh[8];
    if constexpr (INBN) {
        const int grp = n / (N / in_groups);
#pragma unroll
        for (int e = 0; e < 8; ++e) { ms[e] = in_scale[grp * 32 + 8 * g + e]; mh[e] = in_shift[grp * 32 + 8 * g + e]; }
    }
    // the zero pixels at either end of every slot
    for (int i = tid; i < HF_SLOTS * 2 * HF_KP; i += 256) {
        const int s = i / (2 * HF_KP), r = i - s * 2 * HF_KP;
        smem[s * RS + (r < HF_KP ? r : (W + 1) * HF_KP + r - HF_KP)] = 0.f;
    }
    const int ngroups = (W + 15) >> 4;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)((unsigned)((long)N * H * W) * (unsigned)PXB), 0x00020000);
    // P of image row yy into slot yy & 3 (zeros for a row outside the image).  The waves take the row's 16-pixel groups in turn,
    // HF_GPW per wave; a row's loads are REQUESTED one row ahead of their use (a strip is a chain of dependent rows at two
    // workgroups per CU: with the loads inside the row's own phase every row paid two HBM round trips, 65 us per launch)
    struct RowRegs { u32x4 r[HF_GPW][PL]; };                  // (fp32: a lane's eight values are two pieces)
    auto request = [&](int yy, RowRegs& q) {
        const bool rowok = (yy >= 0) & (yy < H);
        const unsigned rowoff = (unsigned)(((long)n * H + (rowok ? yy : 0)) * W) * (unsigned)PXB;
#pragma unroll
        for (int k = 0; k < HF_GPW; ++k) {
            const int xx = (wv + 4 * k) * 16 + pl;
            const unsigned off = (rowok & (xx < W)) ? rowoff + (unsigned)xx * (unsigned)PXB + g * (unsigned)(PXB / 4) : 0x80000000u;
            q.r[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
            if constexpr (F32) q.r[k][PL - 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off + 16u, 0, 0);
        }
    };
    auto p_row = [&](int yy, const RowRegs& q) {
        float* slot = smem + (yy & (HF_SLOTS - 1)) * RS + HF_KP;
        if (yy < 0 || yy >= H) {
            for (int i = tid; i < W * HF_KP; i += 256) slot[i] = 0.f;
            return;
        }
#pragma unroll
        for (int k = 0; k < HF_GPW; ++k) {
            const int xx = (wv + 4 * k) * 16 + pl;
            if ((wv + 4 * k) * 16 >= W) break;                 // (uniform)
            uint4 b[PL];
            f32x4 d0 = f32x4{0.f, 0.f, 0.f, 0.f}, d1 = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (F32) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(q.r[k][0][e]); v[4 + e] = __uint_as_float(q.r[k][PL - 1][e]); }
                if constexpr (INBN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * ms[e] + mh[e], 0.f);
                }
                split_f16_planes(v, b);
                d0 = mma(a0[0], b[0], d0); d0 = mma(a0[0], b[PL - 1], d0); d0 = mma(a0[PL - 1], b[0], d0);
                d1 = mma(a1[0], b[0], d1); d1 = mma(a1[0], b[PL - 1], d1); d1 = mma(a1[PL - 1], b[0], d1);
#pragma unroll
                for (int j = 0; j < 4; ++j) { d0[j] *= 1.f / F32H3_WSCALE; d1[j] *= 1.f / F32H3_WSCALE; }
            } else {
                if constexpr (INBN) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] = fmaxf(__uint_as_float(q.r[k][0][e] << 16) * ms[2 * e] + mh[2 * e], 0.f);
                        v[2 * e + 1] = fmaxf(__uint_as_float(q.r[k][0][e] & 0xffff0000u) * ms[2 * e + 1] + mh[2 * e + 1], 0.f);
                    }
                    b[0] = pack16<bf16>(v);
                } else {
                    b[0] = make_uint4(q.r[k][0][0], q.r[k][0][1], q.r[k][0][2], q.r[k][0][3]);
                }
                d0 = mma(a0[0], b[0], d0);
                d1 = mma(a1[0], b[0], d1);
            }
            if (xx < W) {
                float* p = slot + xx * HF_KP;
                *reinterpret_cast<float2*>(p + 4 * g) = make_float2(d0[0], d0[1]);
                *reinterpret_cast<float2*>(p + 4 * g + 2) = make_float2(d0[2], d0[3]);
                if (g == 0) *reinterpret_cast<float2*>(p + 16) = make_float2(d1[0], d1[1]);
            }
        }
    };
    const float b0 = bias ? bias[0] : 0.f, b1 = (bias && NC > 1) ? bias[1] : 0.f;
    auto out_row = [&](int y) {
        const float* s0 = smem + ((y - 1) & (HF_SLOTS - 1)) * RS;
        const float* s1 = smem + (y & (HF_SLOTS - 1)) * RS;
        const float* s2 = smem + ((y + 1) & (HF_SLOTS - 1)) * RS;
        for (int xx = tid; xx < W; xx += 256) {
            float acc0 = b0, acc1 = b1;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float* s = kh == 0 ? s0 : (kh == 1 ? s1 : s2);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float2 v = *reinterpret_cast<const float2*>(s + (xx + kw) * HF_KP + (kh * 3 + kw) * 2);
                    acc0 += v.x; acc1 += v.y;
                }
            }
            out[(((long)n * NC) * H + y) * W + xx] = acc0;
            if (NC > 1) out[(((long)n * NC + 1) * H + y) * W + xx] = acc1;
        }
    };
    RowRegs qa, qb;
    request(r0 - 1, qa);
    request(r0, qb);
    p_row(r0 - 1, qa);
    request(r0 + 1, qa);
    p_row(r0, qb);
    for (int y = r0; y < r1; y += 2) {                       // (rows in pairs: the two register sets swap roles without moves)
        request(y + 2, qb);
        p_row(y + 1, qa);
        __syncthreads();
        out_row(y);
        if (y + 1 >= r1) break;
        request(y + 3, qa);
        p_row(y + 2, qb);
        __syncthreads();
        out_row(y + 1);
    }
}

}  // namespace

// x [N][H][W][32] bf16, or fp32 (dtype DH_DTYPE_F32: ONLY under dh_set_f32_mma_mode(3), whose arithmetic this is) (pre-BatchNorm when in_scale / in_shift [in_groups][32] are given: relu(x * scale + shift) is the head's
// input, rounded to bf16 as every BatchNorm-on-load consumer sees it); w_oihw [n_class][32][3][3] fp32 master weights;
// bias [n_class] or NULL; logits_nchw [N][n_class][H][W] fp32.  Returns DH_CONV_NO_FIT-style -2 (nothing launched, no error text)
// when the shape is not this kernel's: n_class > 2, or four P rows of W + 2 pixels do not fit a workgroup's LDS share.
extern "C" int dh_head_fwd_supported(int NC, int W) { return NC >= 1 && NC <= 2 && (long)HF_SLOTS * (W + 2) * HF_KP * 4 <= 80 * 1024; }
extern "C" int dh_head_fwd(int dtype, const void* x, const float* w_oihw, const float* bias, int NC, const float* in_scale,
                           const float* in_shift, int in_groups, float* logits_nchw, int N, int H, int W, void* stream) {
    DH_REQUIRE(x && w_oihw && logits_nchw && N > 0 && H > 0 && W > 0, "head_fwd: bad arguments");
    DH_REQUIRE(dh_head_fwd_supported(NC, W), "head_fwd: n_class=%d W=%d is not this kernel's shape (dh_head_fwd_supported)", NC, W);
    DH_REQUIRE((long)N * H * W * 128 < (1L << 31), "head_fwd: %d x %d x %d pixels x 128 bytes do not fit a 2 GiB buffer descriptor", N, H, W);
    if (in_scale) DH_REQUIRE(in_shift && in_groups > 0 && N % in_groups == 0, "head_fwd: BatchNorm-on-load needs in_shift and N %% in_groups == 0");
    // rows per workgroup: >= 512 workgroups (two per CU), each pays two extra P rows
    int rpb = H;
    while (rpb > 8 && (long)N * ((H + rpb / 2 - 1) / (rpb / 2)) <= 512) rpb /= 2;
    const int grid = N * ((H + rpb - 1) / rpb);
    const size_t lds = (size_t)HF_SLOTS * (W + 2) * HF_KP * 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static bool attr_done = false;
    if (!attr_done) {
        DH_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_kernel<bf16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_kernel<bf16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_kernel<float, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                   hipFuncSetAttribute(reinterpret_cast<const void*>(head_fwd_kernel<float, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess,
                   "head_fwd: 80 KiB of dynamic LDS refused");
        attr_done = true;
    }
#define HF_LAUNCH(T, BN) hipLaunchKernelGGL((head_fwd_kernel<T, BN>), dim3(grid), dim3(256), lds, st, (const T*)x, w_oihw, bias, NC, in_scale, \
                                            in_shift, BN ? in_groups : 1, logits_nchw, N, H, W, rpb)
    if (dtype == DH_DTYPE_BF16) { if (in_scale) HF_LAUNCH(bf16, true); else HF_LAUNCH(bf16, false); }
    else { if (in_scale) HF_LAUNCH(float, true); else HF_LAUNCH(float, false); }
#undef HF_LAUNCH
    DH_CHECK_LAUNCH("head_fwd");
    return 0;
}
