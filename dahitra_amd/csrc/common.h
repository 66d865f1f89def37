// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of dahitra_amd.
// Activations are NHWC ("pixels x channels", channels contiguous) in T = float (parity mode)
// or bf16 (throughput mode); accumulation, statistics, master weights and weight gradients
// are always fp32.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct bf16 { unsigned short x; };

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define DH_DTYPE_F32 0
#define DH_DTYPE_BF16 1

// Split-bf16 forms of the fp32 mode (dh_set_f32_mma_mode): tensors stay fp32 in memory (T = f32x3 / f32x6 have float's size and
// layout), but the matrix products of an fp32 launch run on the bf16 matrix cores.  While a tile is staged every operand element
// x becomes NPL bf16 planes in LDS -- x = p0 + p1 (+ p2), p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1): 8 more
// mantissa bits per plane, the remainders are exact in fp32 -- and a product a * b is the sum, in the fp32 accumulator, of the
// v_mfma_f32_16x16x32_bf16 products a_i * b_j with i + j < NPL:
//   f32x3  NPL = 2: a0 b0 + a1 b0 + a0 b1               unit roundoff ~2^-17   3/16 of the exact fp32 MFMA's cycles
//   f32x6  NPL = 3: + a2 b0 + a1 b1 + a0 b2             unit roundoff ~2^-23   6/16
// (fp32: 2^-24).  In the "bf16x3" compute mode f32x3 (mode 1) serves the data and weight gradients; the FORWARD products run by
// default on the fp16-plane form f32h3 below (mode 3, ~2^-21, operands inside fp16's range) -- the activation error is what the
// gradients are sensitive to (measured: forward at 2^-17 puts the gradients 150x above their fp32 distance from the oracle,
// backward at 2^-17 leaves them at it) -- and f32x6 (mode 2, 2^-23, fp32's range) is the forward's fallback: DAHITRA_X3_FWD=2.
struct f32x3 { float v; };
struct f32x6 { float v; };
// f32h3: the two-plane / three-product form on FP16 planes (v_mfma_f32_16x16x32_f16): hi = fp16(x), lo = fp16(x - hi) carry 11
// mantissa bits each, so three products reach ~2^-21 where bf16 planes reach 2^-17 -- for operands inside fp16's RANGE (|x| <
// 65504; values below ~1e-4 lose the lo plane to the subnormals).  That rules it out for gradients, and makes it the form of the
// FORWARD products: activations are O(1), and the weight tile is staged times 2^8 (exact; the accumulators are scaled back by
// 2^-8 before the epilogue) so that weights of size 1e-2 keep a normal lo plane.  |w| must stay below 255.
struct f32h3 { float v; };
constexpr float F32H3_WSCALE = 256.f;
template <typename T> struct Prec {
    static constexpr int NPL = 1;                      // LDS planes per staged operand tile
    static constexpr bool X3 = false;
    static constexpr int CK = 64 / (int)sizeof(T);     // channels of one 64-byte LDS row
};
template <> struct Prec<f32x3> {
    static constexpr int NPL = 2;
    static constexpr bool X3 = true;
    static constexpr int CK = 32;                       // one 64-byte bf16 row per plane
};
template <> struct Prec<f32x6> {
    static constexpr int NPL = 3;
    static constexpr bool X3 = true;
    static constexpr int CK = 32;
};
template <> struct Prec<f32h3> {
    static constexpr int NPL = 2;
    static constexpr bool X3 = true;
    static constexpr int CK = 32;
};

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 converts two values per instruction (v_cvt_pk_bf16_f32).  The former
// software rounding cost ~7 VALU + an exec-mask branch (NaN case) per value in every bf16 store of every kernel.
typedef __attribute__((ext_vector_type(2))) __bf16 dh_bf16x2;
typedef __attribute__((ext_vector_type(2))) float dh_f32x2;
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) {
    const dh_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dh_bf16x2));
}
__device__ __forceinline__ unsigned short f2bf(float f) { return (unsigned short)(f2bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return bf2f(p->x); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { p->x = f2bf(v); }

// 4 consecutive elements (16 B fp32 / 8 B bf16); address must be aligned to that size.
__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void st4(bf16* p, const float (&o)[4]) {
    uint2 v;
    v.x = f2bf2(o[0], o[1]);
    v.y = f2bf2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ float ldf(const f32x3* p) { return p->v; }
__device__ __forceinline__ void stf(f32x3* p, float v) { p->v = v; }
__device__ __forceinline__ void ld4(const f32x3* p, float (&o)[4]) { ld4(reinterpret_cast<const float*>(p), o); }
__device__ __forceinline__ void st4(f32x3* p, const float (&o)[4]) { st4(reinterpret_cast<float*>(p), o); }
__device__ __forceinline__ float ldf(const f32h3* p) { return p->v; }
__device__ __forceinline__ void stf(f32h3* p, float v) { p->v = v; }
__device__ __forceinline__ void ld4(const f32h3* p, float (&o)[4]) { ld4(reinterpret_cast<const float*>(p), o); }
__device__ __forceinline__ void st4(f32h3* p, const float (&o)[4]) { st4(reinterpret_cast<float*>(p), o); }
__device__ __forceinline__ float ldf(const f32x6* p) { return p->v; }
__device__ __forceinline__ void stf(f32x6* p, float v) { p->v = v; }
__device__ __forceinline__ void ld4(const f32x6* p, float (&o)[4]) { ld4(reinterpret_cast<const float*>(p), o); }
__device__ __forceinline__ void st4(f32x6* p, const float (&o)[4]) { st4(reinterpret_cast<float*>(p), o); }
// 8 consecutive elements for bf16 (16 B), 4 for fp32 (16 B): the 16-byte vector unit "V16" -- one dwordx4 per lane
// (measured on the BatchNorm passes: no faster than 8-byte lanes, both sit at ~3.5-4 TB/s of mixed read/write).
template <typename T> struct V16 { static constexpr int N = 16 / sizeof(T); };
__device__ __forceinline__ void ldv(const float* p, float (&o)[4]) { ld4(p, o); }
__device__ __forceinline__ void stv(float* p, const float (&o)[4]) { st4(p, o); }
__device__ __forceinline__ void ldv(const bf16* p, float (&o)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
    o[4] = __uint_as_float(v.z << 16); o[5] = __uint_as_float(v.z & 0xffff0000u);
    o[6] = __uint_as_float(v.w << 16); o[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ void stv(bf16* p, const float (&o)[8]) {
    uint4 v;
    v.x = f2bf2(o[0], o[1]);
    v.y = f2bf2(o[2], o[3]);
    v.z = f2bf2(o[4], o[5]);
    v.w = f2bf2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = v;
}

// one 16-byte piece <-> fp32 values (8 bf16 / 4 fp32)
__device__ __forceinline__ void unpack16(const uint4& u, float (&o)[8]) {
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xffff0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xffff0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xffff0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack16(const uint4& u, float (&o)[4]) {
    o[0] = __uint_as_float(u.x); o[1] = __uint_as_float(u.y); o[2] = __uint_as_float(u.z); o[3] = __uint_as_float(u.w);
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float (&o)[16 / sizeof(T)]);
template <> __device__ __forceinline__ uint4 pack16<bf16>(const float (&o)[8]) {
    return make_uint4(f2bf2(o[0], o[1]), f2bf2(o[2], o[3]), f2bf2(o[4], o[5]), f2bf2(o[6], o[7]));
}
template <> __device__ __forceinline__ uint4 pack16<float>(const float (&o)[4]) {
    return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
}

template <> __device__ __forceinline__ uint4 pack16<f32x3>(const float (&o)[4]) { return pack16<float>(o); }
template <> __device__ __forceinline__ uint4 pack16<f32x6>(const float (&o)[4]) { return pack16<float>(o); }
template <> __device__ __forceinline__ uint4 pack16<f32h3>(const float (&o)[4]) { return pack16<float>(o); }
// fp16 planes of 8 fp32 values (f32h3): hi = fp16(x) and lo = fp16(x - hi), both round to nearest even
typedef __attribute__((ext_vector_type(2))) _Float16 dh_f16x2;
__device__ __forceinline__ unsigned f2h2(float lo, float hi) {
    const dh_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, dh_f16x2));
}
__device__ __forceinline__ void split_f16_planes(const float (&x)[8], uint4 (&pl)[2]) {
    unsigned h[4];
    float r[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = f2h2(x[2 * j], x[2 * j + 1]);
        const dh_f32x2 back = __builtin_convertvector(__builtin_bit_cast(dh_f16x2, h[j]), dh_f32x2);
        r[2 * j] = x[2 * j] - back[0];
        r[2 * j + 1] = x[2 * j + 1] - back[1];
    }
    pl[0] = make_uint4(h[0], h[1], h[2], h[3]);
    pl[1] = make_uint4(f2h2(r[0], r[1]), f2h2(r[2], r[3]), f2h2(r[4], r[5]), f2h2(r[6], r[7]));
}
// the NPL bf16 planes of 8 fp32 values (see f32x3 / f32x6 above): one 16-byte piece per plane
template <int NPL> __device__ __forceinline__ void split_bf16_planes(const float (&x)[8], uint4 (&pl)[NPL]) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j];
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        pl[p] = pack16<bf16>(r);
        if (p + 1 < NPL) {
            float h[8];
            unpack16(pl[p], h);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] -= h[j];
        }
    }
}
template <int NPL> __device__ __forceinline__ void split_bf16_planes(const uint4& a, const uint4& b, uint4 (&pl)[NPL]) {
    const float x[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                        __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
    split_bf16_planes<NPL>(x, pl);
}
// bf16x3 operand split of 8 fp32 values: hi = bf16(x) (round to nearest even), lo = bf16(x - hi) -- one 16-byte piece per
// plane (x - hi is exact in fp32; 24 VALU instructions per 8 values)
__device__ __forceinline__ void split_bf16x3(const float (&x)[8], uint4& hi, uint4& lo) {
    hi = pack16<bf16>(x);
    float h[8], r[8];
    unpack16(hi, h);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j] - h[j];
    lo = pack16<bf16>(r);
}
__device__ __forceinline__ void split_bf16x3(const uint4& a, const uint4& b, uint4& hi, uint4& lo) {
    const float x[8] = {__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w),
                        __uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
    split_bf16x3(x, hi, lo);
}

// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane of the row: four row-rotate adds on
// the vector ALU instead of four ds_bpermute round trips through the LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

#define DH_ACT_NONE 0
#define DH_ACT_RELU 1
#define DH_ACT_GELU 2

// ---- host side -------------------------------------------------------------------------------
#include <stdio.h>
#include <string.h>
extern "C" void dh_set_error(const char* msg);
#define DH_FAIL(...)                                                     \
    do {                                                                 \
        char _b[512];                                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                           \
        dh_set_error(_b);                                                \
        return 1;                                                        \
    } while (0)
#define DH_REQUIRE(cond, ...)           \
    do {                                \
        if (!(cond)) DH_FAIL(__VA_ARGS__); \
    } while (0)
#define DH_CHECK_LAUNCH(name)                                                      \
    do {                                                                           \
        hipError_t _e = hipGetLastError();                                         \
        if (_e != hipSuccess) DH_FAIL("%s launch: %s", name, hipGetErrorString(_e)); \
    } while (0)
static inline int dh_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out[i] (+)= scale * sum_t partial[t][i], t < nt, for the 32 outputs of (virtual) workgroup `bx`: 256 threads = 8 row phases x
// 32 consecutive outputs (coalesced rows), fp64 accumulation in a fixed order.  reduce_partials_kernel (norm.hip) and the
// batched tokenizer backward (tokens.hip) share it, so both give the same bits.
__device__ __forceinline__ void dh_reduce_partials_body(const float* __restrict__ partial, long nt, long n, float scale,
                                                        float* __restrict__ out, int accumulate, int bx, double (*red)[32]) {
    const int lane = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const long i = (long)bx * 32 + lane;
    double s = 0.0;
    if (i < n) {
        // four independent partial sums: the row loop is otherwise one chain of dependent-latency loads
        double s1 = 0.0, s2 = 0.0, s3 = 0.0;
        long t = ph;
#pragma unroll 4
        for (; t + 24 < nt; t += 32) {      // (16 loads in flight; a single output -- the loss -- is 8 lanes x 128 rows: 10 us at four)
            s += (double)partial[t * n + i];
            s1 += (double)partial[(t + 8) * n + i];
            s2 += (double)partial[(t + 16) * n + i];
            s3 += (double)partial[(t + 24) * n + i];
        }
        for (; t < nt; t += 8) s += (double)partial[t * n + i];
        s = (s + s1) + (s2 + s3);
    }
    red[ph][lane] = s;
    __syncthreads();
    if (ph == 0 && i < n) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][lane];
        const float v = (float)(t * scale);
        if (accumulate) out[i] += v; else out[i] = v;
    }
}


// fp64 sum over a workgroup of whole wavefronts (<= 1024 threads); every thread must call it; the result is valid
// in thread 0.  `sh` is a __shared__ double[16].
__device__ __forceinline__ double dh_block_sum_f64(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}
