// NHWC direct convolution on the CDNA4 matrix cores (gfx950), im2col-free.
//
// One 256-thread workgroup (4 wavefronts of 64) produces an 8x16-pixel output tile of one image
// for NT output channels.  Per 64-byte input-channel chunk (32 bf16 / 16 fp32 channels) the
// haloed input tile and the weights of all KSxKS taps are staged once in LDS; every tap is then a
// shifted 1x1 product D[cout][pixel] += W[cout][k] * X[pixel][k] issued as MFMA
//   bf16: v_mfma_f32_16x16x32_bf16   (one per 32 channels)
//   fp32: v_mfma_f32_16x16x4_f32 x4  (exact fp32, parity mode)
// with fp32 accumulation.  Weights are the A operand so that a lane ends up with 4 consecutive
// output channels of one pixel (8/16-byte stores).  Epilogue: +bias, +residual, ReLU/GELU, and
// optional per-workgroup partial sums (sum, sum of squares) per channel for train-mode BatchNorm.
// A data-gradient launch can instead be "gated" for the BatchNorm layer it back-propagates into: the epilogue
// applies that layer's ReLU mask, stores g = dout*mask and emits the per-tile partials (sum g, sum g*xhat) of the
// BN backward -- the separate reduction pass over (dout, out, y) disappears and the apply pass reads 2 tensors
// instead of 3.
//
// The same kernel serves: every 3x3 / 1x1 convolution of the trunk and head (reference
// models/resnet.py:24-32, models/help_funcs.py:7-15, models/networks.py:215), their data
// gradients (transposed, flipped weights), every nn.Linear of the token / pixel transformers
// (models/help_funcs.py:52-63,86-88,111), and the per-image attention products QK^T / PV in their
// re-associated 32x32 form (per-image weights via w_nstride).
#pragma once
#include "common.h"
#include <type_traits>

namespace {

constexpr int TW = 16;                  // output tile: 16 pixels wide, 4*RW rows (RW rows per wavefront)
constexpr int DH_CONV_NO_FIT = -2;      // a split-bf16 launch whose staging planes exceed the LDS: nothing was launched
// LDS rows hold one 64-byte channel chunk.  ds_read_b128 is served in 16-lane groups
// {0-3,12-15,20-27},{4-11,16-19,28-31},... (MI355X_MICROARCH.md), each needing 16 distinct 16-byte slots
// mod 256 B.  Brute force over layouts: for consecutive rows (stride-1 pixels, weight rows) pitch 64 with
// slot ^= ((row>>2)&1)<<1 is conflict-free at every base offset; for every-other-row reads (stride 2) pitch
// 80 without swizzle is.  (The former 80-byte pitch cost 2x on stride 1: SQ_LDS_BANK_CONFLICT = 45 %.)
constexpr int WPITCH = 64;
template <int STRIDE> struct HaloLayout {
    static constexpr int PITCH = STRIDE == 1 ? 64 : 80;
    static __device__ __forceinline__ int off(int row, int q) {
        return row * PITCH + ((STRIDE == 1 ? (q ^ (((row >> 2) & 1) << 1)) : q) << 4);
    }
};
__device__ __forceinline__ int wt_off(int row, int q) { return row * WPITCH + ((q ^ (((row >> 2) & 1) << 1)) << 4); }

}  // namespace

// (external linkage: the per-dtype launchers of the other translation units take it by reference)
struct ConvArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    const void* res;
    float* stats;      // [2][CoutPad][gridDim.x] partial (sum, sumsq) per pixel tile, or null: channel-major, so that
                       // the per-channel combine (bn_finalize) reads contiguous runs
    void* y2;          // optional: value before the activation (needed by the GELU derivative)
    int N, H, W, Cin, OH, OW, Cout, CoutPad, pad, act;
    int npix;          // valid output pixels per image in linear order (OH*OW unless a row view)
    int in_npix;       // valid input pixels per image in linear order (H*W unless a row view)
    long w_nstride;    // elements between per-image weight sets (0: shared)
    int tilesX, tilesY;
    int rw;            // rows per wavefront (tile height = 4*rw)
    int dil;           // dilation (1, or 2 for the ResNet-50 layer3 3x3 convolutions)
    // BatchNorm-backward gating of a data-gradient launch (see the epilogue): tensors of the BN layer whose output
    // this launch differentiates -- same [N][OH][OW][Cout] shape as y
    const void* gate_out;       // that layer's post-ReLU output (null: no ReLU)
    const void* gate_y;         // its pre-normalisation input (null: gating off)
    const float* gate_mean;     // [groups][Cout]
    const float* gate_invstd;   // [groups][Cout]
    int gate_groups;
    // BatchNorm-apply + ReLU on LOAD (train mode): x is the PRE-normalisation output of the previous conv and the kernel
    // consumes relu(x * in_scale[g][c] + in_shift[g][c]) -- the separate bn_apply pass over that tensor (one read + one
    // write of the whole activation) disappears.  Zero padding applies to the post-activation tensor: out-of-image halo
    // pieces stay zero.
    const float* in_scale;      // [in_groups][Cin], or null: off
    const float* in_shift;
    int in_groups;
    // 2x2 PHASE convolutions (KS = 2 only): conv3x3(nearest-upsample-x2(x)) == four 2x2 convolutions on x, one per output
    // parity (a, b), with the 3x3 taps that fall on the same source pixel pre-summed (models/networks.py:251-256).
    //   1: forward.  Logical Cout = 4 * NT (one NT = 32 / 64 block per phase); block (a, b) reads rows oy + t + a - 1 and its
    //      pixel (oy, ox) is STORED at (2 oy + a, 2 ox + b) of the [N][2 OH][2 OW][NT] output (depth-to-space).  `res`, if
    //      given, is a COARSE [N][OH][OW][NT] tensor added to phase (0, 0) only.  The same mode serves the data gradient of
    //      a 3x3 STRIDE-2 convolution (output parity (a, b) receives 1, 2, 2 or 4 of the 9 taps; weights from
    //      dh_pack_s2_dgrad_phase_weights; `res` = the coarse data gradient of the block's 1x1 stride-2 shortcut).
    //   2: data gradient.  Logical Cin = 4 * 32: 32-channel group (a, b) is GATHERED from the [N][2 H][2 W][32] gradient at
    //      (2 (iy + 1 - a) + a, 2 (ix + 1 - b) + b) (space-to-depth with a per-phase shift), so that all four phases share
    //      the tap geometry of a 2x2 convolution with pad 1.
    int phase_mode;
    int no_xcd_remap;       // DAHITRA_NO_XCD_REMAP=1: plain (tile, channel block) = (blockIdx.x, blockIdx.y) order
    // class head (NT = 16 instantiations only): the Cout real channels (+ bias) are written as fp32 NCHW logits
    // [N][Cout][OH][OW] straight from the accumulators -- the reference's output layout -- and nothing goes to y
    float* y_nchw;
    // the same weights in FRAGMENT order [CoutPad / 16][Cin / 32][9][64][8] (dh_pack_weights_multi, dtype | 0x200), or null:
    // read by the register-resident-weights kernel (conv_wreg.hip) only
    const void* w_frag;
    // Data gradient THROUGH the bilinear x4 upsample in front of this convolution's input (compact epilogue, NT = 32, 8x16
    // tiles): the tile is not stored; it is reduced, in fp32, to the 4 x 6 coarse pixels its 8 x 16 fine pixels interpolate
    // from (align_corners = False: fine row o reads coarse rows floor((o + 0.5) / 4 - 0.5) and the next, clamped) and the
    // per-tile partial [tile][4][6][Cout] goes to up4_partial; dh_absdiff_up4_combine sums the <= 4 tiles of a coarse pixel.
    float* up4_partial;
    // Channel concatenation WITHOUT the concatenated tensor (conv_wreg.hip only; dh_conv3x3_split_fwd).  x_split != 0: the input
    // is cat([A, B], channel) of two [N][H][W][Cin / 2] tensors, A at x and B at x + x_split bytes (torch.cat([a_128, b_128], 1),
    // models/networks.py:1344: the two temporal streams are the two halves of ONE [2N] batch).  y_split != 0: the output's
    // channel halves go to two [N][OH][OW][Cout / 2] tensors, y and y + y_split bytes (the data gradient of such a layer).
    long x_split, y_split;
    // |A - B| + bilinear x4 ON LOAD (bf16, 3x3 / stride 1 / pad 1, 32 -> 32 channels: classifier.0 behind
    // nn.Upsample(4, 'bilinear') of abs(x1 - x2), models/networks.py:383-389): x is not read; the haloed input tile is
    // interpolated (align_corners = False) from the 4 x 6 coarse pixels of up4_a / up4_b [N][H / 4][W / 4][32] it depends on, with
    // the terms and their order of dh_absdiff_upsample4_fwd -- the 32 x H x W map (134 MB at the bench size) is never
    // written or read.  H, W stay the FINE sizes.
    const void* up4_a;
    const void* up4_b;
};

namespace {

union V16u {
    uint4 u;
    float f[4];
    s16x8 h;
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const V16u& a, const V16u& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[1], b.f[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[2], b.f[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[3], b.f[3], c, 0, 0, 0);
    }
};
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(const V16u& a, const V16u& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.h, b.h, c, 0, 0, 0);
    }
};
template <> struct Mma<f32x3> : Mma<bf16> {};         // one product of the three / six; the tap loop pairs the planes
template <> struct Mma<f32x6> : Mma<bf16> {};
typedef __attribute__((ext_vector_type(8))) _Float16 dh_f16x8;
template <> struct Mma<f32h3> {                          // fp16 planes (common.h f32h3)
    static __device__ __forceinline__ void run(const V16u& a, const V16u& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dh_f16x8, a.u), __builtin_bit_cast(dh_f16x8, b.u), c, 0, 0, 0);
    }
};
// the planes of 8 staged fp32 values, by form
template <typename T, int NPL> __device__ __forceinline__ void split_planes(const float (&v)[8], uint4 (&pl)[NPL]) {
    if constexpr (std::is_same<T, f32h3>::value) split_f16_planes(v, pl);
    else split_bf16_planes<NPL>(v, pl);
}

// source rows / columns and weight of a bilinear x4 destination index (align_corners = False; = bil_src of pointwise.hip)
__device__ __forceinline__ void up4_src(int d, int in, int& i0, int& i1, float& l) {
    float s = ((float)d + 0.5f) * 0.25f - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l = s - (float)i0;
}

// bf16x3 (T = f32x3, see common.h): the staging of a 32-channel chunk is TWO planes (hi | lo) of the bf16 layout -- more than
// half a CU's LDS for the 64-wide output tile, so one workgroup per CU and the whole register file
template <typename T> constexpr int conv_min_wgs() { return Prec<T>::NPL > 1 ? 1 : 2; }

template <typename T, int KS, int STRIDE, int NT, int RW, int DIL, bool PF, bool FAST, bool INBN = false, bool INUP4 = false>
__global__ __launch_bounds__(256, conv_min_wgs<T>()) void conv_mfma_kernel(ConvArgs p) {
    constexpr int TH = 4 * RW;
    constexpr int HH = (TH - 1) * STRIDE + (KS - 1) * DIL + 1;
    constexpr int HWD = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    constexpr int TAPS = KS * KS;
    constexpr int NPL = Prec<T>::NPL;           // LDS planes (split-bf16 forms: 2 or 3, common.h)
    constexpr bool X3 = NPL > 1;
    constexpr int CK = Prec<T>::CK;             // channels per chunk: one 64-byte LDS row (per plane)
    constexpr int LV = X3 ? 2 : 1;              // 16-byte global loads per staged 16-byte LDS piece (bf16x3: 8 fp32 -> 8 hi + 8 lo)
    constexpr int NS = NT / 16;                 // 16-channel output sub-tiles
    static_assert(!(X3 && INUP4), "bilinear x4 on load is a bf16 form");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HL = HaloLayout<STRIDE>;
    constexpr int HB = HH * HWD * HL::PITCH, WB = TAPS * NT * WPITCH;      // bytes per halo / weight plane
    unsigned char* halo = smem;                                  // [NPL][HH*HWD] rows, HL layout
    unsigned char* wts = smem + NPL * HB;                        // [NPL][TAPS*NT] rows, swizzled pitch 64
    float* bnp = reinterpret_cast<float*>(wts + NPL * WB);       // INBN: [2][Cin] scale | shift of this image's group
                                                                 // INUP4: [4][6][32] |A - B| of the tile's coarse footprint

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    // Workgroups are dispatched x-fastest and round-robin over the 8 XCDs (one L2 each).  Remapped so that an XCD walks a
    // contiguous range of pixel tiles and runs the output-channel blocks of one tile back to back: the input tile and the
    // halo columns shared with the neighbouring tile are then served by that XCD's L2 instead of being fetched once per
    // output-channel block (measured FETCH_SIZE 2.3x the input tensor on the 128/256-channel layers before).
    int tile = blockIdx.x, cb = blockIdx.y;
    if ((gridDim.x & 7) == 0 && !p.no_xcd_remap) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, s = lin >> 3;
        tile = (int)(xcd * (gridDim.x >> 3) + s / gridDim.y);
        cb = (int)(s % gridDim.y);
    }
    int bt = tile;
    const int tx = bt % p.tilesX; bt /= p.tilesX;
    const int ty = bt % p.tilesY;
    const int n = bt / p.tilesY;
    const int co0 = cb * NT;
    const int oy0 = ty * TH, ox0 = tx * TW;
    int pad_y = p.pad, pad_x = p.pad;
    if constexpr (KS == 2) {
        if (p.phase_mode == 1) { pad_y = p.pad - (cb >> 1); pad_x = p.pad - (cb & 1); }
    }
    const int iy0 = oy0 * STRIDE - pad_y, ix0 = ox0 * STRIDE - pad_x;

    const unsigned char* xin = reinterpret_cast<const unsigned char*>(p.x) +
                               (size_t)n * p.H * p.W * p.Cin * sizeof(T);
    const unsigned char* wgt = reinterpret_cast<const unsigned char*>(p.w) +
                               (size_t)n * p.w_nstride * sizeof(T);

    f32x4 acc[NS][RW];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline over the 64-byte channel chunks: the global loads of chunk c+1 are issued into
    // registers before the MFMAs of chunk c and committed to LDS after them (latency hides under matrix work).
    constexpr int NHV = (HH * HWD * 4 + 255) / 256, NWV = (TAPS * NT * 4 + 255) / 256;
    uint4 rh[NHV * LV], rw[NWV * LV];
    // Address generation is branch-free with 32-bit offsets from the (uniform) image / weight base: at two waves per
    // SIMD the prologue runs as one dependent instruction chain (~10 cycles per instruction), so its length is time.
    // An out-of-image halo piece loads offset 0 (always mapped) and is zeroed by a select.
    unsigned hoff[NHV];      // ~0u: piece outside the image (or past the tile)
    auto set_hoff = [&](int ph) {
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int idx = tid + i * 256, px = idx >> 2, q = idx & 3;
            const int hy = px / HWD, hx = px - hy * HWD;
            const int iy = iy0 + hy, ix = ix0 + hx, lin = iy * p.W + ix;
            if (KS == 2 && p.phase_mode == 2) {
                // phase (a, b) of the fine-grid gradient, shifted by (1 - a, 1 - b) coarse pixels; 32 channels per fine pixel
                const int a = ph >> 1, b = ph & 1, ci = iy + 1 - a, cj = ix + 1 - b;
                const bool ok = idx < HH * HWD * 4 && (unsigned)ci < (unsigned)p.H && (unsigned)cj < (unsigned)p.W;
                hoff[i] = ok ? (unsigned)(((2 * ci + a) * (2 * p.W) + 2 * cj + b) * 32) * (unsigned)sizeof(T) + q * (16 * LV) : ~0u;
            } else {
                const bool ok = idx < HH * HWD * 4 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && lin < p.in_npix;
                hoff[i] = ok ? (unsigned)(lin * p.Cin) * (unsigned)sizeof(T) + q * (16 * LV) : ~0u;
            }
        }
    };
    set_hoff(0);
    // weight piece i of a thread is row tid/4 + 64*i of the [TAPS*NT] staged rows: tap advances by 64/NT per piece,
    // the output channel stays -- one per-lane offset plus a uniform step
    static_assert(64 % NT == 0, "weight staging assumes NT divides 64");
    const int wrow = tid >> 2;
    const unsigned wrowb = (unsigned)p.Cin * (unsigned)sizeof(T);                          // bytes per staged weight row in memory
    const unsigned woff0 = (unsigned)((wrow / NT) * p.CoutPad + co0 + (wrow % NT)) * wrowb + (tid & 3) * (16 * LV);
    const unsigned wstep = (unsigned)((64 / NT) * p.CoutPad) * wrowb;
    auto fetch = [&](int c0) {
        const unsigned char* xb = xin + (size_t)c0 * sizeof(T);
        const unsigned char* wb = wgt + (size_t)c0 * sizeof(T);
        if constexpr (KS == 2) {
            if (p.phase_mode == 2) {       // chunk -> (phase, channel offset inside the phase's 32 channels)
                if (c0 > 0 && (c0 & 31) == 0) set_hoff(c0 >> 5);
                xb = xin + (size_t)(c0 & 31) * sizeof(T);
            }
        }
        if constexpr (!INUP4) {
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                const bool ok = hoff[i] != ~0u;
#pragma unroll
                for (int l = 0; l < LV; ++l) {
                    const uint4 v = *reinterpret_cast<const uint4*>(xb + (ok ? hoff[i] : 0u) + 16 * l);
                    rh[i * LV + l] = ok ? v : make_uint4(0, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const bool ok = tid + i * 256 < TAPS * NT * 4;          // only the last piece can be partial
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                const uint4 v = *reinterpret_cast<const uint4*>(wb + (size_t)i * wstep + (ok ? woff0 : 0u) + 16 * l);
                rw[i * LV + l] = ok ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto commit = [&](int c0) {
        if constexpr (X3) {
            // fp32 pieces -> (hi, lo) bf16 planes; BatchNorm + ReLU on load is applied to the fp32 values first
            float sc[8], sh[8];
            if constexpr (INBN) {
                const float* sp = bnp + c0 + (tid & 3) * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                    *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + p.Cin + j);
                }
            }
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                const int idx = tid + i * 256;
                if (idx >= HH * HWD * 4) continue;
                float v[8];
                unpack16(rh[2 * i], reinterpret_cast<float(&)[4]>(v[0]));
                unpack16(rh[2 * i + 1], reinterpret_cast<float(&)[4]>(v[4]));
                if constexpr (INBN) {
                    if (hoff[i] != ~0u) {               // padding of the post-activation tensor stays zero
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                    }
                }
                uint4 pls[NPL];
                split_planes<T, NPL>(v, pls);
#pragma unroll
                for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint4*>(halo + q * HB + HL::off(idx >> 2, idx & 3)) = pls[q];
            }
#pragma unroll
            for (int i = 0; i < NWV; ++i) {
                const int idx = tid + i * 256;
                if (idx >= TAPS * NT * 4) continue;
                uint4 pls[NPL];
                float wv_[8];
                unpack16(rw[2 * i], reinterpret_cast<float(&)[4]>(wv_[0]));
                unpack16(rw[2 * i + 1], reinterpret_cast<float(&)[4]>(wv_[4]));
                if constexpr (std::is_same<T, f32h3>::value) {          // weights times 2^8: a normal lo plane for |w| ~ 1e-2
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv_[j] *= F32H3_WSCALE;
                }
                split_planes<T, NPL>(wv_, pls);
#pragma unroll
                for (int q = 0; q < NPL; ++q) *reinterpret_cast<uint4*>(wts + q * WB + wt_off(idx >> 2, idx & 3)) = pls[q];
            }
            return;
        }
        if constexpr (INUP4) {
            // halo piece = 8 channels of one fine pixel: the four bilinear terms in the order of absdiff_up4_fwd_kernel
            // (rows (y0, y1) x columns (x0, x1); a clamped border index carries weight exactly 0), from the staged footprint
            constexpr int PC = 16 / (int)sizeof(T);
            const int CH = p.H >> 2, CW = p.W >> 2, cyb = (oy0 >> 2) - 1, cxb = (ox0 >> 2) - 1;
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                rh[i] = make_uint4(0, 0, 0, 0);
                if (hoff[i] == ~0u) continue;            // padding of the upsampled tensor
                const int idx = tid + i * 256, px = idx >> 2, q = idx & 3;
                const int hy = px / HWD, hx = px - hy * HWD;
                int y0, y1, x0, x1;
                float ly, lx;
                up4_src(iy0 + hy, CH, y0, y1, ly);
                up4_src(ix0 + hx, CW, x0, x1, lx);
                const float wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
                const int rr[2] = {y0 - cyb, y1 - cyb}, cc[2] = {x0 - cxb, x1 - cxb};
                float acc[PC];
#pragma unroll
                for (int j = 0; j < PC; ++j) acc[j] = 0.f;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                        const float* d = bnp + (rr[pp] * 6 + cc[qq]) * 32 + q * PC;
                        float dv[PC];
#pragma unroll
                        for (int j = 0; j < PC; j += 4) *reinterpret_cast<float4*>(dv + j) = *reinterpret_cast<const float4*>(d + j);
#pragma unroll
                        for (int j = 0; j < PC; ++j) acc[j] += wy[pp] * wx[qq] * dv[j];
                    }
                rh[i] = pack16<T>(acc);
            }
        }
        if constexpr (INBN) {
            // this thread's pieces all hold the same channels (piece index = tid & 3): c0 + (tid & 3) * PIECE ..
            constexpr int PC = 16 / (int)sizeof(T);
            float sc[PC], sh[PC];
            const float* sp = bnp + c0 + (tid & 3) * PC;
#pragma unroll
            for (int j = 0; j < PC; j += 4) {
                *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + p.Cin + j);
            }
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                if (hoff[i] == ~0u) continue;            // padding of the post-activation tensor: stays zero
                float v[PC];
                unpack16(rh[i], v);
#pragma unroll
                for (int j = 0; j < PC; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                rh[i] = pack16<T>(v);
            }
        }
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int idx = tid + i * 256;
            if (idx < HH * HWD * 4) *reinterpret_cast<uint4*>(halo + HL::off(idx >> 2, idx & 3)) = rh[i];
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + i * 256;
            if (idx < TAPS * NT * 4) *reinterpret_cast<uint4*>(wts + wt_off(idx >> 2, idx & 3)) = rw[i];
        }
    };

    // PF = false (layers with at most two channel chunks): nothing to pipeline inside a workgroup, so the staging
    // registers are not kept alive across the MFMAs -- fewer VGPRs, one more resident workgroup per CU does the overlap
    if constexpr (INBN) {
        const int grp = n / (p.N / p.in_groups);
        for (int c = tid; c < p.Cin; c += 256) {
            bnp[c] = p.in_scale[grp * p.Cin + c];
            bnp[p.Cin + c] = p.in_shift[grp * p.Cin + c];
        }
        __syncthreads();
    }
    if constexpr (INUP4) {
        // |A - B| of the 4 x 6 coarse pixels this tile's 10 x 18 halo interpolates from (coarse rows oy0 / 4 - 1 .. + 2, columns
        // ox0 / 4 - 1 .. + 4, clamped to the map: a clamped entry is only ever read with weight 0), fp32, once per workgroup
        const int CH = p.H >> 2, CW = p.W >> 2;
        if (tid < 4 * 6 * 4) {
            const int fp = tid >> 2, q = tid & 3, r = fp / 6, c = fp - r * 6;
            int cy = (oy0 >> 2) - 1 + r, cx = (ox0 >> 2) - 1 + c;
            cy = cy < 0 ? 0 : (cy > CH - 1 ? CH - 1 : cy);
            cx = cx < 0 ? 0 : (cx > CW - 1 ? CW - 1 : cx);
            const size_t off = (((size_t)n * CH + cy) * CW + cx) * 32 + q * 8;
            float u[8], v[8];
            unpack16(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.up4_a) + off), u);
            unpack16(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.up4_b) + off), v);
#pragma unroll
            for (int j = 0; j < 8; ++j) bnp[fp * 32 + q * 8 + j] = fabsf(u[j] - v[j]);
        }
        __syncthreads();
    }
    if (PF) fetch(0);
    for (int c0 = 0; c0 < p.Cin; c0 += CK) {
        if (!PF) fetch(c0);
        commit(c0);
        __syncthreads();
        if (PF && c0 + CK < p.Cin) fetch(c0 + CK);
        // The taps run column-major (kw outer) so that a pixel fragment -- halo row h = r*STRIDE + kh*DIL of column kw -- is
        // read from LDS once and reused by every (r, kh) that lands on it; the fragments of step i+1 (4 weight sub-tiles
        // + the new pixel rows) are issued BEFORE the MFMAs of step i, a whole tap (4*RW MFMAs) ahead of their use.
        // (Left to the scheduler the reads sat 4 MFMAs ahead of their consumers and every group stalled on LDS latency.)
        // split-bf16 forms: one pass over the taps per PIXEL plane pb, against the weight planes 0 .. NPL - 1 - pb (the products
        // a_i * b_j with i + j < NPL; a = weights, b = pixels)
#pragma unroll
        for (int pb = 0; pb < NPL; ++pb) {
            constexpr int HR = (RW - 1) * STRIDE + (KS - 1) * DIL + 1;
            const unsigned char* hp = halo + pb * HB;
            const int na = NPL - pb;              // weight planes of this pass
            V16u B[KS][HR], A[2][NPL][NS];
            bool have[KS][HR];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int h = 0; h < HR; ++h) have[i][h] = false;
            auto issue = [&](int step) {
                const int kw = step / KS, kh = step - kw * KS, tap = kh * KS + kw;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int h = r * STRIDE + kh * DIL;
                    if (!have[kw][h]) {
                        have[kw][h] = true;
                        B[kw][h].u = *reinterpret_cast<const uint4*>(
                            hp + HL::off((RW * wv * STRIDE + h) * HWD + pl * STRIDE + kw * DIL, g));
                    }
                }
#pragma unroll
                for (int ap = 0; ap < NPL; ++ap)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        if (ap < na) A[step & 1][ap][s].u = *reinterpret_cast<const uint4*>(wts + ap * WB + wt_off(tap * NT + s * 16 + pl, g));
            };
            issue(0);
#pragma unroll
            for (int step = 0; step < TAPS; ++step) {
                const int kw = step / KS, kh = step - kw * KS;
                if (step + 1 < TAPS) issue(step + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ap = 0; ap < NPL; ++ap)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int r = 0; r < RW; ++r)
                            if (ap < na) Mma<T>::run(A[step & 1][ap][s], B[kw][r * STRIDE + kh * DIL], acc[s][r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ----
    if constexpr (std::is_same<T, f32h3>::value) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < RW; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[s][r][j] *= (1.f / F32H3_WSCALE);
    }
    if constexpr (NT == 16 && KS == 3 && STRIDE == 1) {
        if (p.y_nchw) {
            float* out = p.y_nchw + (size_t)n * p.Cout * p.OH * p.OW;
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int oy = oy0 + RW * wv + r, ox = ox0 + pl;
                if (oy >= p.OH || ox >= p.OW) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = g * 4 + j;         // NT = 16: one sub-tile, lane group g holds channels 4 g .. 4 g + 3
                    if (c < p.Cout) out[((size_t)c * p.OH + oy) * p.OW + ox] = acc[0][r][j] + (p.bias ? p.bias[c] : 0.f);
                }
            }
            return;
        }
    }
    T* yout = reinterpret_cast<T*>(p.y) + (size_t)n * p.OH * p.OW * p.Cout;
    const T* rin = p.res ? reinterpret_cast<const T*>(p.res) + (size_t)n * p.OH * p.OW * p.Cout : nullptr;
    if constexpr (KS == 2) {
        if (p.phase_mode == 1)          // coarse residual [N][OH][OW][NT], phase (0, 0) only
            rin = (p.res && cb == 0) ? reinterpret_cast<const T*>(p.res) + (size_t)n * p.OH * p.OW * NT : nullptr;
    }
    const bool vec_ok = (p.Cout & 3) == 0;
    // Output path: a lane holds 4 channels of one pixel (8 / 16 bytes), i.e. a direct store writes 32-byte runs at a
    // Cout-sized stride -- measured 14-27 us per launch on the trunk layers.  The tile is instead transposed through
    // the (now free) staging LDS and written as 16-byte pieces, NT*sizeof(T) contiguous bytes per pixel.
    constexpr int PIECE = 16 / (int)sizeof(T);               // channels per 16-byte piece
    constexpr int TPITCH = NT * (int)sizeof(T) + 16;         // LDS bytes per pixel row of the transposed tile
    constexpr int RED_BYTES = 4 * 2 * NT * 4;                // the statistics scratch sits below the tile
    const bool wide = FAST || (vec_ok && (p.Cout % PIECE) == 0);
    unsigned char* otile = smem + RED_BYTES;
    float ssum[NS][4], ssq[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[s][j] = 0.f; ssq[s][j] = 0.f; }

    // The trunk's layers (no gating, no pre-activation copy, no GELU) take a COMPACT epilogue, as a separate
    // instantiation (FAST; chosen by the host).  The fully general one below is ~20 000 instructions once unrolled over
    // the 4*RW accumulator tiles, i.e. ~160 KB of straight-line code that every workgroup had to stream through the
    // instruction cache once: measured 5.6 (RW = 2) to 18 us (RW = 4) per workgroup, more than the matrix work of the
    // 64- and 128-channel layers.  (As a run-time branch in one kernel the compiler interleaved the two paths again.)
    if constexpr (FAST) {
        // straight-line variants, ONE executed: without statistics (data-gradient launches), with statistics on a
        // tile that lies fully inside the image (no masking), the masked general case; each with / without ReLU
        float bs[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = co0 + s * 16 + g * 4 + j;
                bs[s][j] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
            }
        auto body = [&](auto with_stats, auto masked, auto relu) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const int oy = oy0 + RW * wv + r, ox = ox0 + pl;
                const bool pvalid = (oy < p.OH) && (ox < p.OW) && (oy * p.OW + ox < p.npix);
                T* trow = reinterpret_cast<T*>(otile + ((RW * wv + r) * TW + pl) * TPITCH) + g * 4;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[s][r][j] + bs[s][j];
                    if (rin) {
                        const int c = co0 + s * 16 + g * 4;
                        if (pvalid && c < p.Cout) {
                            float rr[4];
                            const size_t roff = (KS == 2 && p.phase_mode == 1) ? (size_t)(oy * p.OW + ox) * NT + (c - co0)
                                                                              : (size_t)(oy * p.OW + ox) * p.Cout + c;
                            ld4(rin + roff, rr);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += rr[j];
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (decltype(relu)::value) v[j] = fmaxf(v[j], 0.f);
                        if constexpr (decltype(with_stats)::value) {
                            // channels beyond Cout carry zero weights and zero bias: they add 0 without a mask
                            const float m = (decltype(masked)::value && !pvalid) ? 0.f : v[j];
                            ssum[s][j] += m;
                            ssq[s][j] += m * m;
                        }
                    }
                    st4(trow + s * 16, v);
                }
            }
        };
        const bool inside = oy0 + TH <= p.OH && ox0 + TW <= p.OW && (oy0 + TH - 1) * p.OW + ox0 + TW - 1 < p.npix;
        auto pick = [&](auto relu) {
            if (!p.stats) body(std::false_type{}, std::false_type{}, relu);
            else if (inside) body(std::true_type{}, std::false_type{}, relu);
            else body(std::true_type{}, std::true_type{}, relu);
        };
        if constexpr (NT == 32 && KS == 3 && STRIDE == 1 && RW == 2 && !INBN) {
            if (p.up4_partial) {
                // fp32 tile -> LDS, then the 4 x 6 x NT coarse sums of this tile by a SEPARABLE reduction (columns, then rows)
                // with the interpolation weights of the tile's 16 columns / 8 rows tabulated once (most of them are zero:
                // a coarse column collects from the 8 fine columns 4 C - 2 .. 4 C + 5)
                constexpr int FP = NT + 4;                            // floats per pixel: 16-byte aligned float4 reads
                float* ft = reinterpret_cast<float*>(smem);           // [TH * TW][FP]
                float* tmp = ft + TH * TW * FP;                       // [TH][6][NT]
                float* wxT = tmp + TH * 6 * NT;                       // [6][TW]
                float* wyT = wxT + 6 * TW;                            // [4][TH]
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        float4 v4 = make_float4(acc[s][r][0] + bs[s][0], acc[s][r][1] + bs[s][1], acc[s][r][2] + bs[s][2],
                                                acc[s][r][3] + bs[s][3]);
                        *reinterpret_cast<float4*>(ft + ((RW * wv + r) * TW + pl) * FP + s * 16 + g * 4) = v4;
                    }
                const int CH = p.OH >> 2, CW = p.OW >> 2;             // coarse grid
                auto weight = [](int d, int in, int target) {         // weight of fine index d on coarse index `target`
                    float sc = ((float)d + 0.5f) * 0.25f - 0.5f;      // (= bil_src of pointwise.hip)
                    if (sc < 0.f) sc = 0.f;
                    const int i0 = (int)sc, i1 = i0 + (i0 < in - 1 ? 1 : 0);
                    const float l = sc - (float)i0;
                    return (i0 == target ? 1.f - l : 0.f) + (i1 == target ? l : 0.f);
                };
                if (tid < 6 * TW) wxT[tid] = weight(ox0 + tid % TW, CW, 4 * tx - 1 + tid / TW);
                else if (tid < 6 * TW + 4 * TH) wyT[tid - 6 * TW] = weight(oy0 + (tid - 6 * TW) % TH, CH, 2 * ty - 1 + (tid - 6 * TW) / TH);
                __syncthreads();
                constexpr int C4 = NT / 4;
                for (int o = tid; o < TH * 6 * C4; o += 256) {        // columns: tmp[r][lc][ch4]
                    const int c4 = o % C4, lc = (o / C4) % 6, r = o / (C4 * 6);
                    const int c_lo = 4 * lc - 6 > 0 ? 4 * lc - 6 : 0, c_hi = 4 * lc + 2 < TW ? 4 * lc + 2 : TW;
                    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int c = c_lo; c < c_hi; ++c) {
                        const float wgt = wxT[lc * TW + c];
                        const float4 f = *reinterpret_cast<const float4*>(ft + (r * TW + c) * FP + c4 * 4);
                        t4.x += wgt * f.x; t4.y += wgt * f.y; t4.z += wgt * f.z; t4.w += wgt * f.w;
                    }
                    *reinterpret_cast<float4*>(tmp + (r * 6 + lc) * NT + c4 * 4) = t4;
                }
                __syncthreads();
                for (int o = tid; o < 4 * 6 * C4; o += 256) {         // rows: out[lr][lc][ch4]
                    const int c4 = o % C4, lc = (o / C4) % 6, lr = o / (C4 * 6);
                    float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int r = 0; r < TH; ++r) {
                        const float wgt = wyT[lr * TH + r];
                        const float4 f = *reinterpret_cast<const float4*>(tmp + (r * 6 + lc) * NT + c4 * 4);
                        t4.x += wgt * f.x; t4.y += wgt * f.y; t4.z += wgt * f.z; t4.w += wgt * f.w;
                    }
                    *reinterpret_cast<float4*>(p.up4_partial + (size_t)tile * (4 * 6 * NT) + (lr * 6 + lc) * NT + c4 * 4) = t4;
                }
                return;
            }
        }
        if (p.act == DH_ACT_RELU) pick(std::true_type{});
        else pick(std::false_type{});
    } else {
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int oy = oy0 + RW * wv + r, ox = ox0 + pl;
        const bool pvalid = (oy < p.OH) && (ox < p.OW) && (oy * p.OW + ox < p.npix);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int c = co0 + s * 16 + g * 4;
            if (!pvalid || c >= p.Cout) continue;
            float v[4] = {acc[s][r][0], acc[s][r][1], acc[s][r][2], acc[s][r][3]};
            const size_t off = (size_t)(oy * p.OW + ox) * p.Cout + c;
            if (vec_ok) {
                if (p.bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += p.bias[c + j];
                }
                if (rin) {
                    float rr[4];
                    ld4(rin + off, rr);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rr[j];
                }
                if (p.y2) st4(reinterpret_cast<T*>(p.y2) + (size_t)n * p.OH * p.OW * p.Cout + off, v);
                if (p.gate_y) {
                    const size_t goff = (size_t)n * p.OH * p.OW * p.Cout + off;
                    const int gi = (n / (p.N / p.gate_groups)) * p.Cout + c;
                    float xv[4];
                    ld4(reinterpret_cast<const T*>(p.gate_y) + goff, xv);
                    if (p.gate_out) {
                        float ov[4];
                        ld4(reinterpret_cast<const T*>(p.gate_out) + goff, ov);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = ov[j] > 0.f ? v[j] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xh = (xv[j] - p.gate_mean[gi + j]) * p.gate_invstd[gi + j];
                        ssum[s][j] += v[j];
                        ssq[s][j] += v[j] * xh;
                    }
                    if (wide) st4(reinterpret_cast<T*>(otile + ((RW * wv + r) * TW + pl) * TPITCH) + s * 16 + g * 4, v);
                    else st4(yout + off, v);
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (p.act == DH_ACT_RELU) v[j] = fmaxf(v[j], 0.f);
                    else if (p.act == DH_ACT_GELU) v[j] = gelu_erf(v[j]);
                    ssum[s][j] += v[j];
                    ssq[s][j] += v[j] * v[j];
                }
                if (wide) st4(reinterpret_cast<T*>(otile + ((RW * wv + r) * TW + pl) * TPITCH) + s * 16 + g * 4, v);
                else st4(yout + off, v);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (c + j >= p.Cout) continue;
                    float t = v[j];
                    if (p.bias) t += p.bias[c + j];
                    if (rin) t += ldf(rin + off + j);
                    if (p.y2) stf(reinterpret_cast<T*>(p.y2) + (size_t)n * p.OH * p.OW * p.Cout + off + j, t);
                    if (p.act == DH_ACT_RELU) t = fmaxf(t, 0.f);
                    else if (p.act == DH_ACT_GELU) t = gelu_erf(t);
                    ssum[s][j] += t;
                    ssq[s][j] += t * t;
                    stf(yout + off + j, t);
                }
            }
        }
    }

    }   // generic epilogue
    if (p.stats) {
        // reduce over the 16 pixel lanes of each lane group, then over the 4 waves through LDS
        float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][NT]; staging LDS is free now
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row16_sum(ssum[s][j]), b = row16_sum(ssq[s][j]);
                if (pl == 0) {
                    red[(wv * 2 + 0) * NT + s * 16 + g * 4 + j] = a;
                    red[(wv * 2 + 1) * NT + s * 16 + g * 4 + j] = b;
                }
            }
    }
    if (p.stats || wide) __syncthreads();
    if (wide) {
        constexpr int PPR = NT * (int)sizeof(T) / 16;         // 16-byte pieces per pixel
        for (int i = tid; i < TH * TW * PPR; i += 256) {
            const int px = i / PPR, q = i - px * PPR;
            const int oy = oy0 + px / TW, ox = ox0 + px % TW, c = co0 + q * PIECE;
            if (oy < p.OH && ox < p.OW && oy * p.OW + ox < p.npix && c < p.Cout) {
                size_t dst = (size_t)(oy * p.OW + ox) * p.Cout + c;
                if (KS == 2 && p.phase_mode == 1)        // depth-to-space: phase (a, b) = this cout block, NT physical channels
                    dst = (size_t)((2 * oy + (cb >> 1)) * (2 * p.OW) + 2 * ox + (cb & 1)) * NT + q * PIECE;
                *reinterpret_cast<uint4*>(yout + dst) = *reinterpret_cast<const uint4*>(otile + px * TPITCH + q * 16);
            }
        }
    }
    if (p.stats) {
        const float* red = reinterpret_cast<const float*>(smem);
        if (tid < 2 * NT) {
            const int which = tid / NT, c = tid - which * NT;
            const float t = red[(0 * 2 + which) * NT + c] + red[(1 * 2 + which) * NT + c] +
                            red[(2 * 2 + which) * NT + c] + red[(3 * 2 + which) * NT + c];
            if (co0 + c < p.CoutPad)
                p.stats[((size_t)which * p.CoutPad + co0 + c) * gridDim.x + tile] = t;   // [2][CoutPad][tiles]
        }
    }
}

// rows per wavefront: 16x16-pixel tiles (RW = 4) for 3x3 stride-1 layers with enough tiles to fill the
// chip -- half the weight staging per FLOP and 8 instead of 6 LDS fragment reads per 16 MFMAs
static inline int pick_rw(int N, int OH, int OW, int Cin, int ks, int stride) {
    // (16-row tiles for the 32-channel 256x256 head convolutions: measured neutral, 3.834 vs 3.840 ms per step)
    // (layers with 1-2 channel chunks have nothing to pipeline and prefer more, smaller workgroups)
    if ((ks == 3 || ks == 2) && stride == 1 && Cin >= 128 && OH >= 16 && (long)N * dh_cdiv(OH, 16) * dh_cdiv(OW, TW) >= 256) return 4;
    return 2;
}

template <typename T, int KS, int STRIDE, int NT, int RW, int DIL, bool PF, bool FAST, bool INBN = false, bool INUP4 = false>
int launch_fast(const ConvArgs& a, hipStream_t st) {
    constexpr int TH = 4 * RW;
    constexpr int HH = (TH - 1) * STRIDE + (KS - 1) * DIL + 1, HWD = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    const size_t staging = Prec<T>::NPL * ((size_t)HH * HWD * HaloLayout<STRIDE>::PITCH + (size_t)KS * KS * NT * WPITCH) +
                           (INBN ? (size_t)2 * a.Cin * sizeof(float) : 0) + (INUP4 ? (size_t)4 * 6 * 32 * sizeof(float) : 0);
    const size_t otile = (size_t)4 * 2 * NT * 4 + (size_t)TH * TW * (NT * sizeof(T) + 16);     // epilogue: stats scratch + transposed tile
    const size_t lds = staging > otile ? staging : otile;
    auto kern = conv_mfma_kernel<T, KS, STRIDE, NT, RW, DIL, PF, FAST, INBN, INUP4>;
    static bool attr_done = false;      // once per instantiation (and never inside a graph capture)
    if (lds > 64 * 1024 && !attr_done) {
        attr_done = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_mfma: cannot raise dynamic LDS to %zu", lds);
        }
    }
    dim3 grid(a.N * a.tilesX * a.tilesY, a.CoutPad / NT);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
    DH_CHECK_LAUNCH("conv_mfma");
    return 0;
}

template <typename T, int KS, int STRIDE, int NT, int RW, int DIL, bool PF>
int launch_pf(const ConvArgs& a, hipStream_t st) {
    // compact-epilogue instantiation: 16-byte output pieces, no gating / pre-activation copy / GELU
    const bool fast = ((a.Cout % (16 / (int)sizeof(T))) == 0 || a.y_nchw) && !a.gate_y && !a.y2 && a.act != DH_ACT_GELU;
    if constexpr (KS == 3 && STRIDE == 1 && DIL == 1) {        // BN-apply + ReLU on load: the 3x3 consumers of a BN layer
        if (a.in_scale) {
            if (fast) return launch_fast<T, KS, STRIDE, NT, RW, DIL, PF, true, true>(a, st);
            return launch_fast<T, KS, STRIDE, NT, RW, DIL, PF, false, true>(a, st);
        }
    } else if (a.in_scale) {
        DH_FAIL("conv_mfma: BatchNorm-on-load is built for 3x3 stride-1 dilation-1 convolutions (got %dx%d s%d d%d)", KS, KS, STRIDE, DIL);
    }
    if constexpr (KS == 3 && STRIDE == 1 && DIL == 1 && NT == 32 && RW == 2 && !PF && sizeof(T) == 2) {
        if (a.up4_a) {
            if (!fast || a.in_scale) DH_FAIL("conv_mfma: the bilinear-x4-on-load form has the compact epilogue and no BatchNorm on load");
            return launch_fast<T, KS, STRIDE, NT, RW, DIL, PF, true, false, true>(a, st);
        }
    }
    if (a.up4_a) DH_FAIL("conv_mfma: bilinear x4 on load is built for the bf16 3x3 / stride 1, 32 -> 32 channel convolution on 8-row tiles");
    if (fast) return launch_fast<T, KS, STRIDE, NT, RW, DIL, PF, true>(a, st);
    return launch_fast<T, KS, STRIDE, NT, RW, DIL, PF, false>(a, st);
}

template <typename T, int KS, int STRIDE, int NT, int RW, int DIL>
int launch_rw(const ConvArgs& a, hipStream_t st) {
    constexpr int CK = Prec<T>::CK;
    if constexpr (RW == 2 && KS == 3 && DIL == 1) {
        // (split-bf16 forms run one workgroup per CU -- no second workgroup to cover a chunk's loads -- so a two-chunk layer
        // prefetches its second chunk under the first one's MFMAs: +2.9 % on the bf16x3 step, same-box)
        if (a.Cin <= 2 * CK && !(Prec<T>::NPL > 1 && a.Cin == 2 * CK)) return launch_pf<T, KS, STRIDE, NT, RW, DIL, false>(a, st);
    }
    return launch_pf<T, KS, STRIDE, NT, RW, DIL, true>(a, st);
}

template <typename T, int KS, int STRIDE, int NT>
int launch(const ConvArgs& a, hipStream_t st) {
    if constexpr (KS == 3 && STRIDE == 1) {
        if (a.dil == 2) return a.rw == 4 ? launch_rw<T, KS, STRIDE, NT, 4, 2>(a, st) : launch_rw<T, KS, STRIDE, NT, 2, 2>(a, st);
        if (a.rw == 4) return launch_rw<T, KS, STRIDE, NT, 4, 1>(a, st);
    }
    if constexpr (KS == 2) {
        if (a.rw == 4) return launch_rw<T, KS, STRIDE, NT, 4, 1>(a, st);
    }
    return launch_rw<T, KS, STRIDE, NT, 2, 1>(a, st);
}

template <typename T, int KS, int STRIDE>
int launch_nt(const ConvArgs& a, hipStream_t st) {
    if constexpr (KS == 2) {      // the phase convolutions: forward = one cout block (32 or 64 channels) per phase
        if (a.phase_mode == 1) return a.Cout == 128 ? launch<T, KS, STRIDE, 32>(a, st) : launch<T, KS, STRIDE, 64>(a, st);
        return a.Cout % 64 ? launch<T, KS, STRIDE, 32>(a, st) : launch<T, KS, STRIDE, 64>(a, st);     // data gradient: Cout = the 3x3's Cin
    }
    if constexpr (Prec<T>::NPL > 1) {
        // split-bf16 forms: NPL planes of halo + weights must fit the CU's 160 KB of LDS -- the widest output tile that does
        // (e.g. three planes of a 16-row 3x3 tile: 32 channels; stride 2 with three planes: none -- DH_CONV_NO_FIT tells the
        // caller to take the exact fp32 kernel for that launch)
        const int rw = (KS == 3 && STRIDE == 1) || KS == 2 ? a.rw : 2, dil = (KS == 3 && STRIDE == 1) ? a.dil : 1;
        const int hh = (4 * rw - 1) * STRIDE + (KS - 1) * dil + 1, hwd = (TW - 1) * STRIDE + (KS - 1) * dil + 1;
        const size_t budget = 160 * 1024 - (a.in_scale ? (size_t)2 * a.Cin * sizeof(float) : 0);
        auto fits = [&](int nt) {
            return Prec<T>::NPL * ((size_t)hh * hwd * HaloLayout<STRIDE>::PITCH + (size_t)KS * KS * nt * WPITCH) <= budget;
        };
        if (a.CoutPad % 64 == 0 && fits(64)) return launch<T, KS, STRIDE, 64>(a, st);
        if (a.CoutPad % 32 == 0 && fits(32)) return launch<T, KS, STRIDE, 32>(a, st);
        // (16-wide tiles only for layers that ARE that narrow: the one 3x3 stride-2 layer, 64 -> 128 channels, fits three planes
        // at 16 channels per workgroup and then takes 165 us where the exact fp32 kernel takes 97)
        if (a.CoutPad % 32 != 0 && fits(16)) return launch<T, KS, STRIDE, 16>(a, st);
        return DH_CONV_NO_FIT;
    }
    if (a.CoutPad % 64 == 0) return launch<T, KS, STRIDE, 64>(a, st);
    if (a.CoutPad % 32 == 0) return launch<T, KS, STRIDE, 32>(a, st);
    return launch<T, KS, STRIDE, 16>(a, st);
}

template <typename T>
int launch_ks(const ConvArgs& a, int ks, int stride, hipStream_t st) {
    if (ks == 3 && stride == 1) return launch_nt<T, 3, 1>(a, st);
    if (ks == 3 && stride == 2) return launch_nt<T, 3, 2>(a, st);
    if (ks == 1 && stride == 1) return launch_nt<T, 1, 1>(a, st);
    if (ks == 1 && stride == 2) return launch_nt<T, 1, 2>(a, st);
    if (ks == 4 && stride == 1) return launch_nt<T, 4, 1>(a, st);     // space-to-depth stem
    if (ks == 2 && stride == 1 && a.phase_mode) return launch_nt<T, 2, 1>(a, st);      // phase convs of upsample-x2 + 3x3
    DH_FAIL("conv_mfma: unsupported kernel %dx%d stride %d", ks, ks, stride);
}

}  // namespace
