// 3x3 stride-1 MFMA convolution of the >= 128-channel layers on 16x32-pixel tiles with 64-channel stages (bf16, 512 threads).
//
// conv_mfma_kernel<bf16, 3, 1, 64, 4, ...> stages, per 32-channel chunk and per 16x16-pixel tile, 20.7 KB of halo + 36.9 KB of
// weights = 100 bytes per MFMA, as 64-byte pieces (half a cache line per pixel / weight row), two independent workgroups per
// CU.  The per-role clocks of the wave-specialised experiments (conv_ws.hip) say that this L2 -> LDS fill path -- ~2.7 us from
// load issue to data, ~40 GB/s per CU -- and not the barrier structure bounds the kernel.  Here ONE 512-thread workgroup
// computes two horizontally adjacent 16x16 tiles for the same 64 output channels and stages 64 channels at a time:
//   * the weights of a stage (73.7 KB) are staged once for both tiles and the two tiles share their inner halo columns
//     (18 x 34 instead of 2 x 18 x 18 pixels): 68 bytes per MFMA;
//   * every load is a full 128-byte line (8 lanes per pixel / weight row);
//   * two barriers per 64 channels instead of four.
// Wave w computes rows 4 (w & 3) .. + 3 of tile half w >> 2.  Fragment layouts, tap order and epilogue arithmetic are those of
// the generic kernel (FAST epilogue: + bias, + residual, ReLU, BatchNorm partials per 16x16 tile, 16-byte stores).
#include "conv_mfma_impl.h"

namespace {

constexpr int CP_TH = 16, CP_TW2 = 32, CP_NT = 64;

template <bool INBN>
__global__ __launch_bounds__(512) void conv3x3_pair_kernel(ConvArgs p) {
    constexpr int KS = 3, TAPS = 9, NT = CP_NT, RW = 4, NS = 4;
    constexpr int HH = CP_TH + 2, HWD = CP_TW2 + 2, NPX = HH * HWD;         // 18 x 34 halo pixels
    using HL = HaloLayout<1>;
    constexpr int HBYTES = 2 * NPX * HL::PITCH;                             // two 32-channel chunk images: 78 336
    constexpr int WBYTES = 2 * TAPS * NT * WPITCH;                          // 73 728
    constexpr int TPITCH = NT * 2 + 16;
    constexpr int OBYTES = CP_TH * CP_TW2 * TPITCH;                         // transposed output tile (aliases the staging area)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* halo = smem;
    unsigned char* wts = smem + HBYTES;
    float* red = reinterpret_cast<float*>(smem + OBYTES);                   // [8 waves][2][NT], behind the output tile
    float* bnp = reinterpret_cast<float*>(smem + HBYTES + WBYTES);          // INBN: [in_groups][2][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, half = wv >> 2, wr = wv & 3;
    const int pl = lane & 15, g = lane >> 4;
    // XCD-aware (pixel tile, channel block) order, as conv_mfma_kernel
    int tile = blockIdx.x, cb = blockIdx.y;
    if ((gridDim.x & 7) == 0 && !p.no_xcd_remap) {
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7, s = lin >> 3;
        tile = (int)(xcd * (gridDim.x >> 3) + s / gridDim.y);
        cb = (int)(s % gridDim.y);
    }
    const int ptx = p.tilesX >> 1;                        // pairs per tile row
    int bt = tile;
    const int tx2 = bt % ptx; bt /= ptx;
    const int ty = bt % p.tilesY;
    const int n = bt / p.tilesY;
    const int oy0 = ty * CP_TH, ox0 = tx2 * CP_TW2, co0 = cb * NT;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;

    if constexpr (INBN) {
        const int grp = n / (p.N / p.in_groups);
        for (int c = tid; c < p.Cin; c += 512) {
            bnp[c] = p.in_scale[grp * p.Cin + c];
            bnp[p.Cin + c] = p.in_shift[grp * p.Cin + c];
        }
    }

    // ---- staging: this thread's 16-byte pieces (piece p8 of 8 = chunk p8 >> 2, slot p8 & 3) ----
    constexpr int NHV = (NPX * 8 + 511) / 512;            // 10
    constexpr int NWV = TAPS * NT * 8 / 512;              // 9
    const int p8 = tid & 7, prow = tid >> 3;              // halo pixel / weight row prow + 64 i
    unsigned hoff[NHV];                                   // byte offset of the pixel from the image base, ~0u outside
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
        const int px = prow + 64 * i, hy = px / HWD, hx = px - hy * HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = px < NPX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        hoff[i] = ok ? (unsigned)(iy * p.W + ix) * (unsigned)p.Cin * 2u + p8 * 16 : ~0u;
    }
    const unsigned char* xin = reinterpret_cast<const unsigned char*>(p.x) + (size_t)n * p.H * p.W * p.Cin * 2;
    const unsigned char* wgt = reinterpret_cast<const unsigned char*>(p.w) + ((size_t)(co0 + prow) * p.Cin) * 2 + p8 * 16;
    const size_t wstep = (size_t)p.CoutPad * p.Cin * 2;   // tap stride (weight row prow + 64 i = tap i, channel co0 + prow)
    uint4 rh[NHV], rw[NWV];
    auto fetch = [&](int c0) {
        const unsigned char* xb = xin + c0 * 2;
        const unsigned char* wb = wgt + c0 * 2;
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const bool ok = hoff[i] != ~0u;
            const uint4 v = *reinterpret_cast<const uint4*>(xb + (ok ? hoff[i] : 0u));
            rh[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) rw[i] = *reinterpret_cast<const uint4*>(wb + (size_t)i * wstep);
    };
    auto commit = [&](int c0) {
        if constexpr (INBN) {
            float sc[8], sh[8];
            const float* sp = bnp + c0 + p8 * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + p.Cin + j);
            }
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                if (hoff[i] == ~0u) continue;                 // padding of the post-activation tensor stays zero
                float v[8];
                unpack16(rh[i], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                rh[i] = pack16<bf16>(v);
            }
        }
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int px = prow + 64 * i;
            if (px < NPX) *reinterpret_cast<uint4*>(halo + (p8 >> 2) * NPX * HL::PITCH + HL::off(px, p8 & 3)) = rh[i];
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i)
            *reinterpret_cast<uint4*>(wts + (p8 >> 2) * TAPS * NT * WPITCH + wt_off(i * NT + prow, p8 & 3)) = rw[i];
    };

    f32x4 acc[NS][RW];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < RW; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    if constexpr (INBN) __syncthreads();
    fetch(0);
    for (int c0 = 0; c0 < p.Cin; c0 += 64) {
        commit(c0);
        __syncthreads();
        if (c0 + 64 < p.Cin) fetch(c0 + 64);
#pragma unroll
        for (int ck = 0; ck < 2; ++ck) {
            const unsigned char* hc = halo + ck * NPX * HL::PITCH;
            const unsigned char* wc = wts + ck * TAPS * NT * WPITCH;
            constexpr int HR = (RW - 1) + (KS - 1) + 1;
            V16u B[KS][HR], A[2][NS];
            bool have[KS][HR];
#pragma unroll
            for (int i = 0; i < KS; ++i)
#pragma unroll
                for (int h = 0; h < HR; ++h) have[i][h] = false;
            auto issue = [&](int step) {
                const int kw = step / KS, kh = step - kw * KS, tap = kh * KS + kw;
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const int h = r + kh;
                    if (!have[kw][h]) {
                        have[kw][h] = true;
                        B[kw][h].u = *reinterpret_cast<const uint4*>(hc + HL::off((RW * wr + h) * HWD + half * 16 + pl + kw, g));
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) A[step & 1][s].u = *reinterpret_cast<const uint4*>(wc + wt_off(tap * NT + s * 16 + pl, g));
            };
            issue(0);
#pragma unroll
            for (int step = 0; step < TAPS; ++step) {
                const int kw = step / KS, kh = step - kw * KS;
                if (step + 1 < TAPS) issue(step + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < RW; ++r)
                        acc[s][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[step & 1][s].h, B[kw][r + kh].h, acc[s][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (the staging LDS is free): + bias, + residual, ReLU, statistics, transpose, 16-byte stores ----
    bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)n * p.OH * p.OW * p.Cout;
    const bf16* rin = p.res ? reinterpret_cast<const bf16*>(p.res) + (size_t)n * p.OH * p.OW * p.Cout : nullptr;
    unsigned char* otile = smem;
    float ssum[NS][4], ssq[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) ssum[s][j] = ssq[s][j] = 0.f;
    const bool relu = p.act == DH_ACT_RELU;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int ly = RW * wr + r, lx = half * 16 + pl, oy = oy0 + ly, ox = ox0 + lx;
        const bool pvalid = oy < p.OH && ox < p.OW;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ch = co0 + s * 16 + g * 4;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[s][r][j] + ((p.bias && ch + j < p.Cout) ? p.bias[ch + j] : 0.f);
            if (rin && pvalid && ch < p.Cout) {
                float rr[4];
                ld4(rin + (size_t)(oy * p.OW + ox) * p.Cout + ch, rr);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rr[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (relu) v[j] = fmaxf(v[j], 0.f);
                const float m = pvalid ? v[j] : 0.f;
                ssum[s][j] += m;
                ssq[s][j] += m * m;
            }
            st4(reinterpret_cast<bf16*>(otile + (ly * CP_TW2 + lx) * TPITCH) + s * 16 + g * 4, v);
        }
    }
    if (p.stats) {
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
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CP_TH * CP_TW2 * 8 / 512; ++k) {
        const int i = tid + 512 * k, px = i >> 3, piece = i & 7;
        const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31), ch = co0 + piece * 8;
        if (oy < p.OH && ox < p.OW && ch < p.Cout)
            *reinterpret_cast<uint4*>(yout + (size_t)(oy * p.OW + ox) * p.Cout + ch) =
                *reinterpret_cast<const uint4*>(otile + px * TPITCH + piece * 16);
    }
    if (p.stats && tid < 4 * NT) {                        // (tile half, which, channel): the four waves of that half
        const int hf = tid / (2 * NT), which = (tid / NT) & 1, ch = tid % NT;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += red[((hf * 4 + w) * 2 + which) * NT + ch];
        const int ntile = p.N * p.tilesX * p.tilesY, t16 = (n * p.tilesY + ty) * p.tilesX + 2 * tx2 + hf;
        if (co0 + ch < p.CoutPad) p.stats[((size_t)which * p.CoutPad + co0 + ch) * ntile + t16] = t;       // [2][CoutPad][16x16 tiles]
    }
}

template <bool INBN>
int launch_pair(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * (CP_TH + 2) * (CP_TW2 + 2) * HaloLayout<1>::PITCH + (size_t)2 * 9 * CP_NT * WPITCH +
                       (INBN ? (size_t)2 * a.Cin * 4 : 0);
    static bool attr_done = false;
    if (!attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_pair_kernel<INBN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv3x3_pair: cannot raise dynamic LDS to 160 KB");
        }
    }
    dim3 grid(a.N * a.tilesY * (a.tilesX / 2), a.CoutPad / CP_NT);
    hipLaunchKernelGGL(conv3x3_pair_kernel<INBN>, grid, dim3(512), lds, st, a);
    DH_CHECK_LAUNCH("conv3x3_pair");
    return 0;
}

}  // namespace

// eligibility: launches of conv_mfma_kernel<bf16, 3, 1, 64, 4, 1, true, true, *> whose tile rows pair up
bool dh_conv_pair_eligible(const ConvArgs& a, int ks, int stride, int dtype) {
    // OFF by default (DAHITRA_CONV_PAIR=1 turns it on): measured SLOWER -- layer3 108 vs 75 us, layer2 34 vs 26 us -- although it
    // moves 32 % fewer bytes through the fill path in full cache lines.  Third data point (after conv_ws.hip and conv64.hip) for
    // the same finding: ONE workgroup per CU whose waves share every barrier exposes the ~2.7 us load round trip and the commit
    // phase in full, whereas two INDEPENDENT 256-thread workgroups drift apart and cover each other's staging with MFMAs; the
    // generic kernel's small independent workgroups, not a bigger tile, are what this chip rewards at these layer sizes.
    static const bool off = getenv("DAHITRA_CONV_PAIR") == nullptr;
    if (off || dtype != DH_DTYPE_BF16 || ks != 3 || stride != 1 || a.rw != 4 || a.dil != 1 || a.pad != 1) return false;
    if (a.Cin % 64 || a.Cin < 128 || a.CoutPad % CP_NT || a.Cout % 8 || a.phase_mode || a.gate_y || a.y2 || a.y_nchw || a.w_nstride || a.w_cm) return false;
    if (a.act == DH_ACT_GELU || a.npix != a.OH * a.OW || a.in_npix != a.H * a.W || a.OH != a.H || a.OW != a.W) return false;
    if ((a.tilesX & 1) || (a.in_scale && (size_t)2 * a.Cin * 4 > 8192)) return false;
    return true;
}
int dh_conv_pair_launch(const ConvArgs& a, hipStream_t st) { return a.in_scale ? launch_pair<true>(a, st) : launch_pair<false>(a, st); }
