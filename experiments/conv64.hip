// The 64 -> 64 channel 3x3 stride-1 convolutions (ResNet layer1: 4 forward + 4 data-gradient launches of the bench step, bf16)
// with the WEIGHTS RESIDENT IN LDS.  The generic kernel (conv_mfma_impl.h) stages the 36.9 KB of weights of a 32-channel chunk
// for every 8x16-pixel tile: 2048 workgroups x 73.7 KB = 151 MB through the L2 -> LDS path per launch against 47 MB of halo and
// 33.5 MB of output -- and that fill path (~40 GB/s per CU, see conv_ws.hip) is what bounds the kernel: 31 - 35 us for a layer
// whose HBM time is 13 us.  Here a persistent 512-thread workgroup per CU stages all 73.7 KB once and then walks through its
// tiles; the two 4-wave groups of the workgroup each take one tile per iteration (own halo and output buffers), the halo of the
// next tile is prefetched into registers during the MFMAs.  Same fragment layouts, tap order and epilogue arithmetic as
// conv_mfma_kernel<bf16, 3, 1, 64, 2, 1, *, true, INBN>.
#include "conv_mfma_impl.h"

namespace {

constexpr int C64_TH = 8;

template <bool INBN>
__global__ __launch_bounds__(512) void conv64_kernel(ConvArgs p, int ntile) {
    constexpr int KS = 3, TAPS = 9, NT = 64, RW = 2, NS = 4, DIL = 1;
    constexpr int HH = C64_TH + 2, HWD = TW + 2, NPX = HH * HWD;             // 10 x 18 halo pixels
    using HL = HaloLayout<1>;
    constexpr int WBYTES = 2 * TAPS * NT * WPITCH;                           // both 32-channel chunks: 73 728
    constexpr int HBYTES = 2 * NPX * HL::PITCH;                              // a group's halo, two chunk images: 23 040
    constexpr int TPITCH = NT * 2 + 16;                                      // transposed output tile: bytes per pixel
    constexpr int OBYTES = C64_TH * TW * TPITCH;                             // 18 432
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wts = smem;
    const int tid = threadIdx.x, grp = tid >> 8, gt = tid & 255, lane = tid & 63, wl = (tid >> 6) & 3;
    const int pl = lane & 15, g = lane >> 4;
    unsigned char* halo = smem + WBYTES + grp * HBYTES;
    unsigned char* otile = smem + WBYTES + 2 * HBYTES + grp * OBYTES;
    float* bnp = reinterpret_cast<float*>(smem + WBYTES + 2 * HBYTES + 2 * OBYTES);      // INBN: [in_groups][2][64]
    // a wave's statistics partials [2][64] live in the 16 pad bytes of its own 32 pixels of the output buffer
    auto red = [&](int wave, int k) { return reinterpret_cast<float*>(otile + (wave * 32 + (k >> 2)) * TPITCH + NT * 2) + (k & 3); };

    // ---- weights of both chunks, once per workgroup ----
    {
        const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w);
        for (int i = tid; i < 2 * TAPS * NT * 4; i += 512) {          // 16-byte pieces: (chunk, tap, cout, q)
            const int q = i & 3, row = (i >> 2) % (TAPS * NT), ck = i / (4 * TAPS * NT);
            const int tap = row / NT, co = row - tap * NT;
            const uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)(tap * p.CoutPad + co) * 64 + ck * 32) * 2 + q * 16);
            *reinterpret_cast<uint4*>(wts + ck * TAPS * NT * WPITCH + wt_off(row, q)) = v;
        }
        if constexpr (INBN) {
            for (int i = tid; i < p.in_groups * 128; i += 512) {
                const int gi = i >> 7, k = (i >> 6) & 1, c = i & 63;
                bnp[i] = (k ? p.in_shift : p.in_scale)[gi * 64 + c];
            }
        }
    }

    // ---- this thread's halo pieces: pixel (gt >> 3) + 32 i, 16-byte piece gt & 7 (chunk = piece >> 2) ----
    constexpr int NHV = (NPX * 8 + 255) / 256;                               // 6
    const int p8 = gt & 7, prow = gt >> 3;
    int h_c[NHV];
#pragma unroll
    for (int i = 0; i < NHV; ++i) {
        const int px = prow + 32 * i;
        h_c[i] = px < NPX ? ((px / HWD) << 8) | (px % HWD) : -1;
    }
    uint4 rh[NHV];
    unsigned okmask = 0;
    int bng = 0;
    auto tile_of = [&](int pair) { return 2 * pair + grp; };
    auto fetch = [&](int tile) {
        okmask = 0;
        if (tile >= ntile) return;
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
        bng = INBN ? n / (p.N / p.in_groups) : 0;
        const int iy0 = ty * C64_TH - 1, ix0 = tx * TW - 1;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + (size_t)n * p.H * p.W * 128 + p8 * 16;
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int iy = iy0 + (h_c[i] >> 8), ix = ix0 + (h_c[i] & 0xff);
            const bool ok = h_c[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint4 v = *reinterpret_cast<const uint4*>(xb + (ok ? (unsigned)(iy * p.W + ix) * 128u : 0u));
            rh[i] = ok ? v : make_uint4(0, 0, 0, 0);
            okmask |= ok ? (1u << i) : 0u;
        }
    };
    auto commit = [&]() {
        if constexpr (INBN) {
            float sc[8], sh[8];
            const float* sp = bnp + bng * 128 + p8 * 8;
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + 64 + j);
            }
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                if (!((okmask >> i) & 1u)) continue;              // padding of the post-activation tensor stays zero
                float v[8];
                unpack16(rh[i], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                rh[i] = pack16<bf16>(v);
            }
        }
#pragma unroll
        for (int i = 0; i < NHV; ++i)
            if (h_c[i] >= 0) *reinterpret_cast<uint4*>(halo + (p8 >> 2) * NPX * HL::PITCH + HL::off(prow + 32 * i, p8 & 3)) = rh[i];
    };

    const int npair = (ntile + 1) / 2;
    int pair = blockIdx.x;
    if (pair < npair) fetch(tile_of(pair));
    for (; pair < npair; pair += gridDim.x) {
        const int tile = tile_of(pair);
        const bool active = tile < ntile;
        commit();                                        // (an idle group commits zeros: harmless)
        __syncthreads();                                 // halo staged (first iteration: the weights too)
        if (pair + (int)gridDim.x < npair) fetch(tile_of(pair + gridDim.x));

        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
        const int oy0 = ty * C64_TH, ox0 = tx * TW;

        f32x4 acc[NS][RW];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ck = 0; ck < 2; ++ck) {
            const unsigned char* hc = halo + ck * NPX * HL::PITCH;
            const unsigned char* wc = wts + ck * TAPS * NT * WPITCH;
            constexpr int HR = (RW - 1) + (KS - 1) * DIL + 1;
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
                    const int h = r + kh * DIL;
                    if (!have[kw][h]) {
                        have[kw][h] = true;
                        B[kw][h].u = *reinterpret_cast<const uint4*>(hc + HL::off((RW * wl + h) * HWD + pl + kw * DIL, g));
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
                        acc[s][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[step & 1][s].h, B[kw][r + kh * DIL].h, acc[s][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: + bias, + residual, ReLU, statistics; the wave's 2 rows through its part of the output buffer ----
        bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)n * p.OH * p.OW * p.Cout;
        const bf16* rin = p.res ? reinterpret_cast<const bf16*>(p.res) + (size_t)n * p.OH * p.OW * p.Cout : nullptr;
        float ssum[NS][4], ssq[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) ssum[s][j] = ssq[s][j] = 0.f;
        const bool relu = p.act == DH_ACT_RELU;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = oy0 + RW * wl + r, ox = ox0 + pl;
            const bool pvalid = active && oy < p.OH && ox < p.OW;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ch = s * 16 + g * 4;
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
                st4(reinterpret_cast<bf16*>(otile + ((RW * wl + r) * TW + pl) * TPITCH) + ch, v);
            }
        }
        if (p.stats) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = row16_sum(ssum[s][j]), b = row16_sum(ssq[s][j]);
                    if (pl == 0) {
                        *red(wl, s * 16 + g * 4 + j) = a;
                        *red(wl, 64 + s * 16 + g * 4 + j) = b;
                    }
                }
        }
        __syncthreads();                                 // output rows + partials staged; every wave is past its halo reads
        if (active) {
#pragma unroll
            for (int k = 0; k < C64_TH * TW * 8 / 256; ++k) {
                const int i = gt + 256 * k, px = i >> 3, piece = i & 7;
                const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15), ch = piece * 8;
                if (oy < p.OH && ox < p.OW && ch < p.Cout)
                    *reinterpret_cast<uint4*>(yout + (size_t)(oy * p.OW + ox) * p.Cout + ch) =
                        *reinterpret_cast<const uint4*>(otile + px * TPITCH + piece * 16);
            }
            if (p.stats && gt < 128) {
                const float tsum = *red(0, gt) + *red(1, gt) + *red(2, gt) + *red(3, gt);
                const int which = gt >> 6, ch = gt & 63;
                p.stats[((size_t)which * p.CoutPad + ch) * ntile + tile] = tsum;          // [2][CoutPad][tiles]
            }
        }
        // (the next iteration's barrier -- after its commit -- orders these reads before the next epilogue's writes)
    }
}

template <bool INBN>
int launch64(const ConvArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * 9 * 64 * WPITCH + 2 * (size_t)2 * (C64_TH + 2) * (TW + 2) * HaloLayout<1>::PITCH +
                       2 * (size_t)C64_TH * TW * (64 * 2 + 16) + (INBN ? (size_t)a.in_groups * 128 * 4 : 0);
    static bool attr_done = false;
    static int cus = 0;
    if (!attr_done) {
        attr_done = true;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv64_kernel<INBN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv64: cannot raise dynamic LDS to 160 KB");
        }
    }
    const int ntile = a.N * a.tilesX * a.tilesY, npair = (ntile + 1) / 2;
    const int grid = npair < cus ? npair : cus;
    hipLaunchKernelGGL(conv64_kernel<INBN>, dim3(grid), dim3(512), lds, st, a, ntile);
    DH_CHECK_LAUNCH("conv64");
    return 0;
}

}  // namespace

// eligibility: the launches conv_mfma_kernel<bf16, 3, 1, 64, 2, 1, false, true, *> (8-row tiles, compact epilogue) serves at 64 -> 64
bool dh_conv64_eligible(const ConvArgs& a, int ks, int stride, int dtype) {
    // OFF by default (DAHITRA_CONV64=1 turns it on): measured 33.0 us against 31.2 us for the generic kernel on the layer1 shape.
    // Removing the 151 MB of per-tile weight staging does not pay for what the persistent form loses: one 8-wave workgroup per CU
    // runs commit / MFMA / epilogue / store in lock step (~8 us per iteration for 1.9 us of MFMA), where the generic kernel keeps
    // three independent workgroups per CU in different phases.  A deeper pipeline (halo two tiles ahead, stores off the critical
    // path) would be needed to bring this layer to its 13 us of HBM time.
    static const bool off = getenv("DAHITRA_CONV64") == nullptr;
    if (off || dtype != DH_DTYPE_BF16 || ks != 3 || stride != 1 || a.rw != 2 || a.dil != 1 || a.pad != 1) return false;
    if (a.Cin != 64 || a.CoutPad != 64 || a.Cout % 8 || a.phase_mode || a.gate_y || a.y2 || a.y_nchw || a.w_nstride || a.w_cm) return false;
    if (a.act == DH_ACT_GELU || a.npix != a.OH * a.OW || a.in_npix != a.H * a.W || a.OH != a.H || a.OW != a.W) return false;
    if (a.in_scale && a.in_groups > 2) return false;
    return (long)a.N * a.tilesX * a.tilesY >= 1024;          // enough tiles per persistent workgroup to amortise the weight staging
}
int dh_conv64_launch(const ConvArgs& a, hipStream_t st) { return a.in_scale ? launch64<true>(a, st) : launch64<false>(a, st); }
