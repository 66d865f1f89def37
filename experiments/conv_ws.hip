// Wave-specialised, persistent form of the 3x3 stride-1 MFMA convolution for the layers with >= 128 input channels (bf16,
// 16x16-pixel tiles x 64 output channels; the layer2 / layer3 convolutions and their data gradients, 14 launches of the bench
// step).  Same tile, LDS layouts, fragment order and epilogue arithmetic as conv_mfma_kernel<bf16, 3, 1, 64, 4, DIL, true, true>
// (conv_mfma_impl.h) -- only WHO does what changes:
//   waves 4-7  PRODUCERS: every global load of the halo and the weights of a 32-channel chunk (registers), BatchNorm-apply +
//              ReLU on load, the LDS commit.  They run one chunk ahead through TWO LDS stages and straight into the next tile
//              of the workgroup's list, so a tile's prologue hides behind the previous tile's epilogue.
//   waves 0-3  CONSUMERS, one per SIMD: 144 MFMAs per chunk back to back (4 rows x 64 output channels each), then the epilogue
//              of their own 4 rows through a wave-private transposition buffer.
// One raw s_barrier per chunk (no vmcnt drain) + one per tile (BatchNorm statistics of the four consumer waves).
// The two-workgroups-per-CU form spent ~3 us of prologue + 2.75 us per chunk (1.9 us of MFMA issue, the co-resident workgroups in
// lock step) + 5 us of epilogue per 16-row tile (tools/conv_timeline.py).
#include "conv_mfma_impl.h"

namespace {

constexpr int WS_TH = 16, WS_NT = 64;

#ifdef DH_WS_TIMING
__device__ long long g_cws[1024 * 16];
#define WS_NOW() ((long long)wall_clock64())
#define WS_T(...) __VA_ARGS__
#else
#define WS_T(...)
#endif
template <int DIL, bool INBN>
__global__ __launch_bounds__(512) void conv3x3_ws_kernel(ConvArgs p, int ntile, int ncb) {
    constexpr int KS = 3, TAPS = 9, NT = WS_NT, RW = 4, NS = 4;
    constexpr int HH = (WS_TH - 1) + (KS - 1) * DIL + 1, HWD = (TW - 1) + (KS - 1) * DIL + 1;
    using HL = HaloLayout<1>;
    constexpr int STAGE = HH * HWD * HL::PITCH + TAPS * NT * WPITCH;
    constexpr int TPITCH = NT * 2 + 16;                      // transposed output tile: bytes per pixel
    constexpr int OT_WAVE = RW * TW * TPITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* otile = smem + 2 * STAGE;                 // [4 consumer waves][64 pixels][TPITCH]
    float* red = reinterpret_cast<float*>(otile + 4 * OT_WAVE);      // [4][2][NT]
    float* bnp = red + 4 * 2 * NT;                           // INBN: [in_groups][2][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    const int nchunks = p.Cin / 32, nitems = ntile * ncb;

    // item -> (pixel tile, output-channel block): an XCD (workgroup index mod 8) walks a contiguous range of pixel tiles and
    // runs the channel blocks of one tile back to back (see conv_mfma_kernel)
    auto decode = [&](int item, int& tile, int& cb) {
        if ((ntile & 7) == 0 && !p.no_xcd_remap) {
            const int xcd = item & 7, s = item >> 3;
            tile = xcd * (ntile >> 3) + s / ncb;
            cb = s % ncb;
        } else {
            tile = item / ncb;
            cb = item % ncb;
        }
    };

    if constexpr (INBN) {
        for (int i = tid; i < p.in_groups * 2 * p.Cin; i += 512) {
            const int gi = i / (2 * p.Cin), r = i - gi * 2 * p.Cin;
            bnp[i] = r < p.Cin ? p.in_scale[gi * p.Cin + r] : p.in_shift[gi * p.Cin + r - p.Cin];
        }
        __syncthreads();
    }

    if (wv >= 4) {
        // ================= producers =================
        __builtin_amdgcn_s_setprio(3);        // few instructions, all on the critical path of the next stage: ahead of the MFMA stream
        const int pt = tid - 256, q = pt & 3, prow = pt >> 2;        // piece q of halo pixels / weight rows prow + 64 i
        constexpr int NHV = (HH * HWD + 63) / 64;
        uint4 rh[NHV], rw[TAPS];
        int h_c[NHV];                                                // hy << 8 | hx, -1 past the halo
#pragma unroll
        for (int i = 0; i < NHV; ++i) {
            const int px = prow + 64 * i;
            h_c[i] = px < HH * HWD ? ((px / HWD) << 8) | (px % HWD) : -1;
        }
        // Per TILE: the 32-bit byte offset of every halo piece from the (uniform) image base, ~0u outside the image; per CHUNK
        // only the uniform bases advance by 64 bytes, so a stage costs the producers 15 loads + the zero selects and no address
        // arithmetic -- next to a consumer wave that issues MFMAs back to back they get ~3 VALU slots per MFMA, and the first
        // version (addresses recomputed per chunk) made the CONSUMERS wait 1.5 us per stage for them.
        unsigned hoff[NHV];
        const unsigned char* xb = nullptr;
        const unsigned char* wb = nullptr;
        const unsigned wlane = (unsigned)(prow * p.Cin) * 2u + q * 16;
        const unsigned cps = (unsigned)p.Cin * 2u;
        const size_t wstep = (size_t)p.CoutPad * cps;
        unsigned okmask = 0;
        int bng = 0, bng_next = 0;
        auto setup = [&](int item) {
            int tile, cb;
            decode(item, tile, cb);
            int t = tile;
            const int tx = t % p.tilesX; t /= p.tilesX;
            const int ty = t % p.tilesY;
            const int n = t / p.tilesY;
            bng_next = INBN ? n / (p.N / p.in_groups) : 0;
            const int iy0 = ty * WS_TH - p.pad, ix0 = tx * TW - p.pad;
            xb = reinterpret_cast<const unsigned char*>(p.x) + (size_t)n * p.H * p.W * cps;
            wb = reinterpret_cast<const unsigned char*>(p.w) + (size_t)cb * NT * cps;
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                const int iy = iy0 + (h_c[i] >> 8), ix = ix0 + (h_c[i] & 0xff);
                const bool ok = h_c[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                hoff[i] = ok ? (unsigned)(iy * p.W + ix) * cps + q * 16 : ~0u;
            }
        };
        auto fetch = [&](int c0) {
            const unsigned char* xc = xb + c0 * 2;
            const unsigned char* wc = wb + c0 * 2;
            bng = bng_next;
            okmask = 0;
#pragma unroll
            for (int i = 0; i < NHV; ++i) {
                const bool ok = hoff[i] != ~0u;
                const uint4 v = *reinterpret_cast<const uint4*>(xc + (ok ? hoff[i] : 0u));
                rh[i] = ok ? v : make_uint4(0, 0, 0, 0);
                okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < TAPS; ++i) rw[i] = *reinterpret_cast<const uint4*>(wc + (size_t)i * wstep + wlane);
        };
        auto commit = [&](int stage, int c0) {
            unsigned char* halo = smem + stage * STAGE;
            unsigned char* wts = halo + HH * HWD * HL::PITCH;
            if constexpr (INBN) {
                float sc[8], sh[8];
                const float* sp = bnp + bng * 2 * p.Cin + c0 + q * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                    *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + p.Cin + j);
                }
#pragma unroll
                for (int i = 0; i < NHV; ++i) {
                    if (!((okmask >> i) & 1u)) continue;          // padding of the post-activation tensor stays zero
                    float v[8];
                    unpack16(rh[i], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                    rh[i] = pack16<bf16>(v);
                }
            }
#pragma unroll
            for (int i = 0; i < NHV; ++i)
                if (h_c[i] >= 0) *reinterpret_cast<uint4*>(halo + HL::off(prow + 64 * i, q)) = rh[i];
#pragma unroll
            for (int i = 0; i < TAPS; ++i) *reinterpret_cast<uint4*>(wts + wt_off(i * NT + prow, q)) = rw[i];
        };
        int item = blockIdx.x, c = 0, stage = 0;
        bool first = true;
        WS_T(long long ts[4] = {0, 0, 0, 0}; const long long tbeg = WS_NOW();)
        // The persistent workgroups march through their chunks in lock step, and the 64 workgroups of one output-channel block
        // read the SAME weight lines: started on the same chunk they hit the same L2 banks at once (measured: 2.7 us from load
        // issue to data, the consumers waiting 1.5 us per stage).  Each workgroup therefore starts its channel loop at another
        // chunk (the sum over chunks is re-ordered per workgroup: fp32 accumulation, deterministic for a given launch).
        const int crot = (int)(blockIdx.x / 8) % nchunks;
        if (item < nitems) { setup(item); fetch(crot * 32); }
        while (item < nitems) {
            WS_T(long long t0 = WS_NOW();)
            commit(stage, ((c + crot) % nchunks) * 32);
            WS_T(long long t1 = WS_NOW(); ts[0] += t1 - t0;)
            int nitem = item, nc = c + 1;
            if (nc == nchunks) { nc = 0; nitem = item + gridDim.x; if (nitem < nitems) setup(nitem); }
            if (nitem < nitems) fetch(((nc + crot) % nchunks) * 32);
            WS_T(t0 = WS_NOW(); ts[1] += t0 - t1;)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c == 0 && !first) __builtin_amdgcn_s_barrier();      // E of the previous tile (its statistics combine)
            __builtin_amdgcn_s_barrier();                            // B: this chunk is staged
            WS_T(ts[2] += WS_NOW() - t0; ts[3] += 1;)
            first = false;
            stage ^= 1;
            item = nitem;
            c = nc;
        }
        if (!first) __builtin_amdgcn_s_barrier();                    // E of the last tile
        WS_T(if (tid == 256 && blockIdx.x < 1024) { long long* o = g_cws + blockIdx.x * 16 + 8; o[0] = ts[0]; o[1] = ts[1]; o[2] = ts[2]; o[3] = ts[3]; o[4] = WS_NOW() - tbeg; })
        return;
    }

    // ================= consumers: rows 4 wv .. 4 wv + 3 of the tile, all 64 output channels =================
    unsigned char* myot = otile + wv * OT_WAVE;
    int stage = 0;
    WS_T(long long tc[4] = {0, 0, 0, 0}; const long long tbeg = WS_NOW();)
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        int tile, cb;
        decode(item, tile, cb);
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
        const int oy0 = ty * WS_TH, ox0 = tx * TW, co0 = cb * NT;
        f32x4 acc[NS][RW];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < RW; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nchunks; ++c) {
            WS_T(long long t0 = WS_NOW();)
            __builtin_amdgcn_s_barrier();
            WS_T(long long t1 = WS_NOW(); tc[0] += t1 - t0; tc[3] += 1;)
            const unsigned char* halo = smem + stage * STAGE;
            const unsigned char* wts = halo + HH * HWD * HL::PITCH;
            stage ^= 1;
            // taps column-major (kw outer): a pixel fragment (halo row h of column kw) is read once and reused by every (r, kh)
            // that lands on it; the fragments of step i + 1 are issued before the MFMAs of step i (as in conv_mfma_kernel)
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
                        B[kw][h].u = *reinterpret_cast<const uint4*>(halo + HL::off((RW * wv + h) * HWD + pl + kw * DIL, g));
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) A[step & 1][s].u = *reinterpret_cast<const uint4*>(wts + wt_off(tap * NT + s * 16 + pl, g));
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's reads of the stage are complete
            WS_T(tc[1] += WS_NOW() - t1;)
        }
        WS_T(const long long te = WS_NOW();)

        // ---- epilogue of this wave's 4 rows: + bias, + residual, ReLU, statistics, transpose, 16-byte stores ----
        bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)n * p.OH * p.OW * p.Cout;
        const bf16* rin = p.res ? reinterpret_cast<const bf16*>(p.res) + (size_t)n * p.OH * p.OW * p.Cout : nullptr;
        float ssum[NS][4], ssq[NS][4], bs[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = co0 + s * 16 + g * 4 + j;
                ssum[s][j] = ssq[s][j] = 0.f;
                bs[s][j] = (p.bias && ch < p.Cout) ? p.bias[ch] : 0.f;
            }
        const bool relu = p.act == DH_ACT_RELU;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int oy = oy0 + RW * wv + r, ox = ox0 + pl;
            const bool pvalid = oy < p.OH && ox < p.OW;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[s][r][j] + bs[s][j];
                const int ch = co0 + s * 16 + g * 4;
                if (rin && pvalid && ch < p.Cout) {
                    float rr[4];
                    ld4(rin + (size_t)(oy * p.OW + ox) * p.Cout + ch, rr);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += rr[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (relu) v[j] = fmaxf(v[j], 0.f);
                    const float m = pvalid ? v[j] : 0.f;           // channels beyond Cout: zero weights and bias add 0
                    ssum[s][j] += m;
                    ssq[s][j] += m * m;
                }
                st4(reinterpret_cast<bf16*>(myot + (r * TW + pl) * TPITCH) + s * 16 + g * 4, v);
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the wave's tile rows (and statistics) are in LDS
        __builtin_amdgcn_s_barrier();                                // E (the producers meanwhile staged the next tile's chunk 0)
#pragma unroll
        for (int k = 0; k < RW * TW * 8 / 64; ++k) {
            const int i = lane + 64 * k, px = i >> 3, piece = i & 7;
            const int oy = oy0 + RW * wv + (px >> 4), ox = ox0 + (px & 15), ch = co0 + piece * 8;
            if (oy < p.OH && ox < p.OW && ch < p.Cout)
                *reinterpret_cast<uint4*>(yout + (size_t)(oy * p.OW + ox) * p.Cout + ch) =
                    *reinterpret_cast<const uint4*>(myot + px * TPITCH + piece * 16);
        }
        if (p.stats && tid < 2 * NT) {
            const int which = tid / NT, ch = tid - which * NT;
            const float tsum = red[(0 * 2 + which) * NT + ch] + red[(1 * 2 + which) * NT + ch] + red[(2 * 2 + which) * NT + ch] +
                               red[(3 * 2 + which) * NT + ch];
            if (co0 + ch < p.CoutPad) p.stats[((size_t)which * p.CoutPad + co0 + ch) * ntile + tile] = tsum;      // [2][CoutPad][tiles]
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // reads of otile / red done before the next tile's epilogue writes
        WS_T(tc[2] += WS_NOW() - te;)
    }
    WS_T(if (tid == 0 && blockIdx.x < 1024) { long long* o = g_cws + blockIdx.x * 16; o[0] = tc[0]; o[1] = tc[1]; o[2] = tc[2]; o[3] = tc[3]; o[4] = WS_NOW() - tbeg; })
}
#ifdef DH_WS_TIMING
extern "C" int dh_debug_cws(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cws), (size_t)n * 8); }
#endif

template <int DIL, bool INBN>
int launch_ws(const ConvArgs& a, hipStream_t st) {
    constexpr int HH = (WS_TH - 1) + 2 * DIL + 1, HWD = (TW - 1) + 2 * DIL + 1;
    const size_t lds = 2 * ((size_t)HH * HWD * HaloLayout<1>::PITCH + (size_t)9 * WS_NT * WPITCH) + 4 * (size_t)(4 * TW * (WS_NT * 2 + 16)) +
                       (size_t)4 * 2 * WS_NT * 4 + (INBN ? (size_t)a.in_groups * 2 * a.Cin * 4 : 0);
    static bool attr_done = false;
    static int cus = 0;
    if (!attr_done) {
        attr_done = true;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws_kernel<DIL, INBN>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv3x3_ws: cannot raise dynamic LDS to 160 KB");
        }
    }
    const int ntile = a.N * a.tilesX * a.tilesY, ncb = a.CoutPad / WS_NT, nitems = ntile * ncb;
    const int grid = nitems < cus ? nitems : cus;           // one persistent workgroup per CU
    hipLaunchKernelGGL((conv3x3_ws_kernel<DIL, INBN>), dim3(grid), dim3(512), lds, st, a, ntile, ncb);
    DH_CHECK_LAUNCH("conv3x3_ws");
    return 0;
}

}  // namespace

// eligibility: exactly the launches that conv_mfma_kernel<bf16, 3, 1, 64, 4, DIL, true, true, INBN> (compact epilogue) serves
bool dh_conv_ws_eligible(const ConvArgs& a, int ks, int stride, int dtype) {
    // OFF by default (DAHITRA_CONV_WS=1 turns it on): measured SLOWER than the two-workgroups-per-CU form -- layer3 114 vs 76 us,
    // layer2 37 vs 26 us.  tools/conv_ws_timeline.py: the consumers wait 1.5 us per stage for the producers, whose 15 loads per
    // stage take 2.7 us from issue to data whatever the address arithmetic (hoisted), the chunk order (rotated per workgroup) or
    // the wave priority: four producer waves keep one 57.6 KB batch of half-used cache lines (64 of 128 B per pixel / weight row
    // and chunk) in flight per CU = 19 GB/s, where two independent workgroups keep two = 42 GB/s.  The L2 -> LDS fill rate per CU,
    // not the barrier structure, bounds this kernel; the weight gradient (full 128-byte rows, distinct data per workgroup) is the
    // case where the same split wins (conv_wgrad_ws_kernel).
    static const bool off = getenv("DAHITRA_CONV_WS") == nullptr;
    if (off || dtype != DH_DTYPE_BF16 || ks != 3 || stride != 1 || a.rw != 4 || (a.dil != 1 && a.dil != 2)) return false;
    if (a.Cin % 32 || a.Cin < 128 || a.CoutPad % WS_NT || a.Cout % 8 || a.phase_mode || a.gate_y || a.y2 || a.y_nchw || a.w_nstride || a.w_cm) return false;
    if (a.act == DH_ACT_GELU || a.npix != a.OH * a.OW || a.in_npix != a.H * a.W || a.pad != a.dil || a.OH != a.H || a.OW != a.W) return false;
    if (a.in_scale && (a.dil != 1 || (size_t)a.in_groups * 2 * a.Cin * 4 > 6144)) return false;
    return true;
}
int dh_conv_ws_launch(const ConvArgs& a, hipStream_t st) {
    if (a.in_scale) return launch_ws<1, true>(a, st);
    return a.dil == 2 ? launch_ws<2, false>(a, st) : launch_ws<1, false>(a, st);
}
