// 1x1 / stride-1 convolution with >= 64 input channels as a K-DEEP GEMM on the CDNA4 matrix cores (bf16):
//
//   Y[pixel][cout] = act( sum_k X[pixel][k] * W[cout][k] + bias[cout] + residual[pixel][cout] )
//
// The ResNet-50 trunk (models/resnet.py:76-122, BASELINE configs[4]) spends two thirds of its FLOPs in such layers with
// K = 64 .. 1024.  The tap-oriented direct-conv kernel (conv_mfma_impl.h) stages 32 channels per barrier pair, i.e. 8
// MFMAs per wave between barriers for a 1x1 layer: 10 % of the MFMA peak.  Here a 256-thread workgroup owns a 128-pixel x
// BN-cout tile (BN = 128 or 64), stages 64 channels (128 B rows) per step for both operands with direct-to-LDS loads
// (global_load_lds_dwordx4: no staging registers, no ds_write pass), double-buffered, the next step's loads in flight
// across the barrier (counted vmcnt), 32 MFMAs (v_mfma_f32_16x16x32_bf16) per wave and step.  Both operands are
// k-contiguous in memory (NHWC activations, [cout][cin] weights), so the LDS image of a step is [row][128 B] filled in
// linear order with the 16-byte chunk index XOR-swizzled on the SOURCE address (chunk ^= row & 7) and on the fragment
// reads (cdna_hip_programming.md rule 21): ds_read_b128 without the 8-way conflict of a linear 128-byte pitch.
// Weights are the MFMA A operand (a lane ends with 4 consecutive couts of one pixel), as in conv_mfma_impl.h, and the
// epilogue is that kernel's compact one: +bias, +residual, ReLU, per-tile BatchNorm partial sums in the same
// [2][CoutPad][tiles] layout (tile = 128 consecutive pixels = the 8x16 tile count of the direct kernel, which the host
// guarantees by OH % 8 == 0, OW % 16 == 0), LDS-transposed 16-byte stores.
//
// Tile size is the lever (round 5): every staged byte crosses the L2 -> LDS fill path, which delivers ~40 GB/s per CU (~10 TB/s
// over the chip), and a BM x BN tile does BM * BN / (BM + BN) FLOP per staged byte-pair: 64 for 128 x 128 (ceiling ~650
// TFLOP/s -- rocprofv3 showed the ResNet-50 launches at 565), 85 for 256 x 128, 128 for 256 x 256.  The layers with >= 256
// pixels x {256, 128} output channels per tile run 512-thread workgroups on 256-pixel tiles (one per CU: 128 KB of stages);
// the per-wave work (64 couts x 64 or 128 pixels) and the LDS image are unchanged.  Measured (tools/gemm1x1_bench.py, 16 x 128 x 128
// pixels): 1024 -> 256: 245 -> 214 us, 256 -> 1024: 311 -> 289, 512 -> 1024: 453 -> 423.  What did NOT help, each built, verified
// against torch and timed (experiments/conv1x1_gemm_deep.inc): 3 - 5 stages of 32 channels with one barrier per step (250 us),
// the same with the next step's fragments read during this step's MFMAs (256 - 288 us), a per-workgroup rotation of the K
// order (211 vs 214).  The streaming shapes track the row pitch of the activation -- 6.6 TB/s algorithmic at 128-byte rows
// (64 -> 64), 5.4 at 512 B, 4.2 at 1 KB, 3.1 at 2 KB (1024 -> 256) -- i.e. what one step fetches per row (128 B) against the
// DRAM page it opens, not the loop structure, bounds the K >= 512 layers.
#include "conv_mfma_impl.h"

namespace {

constexpr int GBM = 128;          // pixels per workgroup
constexpr int GBK = 64;           // channels per step (128-byte rows)

__device__ __forceinline__ void glds16(const unsigned char* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// +bias, +residual, ReLU, BN partial sums (one per 128-pixel half of the tile), LDS-transposed 16-byte stores; the staging LDS
// must be free (all waves past their last fragment read)
template <int BN, int BM, int NT_>
__device__ __forceinline__ void gemm_epilogue(const ConvArgs& p, f32x4 (&acc)[4][NT_], unsigned char* smem, long m0, int co0, int pt,
                                              int npt, int tid, int lane, int wv, int wc, int wp) {
    constexpr int GBM = BM, NW = BM / 32, NTHR = 64 * NW, WC = BN / 64, WP = NW / WC;
    const int pl = lane & 15, g = lane >> 4;
    // ---- epilogue (the staging LDS is free): +bias, +residual, ReLU, BN partial sums, transposed 16-byte stores ----
    constexpr int TPITCH = BN * 2 + 16;
    unsigned char* otile = smem + NW * 2 * BN * 4;       // below it: the statistics scratch [NW waves][2][BN]
    bf16* yout = reinterpret_cast<bf16*>(p.y);
    const bf16* rin = reinterpret_cast<const bf16*>(p.res);
    const bool relu = p.act == DH_ACT_RELU;
    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[s][j] = 0.f; ssq[s][j] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = co0 + wc * 64 + s * 16 + g * 4;
        float bs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[j] = p.bias ? p.bias[c + j] : 0.f;
#pragma unroll
        for (int t = 0; t < NT_; ++t) {
            const int px = wp * (GBM / WP) + t * 16 + pl;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[s][t][j] + bs[j];
            if (rin) {
                float rr[4];
                ld4(rin + (size_t)(m0 + px) * p.Cout + c, rr);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += rr[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (relu) v[j] = fmaxf(v[j], 0.f);
                ssum[s][j] += v[j];
                ssq[s][j] += v[j] * v[j];
            }
            st4(reinterpret_cast<bf16*>(otile + px * TPITCH) + (c - co0), v);
        }
    }
    if (p.stats) {
        float* red = reinterpret_cast<float*>(smem);     // [NW waves][2][BN]
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row16_sum(ssum[s][j]), b = row16_sum(ssq[s][j]);
                if (pl == 0) {
                    red[(wv * 2 + 0) * BN + wc * 64 + s * 16 + g * 4 + j] = a;
                    red[(wv * 2 + 1) * BN + wc * 64 + s * 16 + g * 4 + j] = b;
                }
            }
    }
    __syncthreads();
    constexpr int PPR = BN * 2 / 16;                      // 16-byte pieces per pixel
    for (int i = tid; i < GBM * PPR; i += NTHR) {
        const int px = i / PPR, q = i - px * PPR;
        *reinterpret_cast<uint4*>(yout + (size_t)(m0 + px) * p.Cout + co0 + q * 8) =
            *reinterpret_cast<const uint4*>(otile + px * TPITCH + q * 16);
    }
    // one partial per 128-pixel HALF of the tile: the [2][CoutPad][tiles] layout counts 128-pixel tiles whatever BM is
    constexpr int HALVES = GBM / 128, WPH = WP / HALVES;      // pixel-wave rows per half
    static_assert(WP % HALVES == 0, "a wave's pixels lie in one 128-pixel half");
    if (p.stats) {
        const float* red = reinterpret_cast<const float*>(smem);
        for (int i = tid; i < HALVES * 2 * BN; i += NTHR) {
            const int half = i / (2 * BN), which = (i / BN) & 1, c = i % BN;
            // waves that hold couts [wc * 64, +64) of this half: wv = wp * WC + wc for wp in [half * WPH, +WPH), in wave order
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < WPH; ++q) t += red[(((half * WPH + q) * WC + c / 64 % WC) * 2 + which) * BN + c];
            p.stats[((size_t)which * p.CoutPad + co0 + c) * ((size_t)npt * HALVES) + (size_t)pt * HALVES + half] = t;
        }
    }
}

// XDEEP (256-pixel tiles): the ACTIVATION rows are staged three steps deep, the weights two.  The activation is what comes from
// HBM (every byte once, ~2 us away); the weights come from L2.  With both two deep a CU has one 32 KB activation stage in
// flight, bursty; with three, two are in flight at any time.  Measured: 222 -> 214 us on the pure-streaming 1024 -> 256 shape and
// nothing elsewhere -- every K >= 512 shape sits at ~3.3 us per 64-channel step whether its activation comes from HBM or from
// L2 (512 -> 1024: four cout tiles per pixel tile), i.e. the two-barrier step itself (LDS fragment reads, then MFMAs, all
// eight waves in lock step) is the bound, ~650 TFLOP/s, as the guide's tile table says of this structure (792 TFLOP/s at
// 256 x 256 on L2-resident operands).  (A plain streaming kernel reads HBM at 6 TB/s for any row pitch and any bytes per row
// and step: experiments/microbench/dram_pitch.hip -- the access pattern is not the problem either.)  Loads
// retire in issue order per wave (vmcnt), so a step issues the weights of step k + 1 BEFORE the activation of step k + 2: the
// wait for step k's data then leaves the younger activation loads outstanding.
template <int BN, int BM = GBM, bool XDEEP = false>
__global__ __launch_bounds__(2 * BM, 256 / BM) void conv1x1_gemm_kernel(ConvArgs p) {
    constexpr int GBM = BM;                              // pixels per workgroup (128: 4 waves, 256: 8 waves)
    constexpr int NW = BM / 32, NTHR = 64 * NW;
    constexpr int ROWS = BN + GBM;                       // LDS rows per stage: weights first, then pixels
    constexpr int STAGE = ROWS * 128;                    // bytes
    constexpr int NI = ROWS / (8 * NW);                  // glds instructions per wave and stage (8 rows each)
    constexpr int WC = BN / 64;                          // waves along cout (1, 2 or 4)
    constexpr int WP = NW / WC;                          // waves along pixels
    constexpr int NT_ = GBM / WP / 16;                   // 16-pixel sub-tiles per wave (2, 4 or 8)
    static_assert(ROWS % (8 * NW) == 0 && WP >= 1 && (GBM / WP) % 16 == 0 && (128 % (GBM / WP) == 0 || (GBM / WP) % 128 == 0), "tile / wave layout");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const int wc = wv % WC, wp = wv / WC;
    // Workgroup -> (pixel tile, cout tile).  The cout tiles of ONE pixel tile must run back to back on ONE XCD: they re-read
    // the same 128 x Cin activation tile, which then comes from that XCD's L2 instead of HBM (with the cout tile as the slow
    // grid dimension a 256 -> 1024 layer fetched its 134 MB input eight times).  Dispatch is round-robin over the 8 XCDs
    // (block b -> XCD b % 8: a speed assumption only), so XCD x walks pixel tiles x, x + 8, ... with the cout tile fastest.
    const int nco = p.Cout / BN, npt = (int)gridDim.x / nco;
    int pt, cot;
    if ((npt & 7) == 0) {
        const int b = blockIdx.x, q = b >> 3;
        cot = q % nco;
        pt = (q / nco) * 8 + (b & 7);
    } else {
        cot = blockIdx.x % nco;
        pt = blockIdx.x / nco;
    }
    const long m0 = (long)pt * GBM;                      // first pixel (flattened over N, OH, OW)
    const int co0 = cot * BN;
    const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* Wt = reinterpret_cast<const unsigned char*>(p.w);
    const unsigned rowbytes = (unsigned)p.Cin * 2u;

    // per-lane source row / swizzled chunk of each of this wave's NI loads (the same for every step)
    const unsigned char* src[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = (j * NW + wv) * 8 + (lane >> 3);                      // LDS row
        const int c = (lane & 7) ^ (r & 7);                                  // source chunk landing in LDS chunk lane & 7
        // (a stride-2 layer -- the shortcut of a stride-2 Bottleneck -- gathers its pixel rows: output pixel (n, oy, ox) reads the
        // input row (n, 2 oy, 2 ox); every lane supplies its own source address anyway)
        long xrow = m0 + r - BN;
        if (r >= BN && (p.H != p.OH || p.W != p.OW)) {
            const long ohw = (long)p.OH * p.OW, n = xrow / ohw, rem = xrow - n * ohw;
            const long oy = rem / p.OW, ox = rem - oy * p.OW;
            xrow = (n * p.H + 2 * oy) * p.W + 2 * ox;
        }
        src[j] = (r < BN ? Wt + (size_t)(co0 + r) * rowbytes : X + (size_t)xrow * rowbytes) + c * 16;
    }
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int j = 0; j < NI; ++j) glds16(src[j] + (size_t)k0 * 2, smem + stage * STAGE + (j * NW + wv) * 1024);
    };
    // XDEEP: weight rows = instructions j < NIW of a wave (rows [0, BN)), activation rows the others; their own stage rings
    constexpr int NIW = BN / (8 * NW), NIX = GBM / (8 * NW);
    constexpr int WSTAGE = BN * 128, XSTAGE = GBM * 128, XBASE = 2 * WSTAGE;         // LDS: [2 weight stages][3 activation stages]
    static_assert(!XDEEP || (NIW >= 1 && NIX >= 1 && NIW + NIX == NI), "a wave's loads split into whole weight / activation instructions");
    auto issue_w = [&](int kt) {
#pragma unroll
        for (int j = 0; j < NIW; ++j) glds16(src[j] + (size_t)kt * (GBK * 2), smem + (kt & 1) * WSTAGE + (j * NW + wv) * 1024);
    };
    auto issue_x = [&](int kt) {
#pragma unroll
        for (int j = NIW; j < NI; ++j)
            glds16(src[j] + (size_t)kt * (GBK * 2), smem + XBASE + (kt % 3) * XSTAGE + ((j - NIW) * NW + wv) * 1024);
    };

    f32x4 acc[4][NT_];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < NT_; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.Cin / GBK;
    if constexpr (XDEEP) {
        issue_w(0);
        issue_x(0);
        if (KT > 1) issue_x(1);
    } else {
        issue(0, 0);
    }
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if constexpr (XDEEP) {
            // issue order [W(kt+1), X(kt+2)] after [.., W(kt), X(kt+1)]: everything up to X(kt) and W(kt) has landed when only
            // X(kt+1), W(kt+1), X(kt+2) are outstanding
            if (more) issue_w(kt + 1);
            if (kt + 2 < KT) issue_x(kt + 2);
            if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIX + NIW + NIX) : "memory");
            else if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIX + NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (more) issue((kt + 1) & 1, (kt + 1) * GBK);
            // this wave's loads of step kt have landed when at most the NI just-issued ones are outstanding
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NI) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                    // (raw: a __syncthreads would drain the loads in flight)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sa = XDEEP ? smem + (kt & 1) * WSTAGE : smem + (kt & 1) * STAGE;
        const unsigned char* sb = XDEEP ? smem + XBASE + (kt % 3) * XSTAGE : sa + BN * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + g;
            V16u A[4], B[NT_];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int r = wc * 64 + s * 16 + pl;
                A[s].u = *reinterpret_cast<const uint4*>(sa + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int t = 0; t < NT_; ++t) {
                const int r = wp * (GBM / WP) + t * 16 + pl;
                B[t].u = *reinterpret_cast<const uint4*>(sb + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < NT_; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s].h, B[t].h, acc[s][t], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave is done reading this stage
        asm volatile("" ::: "memory");
    }

    gemm_epilogue<BN, BM>(p, acc, smem, m0, co0, pt, npt, tid, lane, wv, wc, wp);
}


template <int BN, int BM = GBM, bool XDEEP = false>
int launch_gemm(const ConvArgs& a, hipStream_t st) {
    constexpr int GBM = BM;
    constexpr int ROWS = BN + GBM;
    const size_t staging = XDEEP ? (size_t)(2 * BN + 3 * GBM) * 128 : (size_t)2 * ROWS * 128;
    const size_t otile = (size_t)(BM / 32) * 2 * BN * 4 + (size_t)GBM * (BN * 2 + 16);
    const size_t lds = staging > otile ? staging : otile;
    auto kern = conv1x1_gemm_kernel<BN, BM, XDEEP>;
    static bool attr_done = false;
    if (lds > 64 * 1024 && !attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv1x1_gemm: cannot raise dynamic LDS to %zu", lds);
        }
    }
    const long M = (long)a.N * a.OH * a.OW;
    hipLaunchKernelGGL(kern, dim3((unsigned)((M / GBM) * (a.Cout / BN))), dim3(2 * BM), lds, st, a);
    DH_CHECK_LAUNCH("conv1x1_gemm");
    return 0;
}

}  // namespace

// the shapes this kernel serves (everything else stays on the direct kernel); bf16 only
bool dh_conv1x1_gemm_eligible(const ConvArgs& a, int ks, int stride, int dtype) {
    static const bool off = getenv("DAHITRA_NO_GEMM1X1") != nullptr;
    static const bool no_s2 = getenv("DAHITRA_GEMM1X1_NO_S2") != nullptr;       // A/B switch: stride-2 layers stay on the direct kernel
    const bool geo = stride == 1 ? (a.H == a.OH && a.W == a.OW)
                                 : (stride == 2 && !no_s2 && a.OH == (a.H + 1) / 2 && a.OW == (a.W + 1) / 2 && a.in_npix == a.H * a.W);
    return !off && dtype == DH_DTYPE_BF16 && ks == 1 && geo && a.pad == 0 && a.Cin >= 64 && a.Cin % GBK == 0 &&
           a.Cout % 64 == 0 && a.CoutPad == a.Cout && a.OH % 8 == 0 && a.OW % 16 == 0 &&
           !a.gate_y && !a.y2 && !a.in_scale && a.w_nstride == 0 && a.npix == a.OH * a.OW && a.act != DH_ACT_GELU &&
           !(a.stats && (a.res || a.act != DH_ACT_NONE)) && ((long)a.N * a.OH * a.OW) % GBM == 0;
}
int dh_conv1x1_gemm_launch(const ConvArgs& a, hipStream_t st) {
    static const bool small = getenv("DAHITRA_GEMM1X1_SMALL") != nullptr;       // A/B switch: 128-pixel tiles only
    const long M = (long)a.N * a.OH * a.OW;
    // 256-pixel tiles where they still give the chip at least two rounds of workgroups
    static const bool shallow = getenv("DAHITRA_GEMM1X1_SHALLOW") != nullptr;   // A/B switch: activation two stages deep as the weights
    if (!small && M % 256 == 0 && a.Cout % 128 == 0 && (M / 256) * (a.Cout / (a.Cout % 256 == 0 ? 256 : 128)) >= 512) {
        if (shallow) return a.Cout % 256 == 0 ? launch_gemm<256, 256>(a, st) : launch_gemm<128, 256>(a, st);
        // (the three-deep activation ring: 222 -> 214 us on 1024 -> 256, neutral on the others, slower at 128 couts: 83 -> 86)
        return a.Cout % 256 == 0 ? launch_gemm<256, 256, true>(a, st) : launch_gemm<128, 256>(a, st);
    }
    return a.Cout % 128 == 0 ? launch_gemm<128>(a, st) : launch_gemm<64>(a, st);
}
