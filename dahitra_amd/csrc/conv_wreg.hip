// 3x3 / stride-1 / pad-1 NHWC convolution (bf16) with the WEIGHTS RESIDENT IN REGISTERS.
//
// The tap-oriented kernel (conv_mfma_impl.h) stages, per 32-channel chunk and per pixel tile, the weights of all nine taps
// next to the haloed input: 36.9 KB of weights for 11.5 - 20.7 KB of pixels, i.e. 100 - 170 bytes through the L2 -> LDS fill
// path per MFMA -- and that fill path, not the matrix pipe, bounds it (0.25 - 0.44 of the bf16 peak).
// Here the roles are turned around.  A wavefront owns 16 * NSUB output channels for ALL input channels and keeps their
// weights -- 9 taps x NCH chunks x NSUB fragments = 36 * NSUB * NCH registers per lane -- in the (512-entry, unified)
// register file for the life of a PERSISTENT workgroup; the four wavefronts of a workgroup cover 64 * NSUB output channels
// and walk through 8x16-pixel tiles.  Only the input halo travels: 11.5 KB per 32-channel chunk, 20 bytes per MFMA, by
// direct-to-LDS loads (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of stages that runs
// WR_D - 2 stages ahead of the matrix work behind counted vmcnt waits and ONE raw s_barrier per stage.  The A operand of
// every MFMA is a register-resident weight fragment, the B operand a halo fragment read once from LDS and used by up to
// 3 * NSUB MFMAs (the three kernel rows that land on that halo row).
//
// Same arithmetic as conv_mfma_kernel<bf16, 3, 1, 64, ...>: the same fragments, and every accumulator receives its products
// in the same (chunk, kernel column, kernel row) order, so the results are bit-identical; same compact epilogue (+bias,
// +residual, ReLU, BatchNorm partial sums in the [2][CoutPad][tiles] layout of dh_conv2d_fwd_num_tiles, LDS-transposed
// 16-byte stores) and the same BatchNorm-apply + ReLU on load (INBN: applied in LDS by the lane whose load brought the piece).
//
// Reference: the 3x3 convolutions of models/resnet.py:24-73 (BasicBlock conv1 / conv2) and their data gradients.
#include "conv_mfma_impl.h"

namespace {

constexpr int WR_TH = 8;                                   // tile rows (TW = 16 pixels wide)
constexpr int WR_HH = WR_TH + 2, WR_HW = TW + 2;           // halo 10 x 18 pixels
constexpr int WR_NPX = WR_HH * WR_HW;                      // 180
constexpr int WR_NI = 3;                                   // 1-KiB load instructions per wave and chunk image: 12 KiB >= 180 * 64 B
#ifndef WR_EXP       // experiment builds (tools/wreg_bench.py --only): 1 = no ring loads in the stream, 2 = no stage barrier
#define WR_EXP 0    //  (wrong results; what the stream loop costs without them)
#endif
constexpr int WR_IMG = WR_NI * 4 * 1024;                   // bytes of one chunk image in the ring

__device__ __attribute__((aligned(256))) unsigned int wr_zero[64];      // source of every padding / out-of-stream piece

// One direct-to-LDS load instruction: lane l's 16 bytes at gsrc land at LDS byte address lds_wave_base + 16 l.  Inline asm,
// not __builtin_amdgcn_global_load_lds: hipcc tracks a builtin LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in
// front of the next ds_read of the (may-alias) dynamic LDS array -- which drains the run-ahead ring at every stage.  The
// loads are therefore invisible to the compiler's counters and are waited for by the counted vmcnt in the stream loop
// (cdna_hip_programming.md 5.7: M0 is written in the statement that reads it and restored).
__device__ __forceinline__ void wr_glds16(const unsigned char* gsrc, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
__device__ __forceinline__ unsigned wr_lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// 8-byte global load / its counted wait as inline asm: a residual row is fetched a whole stage before it is used, across the
// stream loop's back edge, where a compiler-tracked load would be waited for with vmcnt(0) -- a drain of the run-ahead ring.
// The wait takes the destination as an in/out operand, so nothing that reads it can be scheduled above the wait; and the
// destination is an ACCUMULATION register ("a": gfx950 loads into the unified file directly) because, with the vector half
// full of weights, the register allocator otherwise parks a freshly "defined" value there itself -- copying it out before
// the load has landed (seen in the ISA).  tools/check_wreg_isa.py verifies that nothing touches the destinations in between.
__device__ __forceinline__ void wr_gload8(const void* src, unsigned long long& d) {
    asm volatile("global_load_dwordx2 %0, %1, off" : "=a"(d) : "v"(src) : "memory");
}
template <int N>
__device__ __forceinline__ void wr_wait_for(unsigned long long& d) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+a"(d) : "n"(N) : "memory");
}
// hide a loop-invariant value from the optimiser, so that what is derived from it is recomputed where it is used instead of
// being hoisted into (scarce) registers for the life of the stream loop
__device__ __forceinline__ int wr_opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// halo pixel (hy, hx), 16-byte piece q of its 64-byte chunk row -> byte offset in a chunk image.  Pitch 64 with the piece
// index XORed by bit 2 of the COLUMN: a ds_read_b128 of 16 consecutive columns is conflict-free in every 16-lane service
// group (MI355X_MICROARCH.md, LDS table), and -- unlike the pixel-index swizzle of conv_mfma_impl.h -- the lane part of a
// fragment address does not depend on the row, so a halo row is an immediate offset.
__device__ __forceinline__ int wr_off(int hy, int hx, int q) { return (hy * WR_HW + hx) * 64 + ((q ^ (((hx >> 2) & 1) << 1)) << 4); }

// -DWR_TIMING: per-workgroup wall-clock stamps (100 MHz) for tools/wreg_timeline.py; experiment builds only
#ifdef WR_TIMING
__device__ long long wr_ts[4096 * 32];
#define WR_TS(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096 && (k) < 32) wr_ts[blockIdx.x * 32 + (k)] = (long long)wall_clock64(); } while (0)
#define WR_CYC(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) wr_ts[blockIdx.x * 32 + (k)] = (long long)clock64(); } while (0)
#else
#define WR_TS(k) do { } while (0)
#define WR_CYC(k) do { } while (0)
#endif

struct WrArgs {
    ConvArgs c;
    int ncb;          // output-channel blocks (Cout / (64 * NSUB))
    int J;            // workgroups per output-channel block
    int nunits;       // statistics units (= rows of the stats buffer): `subt` vertically adjacent tiles each
    int subt;         // 8-row tiles per unit: 1, or 2 where dh_conv2d_fwd_num_tiles counts 16-row tiles
    int tilesX, unitsY;
    int wfrag;        // weights in FRAGMENT order [Cout/16][Cin/32][tap][lane][8]: a wave's fragment is 1 KiB contiguous
};

struct WrTile {       // one 8x16 tile of the stream: per-lane source offsets of its WR_NI pieces (one chunk image)
    const unsigned char* img;      // image base (uniform)
    unsigned o0, o1, o2;           // byte offset of this lane's piece k at chunk 0, ~0u: padding (scalars, not an array: the
                                   // struct must stay in registers)
    int n, oy0, ox0;
    __device__ __forceinline__ unsigned off(int i) const { return i == 0 ? o0 : (i == 1 ? o1 : o2); }
};
static_assert(WR_NI == 3, "WrTile carries three piece offsets");

template <int NSUB, int NCH, int D, int PFD, bool INBN, int WPS, bool RES, bool RELU>
__global__ __launch_bounds__(256, WPS) void conv3x3_wreg_kernel(WrArgs a) {
    // One tile = NCH stages (one 32-channel chunk image each) = NCH * 30 steps; a step = one halo fragment (kernel column kw,
    // halo row h) read from LDS and the <= 3 * NSUB MFMAs it feeds.  The steps of the whole stream form ONE software pipeline:
    // the fragment of step g + PFD is read before the MFMAs of step g, across stage and tile boundaries, so the matrix pipe
    // does not drain at a stage change.  That works because stage t + 1 is PUBLISHED in the middle of stage t: at step SYNC of
    // every stage each wave waits for its own loads of the next stage (counted vmcnt), passes the one barrier of the stage,
    // and then refills the ring slot of the stage before (which every wave has left, or it would not be at this barrier)
    // with the stage D - 1 ahead.  Loads thus run D - 2 stages ahead of their publication.
    constexpr int NSTEP = 3 * WR_HH;                       // 30
    constexpr int TSTEPS = NCH * NSTEP;
    constexpr int NB = PFD + 1;                            // fragment registers (rolling)
    static_assert(TSTEPS % NB == 0, "the rolling fragment buffer must line up at a tile change");
    static_assert(D >= 3 && D - 1 <= 2 * NCH + 1, "ring depth");
    constexpr int SYNC = 14;                               // step of a stage at which the next stage is published
    constexpr int BN0 = 2;                                 // INBN: steps BN0 .. BN0 + 2 WR_NI carry the next stage's transform
    constexpr bool STATIC_SLOT = NCH % D == 0;
    constexpr int NCO = 64 * NSUB;                         // output channels per workgroup
    constexpr int TPITCH = NCO * 2 + 16;                   // transposed output tile: bytes per pixel
    constexpr int PPR = NCO * 2 / 16;                      // 16-byte pieces per output pixel
    const ConvArgs& p = a.c;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;                            // [D][WR_IMG]
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(wr_lds_addr(smem));
    unsigned char* otile = smem + D * WR_IMG;              // [128 px][TPITCH]
    float* bnp = reinterpret_cast<float*>(otile + WR_TH * TW * TPITCH);      // INBN: [in_groups][2][Cin]

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    WR_TS(0);
    // block -> (output-channel block, stream): the ncb workgroups that walk the SAME tiles sit on one XCD (dispatch is
    // round-robin over the 8 XCDs: a speed assumption only), so the input is fetched from HBM once and then from that L2
    const int b = blockIdx.x, xcd = b & 7, qb = b >> 3;
    const int cb = qb % a.ncb, j0 = (qb / a.ncb) * 8 + xcd;
    const int co_w = cb * NCO + wv * 16 * NSUB;            // first output channel of this wave
    const int Cin = p.Cin;
    const int xps = p.x_split ? Cin : Cin * 2;             // bytes per pixel of an input tensor (split input: Cin / 2 channels each)

    // ---- this wave's weights: [sub][chunk][tap] fragments, 16 bytes per lane each, loaded once ----
    s16x8 A[NSUB][NCH][9];
    {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    A[s][c][t] = *reinterpret_cast<const s16x8*>(
                        a.wfrag ? wb + ((size_t)(((co_w >> 4) + s) * NCH + c) * 9 + t) * 1024 + lane * 16
                                : wb + ((size_t)(t * p.CoutPad + co_w + s * 16 + pl) * Cin + c * 32 + g * 8) * 2);
    }
    if constexpr (INBN) {
        for (int i = tid; i < p.in_groups * 2 * Cin; i += 256) {
            const int gi = i / (2 * Cin), r = i - gi * 2 * Cin;
            bnp[i] = r < Cin ? p.in_scale[gi * Cin + r] : p.in_shift[gi * Cin + r - Cin];
        }
    }
    float bs[NSUB][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) bs[s][j] = p.bias ? p.bias[co_w + s * 16 + g * 4 + j] : 0.f;

    // ---- per-lane constants of the staging: piece i = (k * 4 + wave) * 64 + lane of a chunk image ----
    int hyx[WR_NI];        // (hy << 8) | hx, or -1 past the image;  logical piece in bits 16..17
#pragma unroll
    for (int k = 0; k < WR_NI; ++k) {
        const int i = (k * 4 + wv) * 64 + lane, px = i >> 2, qs = i & 3;
        const int hy = px / WR_HW, hx = px - hy * WR_HW;
        hyx[k] = px < WR_NPX ? (((qs ^ (((hx >> 2) & 1) << 1)) << 16) | (hy << 8) | hx) : -1;
    }
    // fragment read offsets of this lane for the three kernel columns (the halo row is an immediate)
    int lo[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) lo[kw] = wr_off(0, pl + kw, g);

    const int K = (a.nunits > j0 ? (a.nunits - j0 + a.J - 1) / a.J : 0) * a.subt;      // tiles of this workgroup's stream
    auto tile_desc = [&](int k, WrTile& t) {
        if (k >= K) {
            t.img = reinterpret_cast<const unsigned char*>(wr_zero);
            t.o0 = t.o1 = t.o2 = ~0u;
            t.n = 0; t.oy0 = 0; t.ox0 = 0;
            return;
        }
        const int u = j0 + (k / a.subt) * a.J, half = k % a.subt;
        const int tx = u % a.tilesX, r = u / a.tilesX, uy = r % a.unitsY, n = r / a.unitsY;
        t.n = n; t.oy0 = (uy * a.subt + half) * WR_TH; t.ox0 = tx * TW;
        t.img = reinterpret_cast<const unsigned char*>(p.x) + (size_t)n * p.H * p.W * xps;
        auto piece = [&](int code) {
            code = wr_opaque(code);
            const int hy = (code >> 8) & 0xff, hx = code & 0xff, q = (code >> 16) & 3;
            const int iy = t.oy0 - 1 + hy, ix = t.ox0 - 1 + hx;
            const bool ok = code >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            return ok ? (unsigned)(iy * p.W + ix) * (unsigned)xps + q * 16 : ~0u;
        };
        t.o0 = piece(hyx[0]); t.o1 = piece(hyx[1]); t.o2 = piece(hyx[2]);
    };
    // loads of chunk image `c` of tile t into ring slot `slot`: WR_NI instructions per wave.  EVERY lane of EVERY wave issues
    // EVERY instruction (padding and past-the-end pieces read the zero block), so the number of loads in flight is a
    // compile-time constant at every wait.  Split input (ConvArgs::x_split): the chunks of the upper channel half come from the
    // second tensor -- the same pixel offsets (both have Cin / 2 channels per pixel) from another, uniform, base.
    auto issue = [&](const WrTile& t, int c, int slot) {
        const long coff = (NCH >= 2 && p.x_split && c >= NCH / 2) ? p.x_split + (c - NCH / 2) * 64 : (long)c * 64;
#pragma unroll
        for (int i = 0; i < WR_NI; ++i) {
            const unsigned char* src = t.off(i) != ~0u ? t.img + t.off(i) + coff : reinterpret_cast<const unsigned char*>(wr_zero);
            wr_glds16(src, ring_lds + slot * WR_IMG + (i * 4 + wv) * 1024);
        }
    };
    // INBN: BatchNorm-apply + ReLU of the previous layer on chunk image c of tile t, in place, by the lane whose load brought
    // the piece (padding pieces stay zero: they are padding of the POST-activation tensor)
    // one piece at a time, in two halves that the stream loop places in DIFFERENT steps (the LDS reads of the first half are
    // under way while the MFMAs of a step issue; the second half's ~30 VALU instructions slot in between the MFMAs of a later
    // step): done back to back at the publication point it cost ~600 cycles per 1450-cycle stage (85.7 vs 60.0 us on 256 -> 256)
    uint4 bn_px;
    float bn_sc[8], bn_sh[8];
    auto bn_load = [&](const WrTile& t, int c, int slot, int i) {
        const int grp = t.n / (p.N / p.in_groups);
        const unsigned char* pc = ring + slot * WR_IMG + ((i * 4 + wv) * 64 + lane) * 16;
        const float* sp = bnp + grp * 2 * Cin + c * 32 + ((wr_opaque(hyx[i]) >> 16) & 3) * 8;
        *reinterpret_cast<float4*>(bn_sc) = *reinterpret_cast<const float4*>(sp);
        *reinterpret_cast<float4*>(bn_sc + 4) = *reinterpret_cast<const float4*>(sp + 4);
        *reinterpret_cast<float4*>(bn_sh) = *reinterpret_cast<const float4*>(sp + Cin);
        *reinterpret_cast<float4*>(bn_sh + 4) = *reinterpret_cast<const float4*>(sp + Cin + 4);
        bn_px = *reinterpret_cast<const uint4*>(pc);
    };
    auto bn_store = [&](const WrTile& t, int slot, int i) {
        if (t.off(i) == ~0u) return;                       // padding of the POST-activation tensor stays zero
        unsigned char* pc = ring + slot * WR_IMG + ((i * 4 + wv) * 64 + lane) * 16;
        float v[8];
        unpack16(bn_px, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] * bn_sc[e] + bn_sh[e], 0.f);
        *reinterpret_cast<uint4*>(pc) = pack16<bf16>(v);
    };
    auto bn_transform = [&](const WrTile& t, int c, int slot) {
#pragma unroll
        for (int i = 0; i < WR_NI; ++i) { bn_load(t, c, slot, i); bn_store(t, slot, i); }
    };

    f32x4 acc[NSUB][WR_TH];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < WR_TH; ++r) acc[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[NSUB][4], ssq[NSUB][4];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) { ssum[s][j] = 0.f; ssq[s][j] = 0.f; }

    // ---- prologue: stages 0 .. D - 2 of the stream in flight, stage 0 published ----
    WrTile t0, t1, t2;                                     // tiles k, k + 1, k + 2 of the stream
    tile_desc(0, t0);
    tile_desc(1, t1);
    tile_desc(2, t2);
    auto tile_at = [&](int dk) -> const WrTile& { return dk == 0 ? t0 : (dk == 1 ? t1 : t2); };
    int s0 = 0;                                            // ring slot of the current tile's stage 0 (constant 0 if NCH % D == 0)
    auto slot_of = [&](int st) { return STATIC_SLOT ? st % D : (s0 + st) % D; };
#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue(tile_at(s / NCH), s % NCH, s % D);
    // The weight and bias loads must be COMPLETE for the compiler's wait-count bookkeeping before the stream loop: a load
    // still pending at the loop header would put a vmcnt(0) -- a drain of the run-ahead ring -- in front of its first use
    // in every iteration.  An empty asm that takes each register as an in/out operand forces the one-time wait here.
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(A[s][c][t]));
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bs[s][j]));
    }
    WR_TS(1);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 2) * WR_NI) : "memory");
    if constexpr (INBN) {
        __syncthreads();                                   // bnp staged (every load is long complete: the weights were waited for)
        bn_transform(t0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    WR_TS(2);
    WR_CYC(28);                                            // shader cycles at the start / end of the stream (slots 28, 29)
    V16u B[NB];
    // fragment of step gg (0 .. TSTEPS + PFD - 1; past TSTEPS: the next tile's first stage) into its rolling register
    auto rd = [&](int gg) {
        const int st = gg / NSTEP, i = gg - st * NSTEP, kw = i / WR_HH, h = i - kw * WR_HH;
        B[gg % NB].u = *reinterpret_cast<const uint4*>(ring + slot_of(st) * WR_IMG + lo[kw] + h * (WR_HW * 64));
    };
#pragma unroll
    for (int gg = 0; gg < PFD; ++gg) rd(gg);

    // ---- the epilogue of a tile (+ bias, + residual, ReLU, statistics, transposed 16-byte stores), in pieces that ride in
    // the FIRST STAGE of the next tile: output row r right before step r (whose first MFMA restarts that row's accumulator
    // from zero), the statistics at step 8, the staged tile published by that stage's one barrier, the cooperative stores in
    // the steps after it -- the matrix pipe never waits for an epilogue (one wave per SIMD at 256 input channels: 1.1 us of
    // every 8 us tile before).  The residual rows of a tile are requested at the first step of its LAST stage (inline-asm
    // loads, see wr_gload8) and waited for row by row with the exact count of younger loads: the rows after it and the
    // WR_NI ring loads of that stage's publication point.
    constexpr bool relu = RELU, has_res = RES;             // (compile-time: as run-time flags they doubled the row's instructions)
    unsigned long long rr[NSUB][WR_TH];                    // residual rows of the tile whose epilogue is pending (4 bf16 each)
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int r = 0; r < WR_TH; ++r) rr[s][r] = 0;
    auto res_issue = [&](const WrTile& t) {
        const bf16* rin = reinterpret_cast<const bf16*>(p.res) + (size_t)t.n * p.OH * p.OW * p.Cout;
#pragma unroll
        for (int r = 0; r < WR_TH; ++r)
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
                wr_gload8(rin + (size_t)((t.oy0 + r) * p.OW + t.ox0 + pl) * p.Cout + co_w + s * 16 + wr_opaque(g) * 4, rr[s][r]);
    };
    int en = 0, eoy0 = 0, eox0 = 0;                        // the tile whose epilogue is pending
    constexpr int NST = WR_TH * TW * PPR / 256;            // cooperative store rounds
    static_assert((WR_TH * TW * PPR) % 256 == 0 && SYNC + NST + 1 < NSTEP && WR_TH < SYNC, "epilogue pieces fit the first stage");
    auto epi_row = [&](int r) {                            // (r is a constant after unrolling: the switch below folds)
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            float v[4], res4[4] = {0.f, 0.f, 0.f, 0.f};
            if constexpr (has_res) {
                if (s == 0) {                              // row r of every sub-block has landed
                    switch (r) {
                    case 0: wr_wait_for<7 * NSUB + WR_NI>(rr[0][0]); break;
                    case 1: wr_wait_for<6 * NSUB + WR_NI>(rr[0][1]); break;
                    case 2: wr_wait_for<5 * NSUB + WR_NI>(rr[0][2]); break;
                    case 3: wr_wait_for<4 * NSUB + WR_NI>(rr[0][3]); break;
                    case 4: wr_wait_for<3 * NSUB + WR_NI>(rr[0][4]); break;
                    case 5: wr_wait_for<2 * NSUB + WR_NI>(rr[0][5]); break;
                    case 6: wr_wait_for<1 * NSUB + WR_NI>(rr[0][6]); break;
                    default: wr_wait_for<WR_NI>(rr[0][7]); break;
                    }
                }
                const unsigned lo = (unsigned)rr[s][r], hi = (unsigned)(rr[s][r] >> 32);
                res4[0] = __uint_as_float(lo << 16); res4[1] = __uint_as_float(lo & 0xffff0000u);
                res4[2] = __uint_as_float(hi << 16); res4[3] = __uint_as_float(hi & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = acc[s][r][j] + bs[s][j];
                if constexpr (has_res) v[j] += res4[j];
                if constexpr (relu) v[j] = fmaxf(v[j], 0.f);
                ssum[s][j] += v[j];
                ssq[s][j] += v[j] * v[j];
            }
            st4(reinterpret_cast<bf16*>(otile + (r * TW + pl) * TPITCH) + wv * 16 * NSUB + s * 16 + g * 4, v);
        }
    };
    auto epi_stats = [&](int kk) {                         // kk: stream index of the tile whose rows were just added
        if (p.stats && (kk % a.subt) == a.subt - 1) {
            const int unit = j0 + (kk / a.subt) * a.J;
#pragma unroll
            for (int s = 0; s < NSUB; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sa = row16_sum(ssum[s][j]), sq = row16_sum(ssq[s][j]);
                    if (pl == 0) {
                        const int c = co_w + s * 16 + wr_opaque(g) * 4 + j;
                        p.stats[((size_t)0 * p.CoutPad + c) * a.nunits + unit] = sa;
                        p.stats[((size_t)1 * p.CoutPad + c) * a.nunits + unit] = sq;
                    }
                    ssum[s][j] = 0.f; ssq[s][j] = 0.f;
                }
        }
    };
    // round `it` of the staged tile (en, eoy0, eox0) -> y, 16 bytes per lane: read from LDS in one step, stored in the next
    uint4 ehold;
    auto epi_fetch = [&](int it) {
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        ehold = *reinterpret_cast<const uint4*>(otile + px * TPITCH + q * 16);
    };
    auto epi_store = [&](int it) {
        // split output (ConvArgs::y_split): this workgroup's channel block lies in one of two [N][OH][OW][Cout / 2] tensors
        const int half = p.Cout >> 1, upper = (p.y_split && cb * NCO >= half) ? 1 : 0;
        const int ypitch = p.y_split ? half : p.Cout, cbase = cb * NCO - upper * half;
        bf16* yout = reinterpret_cast<bf16*>(reinterpret_cast<unsigned char*>(p.y) + (upper ? p.y_split : 0L)) + (size_t)en * p.OH * p.OW * ypitch;
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        *reinterpret_cast<uint4*>(yout + (size_t)((eoy0 + (px >> 4)) * p.OW + eox0 + (px & 15)) * ypitch + cbase + q * 8) = ehold;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < K; ++k) {
        const bool pend = k > 0;
#pragma unroll
        for (int st = 0; st < NCH; ++st)
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int gg = st * NSTEP + i, kw = i / WR_HH, h = i - kw * WR_HH;
            if (st == 0 && i <= WR_TH && pend) {
                if (i < WR_TH) epi_row(i);
                else epi_stats(k - 1);
            }
            if constexpr (has_res) { if (st == NCH - 1 && i == 0) res_issue(t0); }
            if constexpr (INBN) {
                // BatchNorm + ReLU of the NEXT stage's image, spread over steps BN0 .. (see bn_load)
                const int nx = st + 1;
                if (i == BN0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 3) * WR_NI) : "memory");      // its loads have landed
                if (i >= BN0 && i < BN0 + 2 * WR_NI) {
                    const int k = (i - BN0) >> 1;
                    if (((i - BN0) & 1) == 0) bn_load(tile_at(nx / NCH), nx % NCH, slot_of(nx), k);
                    else bn_store(tile_at(nx / NCH), slot_of(nx), k);
                }
            }
            if (i == SYNC) {
                // publish stage st + 1 (of this tile, or stage 0 of the next one) -- and, in a tile's first stage, the staged
                // output tile of the one before --, then refill the slot of stage st - 1
                if constexpr (INBN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the transform's writes
                else {
                    if constexpr (has_res) { if (st == NCH - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 3) * WR_NI + WR_TH * NSUB) : "memory"); }
                    if (!(has_res && st == NCH - 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 3) * WR_NI) : "memory");
                    if (st == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the epilogue rows' writes
                }
                if (!(WR_EXP & 2)) __builtin_amdgcn_s_barrier();      // (raw: __syncthreads would drain the loads in flight)
                asm volatile("" ::: "memory");
                // (the 12 load instructions of a workgroup issued one per step, wave by wave, instead of together here:
                // 69.6 vs 59.6 us on 256 -> 256 -- the per-wave branches cost more than the queueing in the address path)
                const int far = st + D - 1;
                if (!(WR_EXP & 1)) issue(tile_at(far / NCH), far % NCH, slot_of(far));
            }
            if (st == 0 && i > SYNC && i <= SYNC + NST + 1 && pend) {
                if (i > SYNC + 1) epi_store(i - SYNC - 2);
                if (i <= SYNC + NST) epi_fetch(i - SYNC - 1);
            }
            rd(gg + PFD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = h - kh;
                if (r < 0 || r >= WR_TH) continue;
                const bool first = st == 0 && kw == 0 && kh == 0;       // the row's first product of this tile
#pragma unroll
                for (int s = 0; s < NSUB; ++s)
                    acc[s][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][st][kh * 3 + kw], B[gg % NB].h, first ? zero4 : acc[s][r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        WR_TS(3 + 3 * k);
        en = t0.n; eoy0 = t0.oy0; eox0 = t0.ox0;
        t0 = t1;
        t1 = t2;
        tile_desc(k + 3, t2);
        if constexpr (!STATIC_SLOT) s0 = (s0 + NCH) % D;
    }
    WR_CYC(29);
    WR_TS(30);
    if (K > 0) {                                           // the last tile's epilogue
#pragma unroll
        for (int r = 0; r < WR_TH; ++r) epi_row(r);
        epi_stats(K - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < NST; ++it) { epi_fetch(it); epi_store(it); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the run-ahead loads past the end of the stream (zero block)
    WR_TS(31);
}

template <int NSUB, int NCH, int D, int PFD, bool INBN, int WPS, bool RES, bool RELU>
int wr_launch(const ConvArgs& c, hipStream_t st, int cus) {
    constexpr int NCO = 64 * NSUB;
    WrArgs a;
    a.c = c;
    a.ncb = c.Cout / NCO;
    a.subt = c.rw == 4 ? 2 : 1;
    a.tilesX = c.OW / TW;
    a.unitsY = c.OH / (WR_TH * a.subt);
    a.nunits = c.N * a.unitsY * a.tilesX;
    int J = (cus * WPS / a.ncb) & ~7;                      // WPS workgroups per CU, a multiple of 8 per output-channel block
    if (J > a.nunits) J = (a.nunits + 7) & ~7;
    if (J < 8) J = 8;
    a.J = J;
    a.wfrag = c.w_frag != nullptr;
    if (a.wfrag) a.c.w = c.w_frag;
    const size_t lds = (size_t)D * WR_IMG + (size_t)WR_TH * TW * (NCO * 2 + 16) + (INBN ? (size_t)c.in_groups * 2 * c.Cin * 4 : 0);
    auto kern = conv3x3_wreg_kernel<NSUB, NCH, D, PFD, INBN, WPS, RES, RELU>;
    static bool attr_done = false;
    if (!attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wreg: cannot raise dynamic LDS to 160 KB");
        }
    }
    hipLaunchKernelGGL(kern, dim3(a.ncb * J), dim3(256), lds, st, a);
    DH_CHECK_LAUNCH("conv_wreg");
    return 0;
}

template <int NSUB, int NCH, int D, int PFD, int WPS>
int wr_launch_bn(const ConvArgs& c, hipStream_t st, int cus) {
    // (BatchNorm on load comes with neither residual nor ReLU in any caller: dh_conv_wreg_eligible refuses the combination)
    const bool relu = c.act == DH_ACT_RELU;
    if (c.in_scale) return wr_launch<NSUB, NCH, D, PFD, true, WPS, false, false>(c, st, cus);
    if (c.res) return relu ? wr_launch<NSUB, NCH, D, PFD, false, WPS, true, true>(c, st, cus)
                           : wr_launch<NSUB, NCH, D, PFD, false, WPS, true, false>(c, st, cus);
    return relu ? wr_launch<NSUB, NCH, D, PFD, false, WPS, false, true>(c, st, cus)
                : wr_launch<NSUB, NCH, D, PFD, false, WPS, false, false>(c, st, cus);
}

// ---- 32 -> 32 channels (classifier.0, the 32-channel convolutions of the UNet up path) ----------------------------------------
// One 32-channel chunk, 32 output channels: the whole layer is 18 KB of weights, which the tap kernel stages next to EVERY
// 11.5 KB halo (447 TFLOP/s on 32 x 256 x 256 pixels, 86 us where the tensors' HBM time is 54).  Same stream as above with
// one stage per tile; the four waves split a tile as (16-channel half) x (4-row half), nine weight fragments = 36 registers
// per lane, two workgroups per CU.  The two row halves of a channel meet in the BatchNorm partial sums: the upper half
// parks its row sums in LDS, the lower half adds them after the stage barrier (so the sums are (rows 0-3) + (rows 4-7), not
// the tap kernel's running sum over eight rows: equal up to fp32 rounding, unlike y, which is bit-identical).
// (The bilinear-x4 footprint epilogue of classifier.0's data gradient, ConvArgs::up4_partial, was tried here too -- whole at the
// end of its tile, two extra barriers: 98.6 us in the step against the tap kernel's 86 -- and stays with the tap kernel.)
template <bool RES, bool RELU, int WPS>
__global__ __launch_bounds__(256, WPS) void conv3x3_wreg32_kernel(WrArgs a) {
    constexpr int D = 3, PFD = 2, NB = PFD + 1;
    constexpr int RH = 4, HR = RH + 2;                     // output rows / halo rows per wave
    constexpr int NSTEP = 3 * HR;                          // 18 steps per tile and wave
    constexpr int SYNC = 8;
    constexpr int NCO = 32, TPITCH = NCO * 2 + 16, PPR = NCO * 2 / 16;
    constexpr int NST = WR_TH * TW * PPR / 256;            // 2 cooperative store rounds
    static_assert(NSTEP % NB == 0 && SYNC + NST + 1 < NSTEP && RH < SYNC, "pipeline shape");
    const ConvArgs& p = a.c;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(wr_lds_addr(smem));
    // The staged output tile and the parked sums are DOUBLE-BUFFERED by tile parity: with one stage per tile there is no
    // barrier between the cooperative reads of tile k - 1 (steps 9 .. 11 of tile k) and the row writes of tile k (steps 0 .. 3
    // of tile k + 1); a wave can be at most one barrier ahead of another, so two buffers are enough.
    constexpr int OT = WR_TH * TW * TPITCH;
    unsigned char* otile0 = smem + D * WR_IMG;             // [2][128 px][TPITCH]
    float* spart0 = reinterpret_cast<float*>(otile0 + 2 * OT);      // [2][2 channel halves][4 groups][2][4]: upper rows' sums

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const int ch = wv & 1, rh = wv >> 1;
    const int j0 = blockIdx.x;
    const int co_w = ch * 16;
    const int Cin = p.Cin;                                 // 32

    s16x8 A[9];
    {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            A[t] = *reinterpret_cast<const s16x8*>(
                a.wfrag ? wb + ((size_t)ch * 9 + t) * 1024 + lane * 16
                        : wb + ((size_t)(t * p.CoutPad + co_w + pl) * Cin + g * 8) * 2);
    }
    float bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = p.bias ? p.bias[co_w + g * 4 + j] : 0.f;

    int hyx[WR_NI];
#pragma unroll
    for (int k = 0; k < WR_NI; ++k) {
        const int i = (k * 4 + wv) * 64 + lane, px = i >> 2, qs = i & 3;
        const int hy = px / WR_HW, hx = px - hy * WR_HW;
        hyx[k] = px < WR_NPX ? (((qs ^ (((hx >> 2) & 1) << 1)) << 16) | (hy << 8) | hx) : -1;
    }
    int lo[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) lo[kw] = wr_off(rh * RH, pl + kw, g);

    const int K = a.nunits > j0 ? (a.nunits - j0 + a.J - 1) / a.J : 0;      // tiles of this workgroup's stream
    auto tile_desc = [&](int k, WrTile& t) {
        if (k >= K) {
            t.img = reinterpret_cast<const unsigned char*>(wr_zero);
            t.o0 = t.o1 = t.o2 = ~0u;
            t.n = 0; t.oy0 = 0; t.ox0 = 0;
            return;
        }
        const int u = j0 + k * a.J;
        const int tx = u % a.tilesX, r = u / a.tilesX, uy = r % a.unitsY, n = r / a.unitsY;
        t.n = n; t.oy0 = uy * WR_TH; t.ox0 = tx * TW;
        t.img = reinterpret_cast<const unsigned char*>(p.x) + (size_t)n * p.H * p.W * Cin * 2;
        auto piece = [&](int code) {
            code = wr_opaque(code);
            const int hy = (code >> 8) & 0xff, hx = code & 0xff, q = (code >> 16) & 3;
            const int iy = t.oy0 - 1 + hy, ix = t.ox0 - 1 + hx;
            const bool ok = code >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            return ok ? (unsigned)(iy * p.W + ix) * (unsigned)(Cin * 2) + q * 16 : ~0u;
        };
        t.o0 = piece(hyx[0]); t.o1 = piece(hyx[1]); t.o2 = piece(hyx[2]);
    };
    auto issue = [&](const WrTile& t, int slot) {
#pragma unroll
        for (int i = 0; i < WR_NI; ++i) {
            const unsigned char* src = t.off(i) != ~0u ? t.img + t.off(i) : reinterpret_cast<const unsigned char*>(wr_zero);
            wr_glds16(src, ring_lds + slot * WR_IMG + (i * 4 + wv) * 1024);
        }
    };

    f32x4 acc[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[4], ssq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }

    WrTile t0, t1, t2;
    tile_desc(0, t0);
    tile_desc(1, t1);
    tile_desc(2, t2);
    int s0 = 0;                                            // ring slot of the current tile
    issue(t0, 0);
    issue(t1, 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(A[t]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bs[j]));
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WR_NI) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    V16u B[NB];
    auto rd = [&](int gg) {                                // fragment of step gg (>= NSTEP: the next tile)
        const int st = gg / NSTEP, i = gg - st * NSTEP, kw = i / HR, hh = i - kw * HR;
        int slot = s0 + st;
        slot = slot >= D ? slot - D : slot;
        B[gg % NB].u = *reinterpret_cast<const uint4*>(ring + slot * WR_IMG + lo[kw] + hh * (WR_HW * 64));
    };
#pragma unroll
    for (int gg = 0; gg < PFD; ++gg) rd(gg);

    int en = 0, eoy0 = 0, eox0 = 0;                        // the tile whose epilogue is pending
    unsigned long long rr[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) rr[r] = 0;
    auto res_issue = [&](const WrTile& t) {
        const bf16* rin = reinterpret_cast<const bf16*>(p.res) + (size_t)t.n * p.OH * p.OW * p.Cout;
#pragma unroll
        for (int r = 0; r < RH; ++r)
            wr_gload8(rin + (size_t)((t.oy0 + rh * RH + r) * p.OW + t.ox0 + pl) * p.Cout + co_w + wr_opaque(g) * 4, rr[r]);
    };
    unsigned char* otile = otile0;                         // buffers of the pending tile
    float* spart = spart0;
    auto epi_row = [&](int r) {                            // r: row of this wave's half (constant after unrolling)
        float v[4], res4[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES) {
            switch (r) {                                   // younger loads: the rows after it and the WR_NI ring loads of the stage
            case 0: wr_wait_for<3 + WR_NI>(rr[0]); break;
            case 1: wr_wait_for<2 + WR_NI>(rr[1]); break;
            case 2: wr_wait_for<1 + WR_NI>(rr[2]); break;
            default: wr_wait_for<WR_NI>(rr[3]); break;
            }
            const unsigned lo32 = (unsigned)rr[r], hi32 = (unsigned)(rr[r] >> 32);
            res4[0] = __uint_as_float(lo32 << 16); res4[1] = __uint_as_float(lo32 & 0xffff0000u);
            res4[2] = __uint_as_float(hi32 << 16); res4[3] = __uint_as_float(hi32 & 0xffff0000u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[r][j] + bs[j];
            if constexpr (RES) v[j] += res4[j];
            if constexpr (RELU) v[j] = fmaxf(v[j], 0.f);
            ssum[j] += v[j];
            ssq[j] += v[j] * v[j];
        }
        st4(reinterpret_cast<bf16*>(otile + ((rh * RH + r) * TW + pl) * TPITCH) + co_w + g * 4, v);
    };
    float tsum[4], tsq[4];                                 // this wave's row sums of the pending tile (statistics)
    auto epi_stats_park = [&]() {
        if (!p.stats) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tsum[j] = row16_sum(ssum[j]); tsq[j] = row16_sum(ssq[j]);
            ssum[j] = 0.f; ssq[j] = 0.f;
        }
        if (rh == 1 && pl == 0) {
            *reinterpret_cast<float4*>(spart + ((ch * 4 + g) * 2 + 0) * 4) = make_float4(tsum[0], tsum[1], tsum[2], tsum[3]);
            *reinterpret_cast<float4*>(spart + ((ch * 4 + g) * 2 + 1) * 4) = make_float4(tsq[0], tsq[1], tsq[2], tsq[3]);
        }
    };
    auto epi_stats_write = [&](int kk) {                   // after the stage barrier: lower half + parked upper half
        if (!p.stats || rh != 0 || pl != 0) return;
        const int unit = j0 + kk * a.J;
        const float4 us = *reinterpret_cast<const float4*>(spart + ((ch * 4 + g) * 2 + 0) * 4);
        const float4 uq = *reinterpret_cast<const float4*>(spart + ((ch * 4 + g) * 2 + 1) * 4);
        const float hs[4] = {us.x, us.y, us.z, us.w}, hq[4] = {uq.x, uq.y, uq.z, uq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = co_w + wr_opaque(g) * 4 + j;
            p.stats[((size_t)0 * p.CoutPad + c) * a.nunits + unit] = tsum[j] + hs[j];
            p.stats[((size_t)1 * p.CoutPad + c) * a.nunits + unit] = tsq[j] + hq[j];
        }
    };
    uint4 ehold;
    auto epi_fetch = [&](int it) {
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        ehold = *reinterpret_cast<const uint4*>(otile + px * TPITCH + q * 16);
    };
    auto epi_store = [&](int it) {
        bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)en * p.OH * p.OW * p.Cout;
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        *reinterpret_cast<uint4*>(yout + (size_t)((eoy0 + (px >> 4)) * p.OW + eox0 + (px & 15)) * p.Cout + q * 8) = ehold;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < K; ++k) {
        const bool pend = k > 0;
        otile = otile0 + ((k - 1) & 1) * OT;               // (the pending tile is k - 1)
        spart = spart0 + ((k - 1) & 1) * 64;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int kw = i / HR, hh = i - kw * HR;
            if (i <= RH && pend) {
                if (i < RH) epi_row(i);
                else epi_stats_park();
            }
            if constexpr (RES) { if (i == RH) res_issue(t0); }         // (after the pending tile's rows have used rr)
            if (i == SYNC) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(RES ? RH : 0) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                int far = s0 + 2;
                far = far >= D ? far - D : far;
                if (!(WR_EXP & 1)) issue(t2, far);
            }
            if (i == SYNC + 1 && pend) epi_stats_write(k - 1);
            if (i > SYNC && i <= SYNC + NST + 1 && pend) {
                if (i > SYNC + 1) epi_store(i - SYNC - 2);
                if (i <= SYNC + NST) epi_fetch(i - SYNC - 1);
            }
            rd(i + PFD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = hh - kh;
                if (r < 0 || r >= RH) continue;
                const bool first = kw == 0 && kh == 0;
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kh * 3 + kw], B[i % NB].h, first ? zero4 : acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        en = t0.n; eoy0 = t0.oy0; eox0 = t0.ox0;
        t0 = t1;
        t1 = t2;
        tile_desc(k + 3, t2);
        s0 = s0 + 1 >= D ? 0 : s0 + 1;
    }
    if (K > 0) {                                           // the last tile's epilogue
        otile = otile0 + ((K - 1) & 1) * OT;
        spart = spart0 + ((K - 1) & 1) * 64;
#pragma unroll
        for (int r = 0; r < RH; ++r) epi_row(r);
        epi_stats_park();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        epi_stats_write(K - 1);
#pragma unroll
        for (int it = 0; it < NST; ++it) { epi_fetch(it); epi_store(it); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool RES, bool RELU>
int wr32_launch(const ConvArgs& c, hipStream_t st, int cus) {
    constexpr int WPS = 2;                                 // (57 KB of LDS per workgroup)
    WrArgs a;
    a.c = c;
    a.ncb = 1;
    a.subt = 1;
    a.tilesX = c.OW / TW;
    a.unitsY = c.OH / WR_TH;
    a.nunits = c.N * a.unitsY * a.tilesX;
    int J = (cus * WPS) & ~7;
    if (J > a.nunits) J = (a.nunits + 7) & ~7;
    if (J < 8) J = 8;
    a.J = J;
    a.wfrag = c.w_frag != nullptr;
    if (a.wfrag) a.c.w = c.w_frag;
    const size_t lds = (size_t)3 * WR_IMG + (size_t)2 * WR_TH * TW * (32 * 2 + 16) + 2 * 64 * 4;
    auto kern = conv3x3_wreg32_kernel<RES, RELU, WPS>;
    static bool attr_done = false;
    if (!attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wreg32: cannot raise dynamic LDS");
        }
    }
    hipLaunchKernelGGL(kern, dim3(J), dim3(256), lds, st, a);
    DH_CHECK_LAUNCH("conv_wreg32");
    return 0;
}


// ---- classifier.0 on the bilinear-x4 upsampled |A - B| map WITHOUT that map (models/networks.py:200,383-389) -------------------
// The 32 -> 32 stream above, fed from the two COARSE maps a, b [N][H / 4][W / 4][32]: the upsampled tensor (134 MB at 32 pairs of
// 256 x 256) is neither written by dh_absdiff_upsample4_fwd nor read here.  Per 8 x 16 tile the 10 x 18 halo depends on 4 x 6
// coarse pixels (rows oy0 / 4 - 1 .. + 2, columns ox0 / 4 - 1 .. + 4, clamped to the map: a clamped entry is only ever read with
// weight 0).  Three small stages run AHEAD of the matrix stream, each published by the stream's one barrier per tile:
//   tile T - 4 (after its barrier)   the 2 x 24 x 64 B of a and b are requested into registers of lanes 0 .. 95 (plain loads)
//   tile T - 3 (after its barrier)   d = |a - b| in fp32, once per coarse value (those 96 lanes: one 8-channel piece each), to LDS
//   tile T - 1 (steps before the barrier)  every lane interpolates ITS three pieces of the halo image -- the pieces its loads bring
//                                     in the kernel above -- from d: the four bilinear terms in the order of absdiff_up4_fwd_kernel
//                                     (bit-identical to convolving that kernel's bf16 output), one piece per step between the MFMAs
// so the interpolation happens ONCE per halo element, in LDS, beside the matrix work of the tile before.  (The first attempt
// put it on the load path of the tap kernel, which stages 18 KB of weights per tile and cannot overlap anything inside its one
// tile per workgroup: 121 us against 32 + 76 for the two-kernel path.)
// THIS FILE IS COMPILED WITH -fno-slp-vectorize (Makefile): with the default SLP packing of the interpolation's fp32 chains
// into v_pk_fma_f32 / v_pk_mul_f32 the kernel returned, from run to run, wrong EVEN channels for groups of 16 consecutive lanes
// (4 adjacent halo pixels; 0.1 - 0.4 % of the pixels), with every cross-wave stage fenced by full barriers and with or without
// the direct-to-LDS loads -- scalar fp32 instructions give the materialised path's bits on every run (tools/up4_bench.py,
// test_classifier0_upsample_fused_into_the_weights_resident_stream).  The other kernels of the file are bit-identical and time-
// neutral under the flag (s4 step 9667 / 9632 against 9656 / 9628 pairs/s, same box, interleaved).
template <int WPS>
__global__ __launch_bounds__(256, WPS) void conv3x3_up4_wreg32_kernel(WrArgs a) {
    constexpr int D = 3, PFD = 2, NB = PFD + 1;
    constexpr int RH = 4, HR = RH + 2;
    constexpr int NSTEP = 3 * HR;
    constexpr int SYNC = 8;
    constexpr int NCO = 32, TPITCH = NCO * 2 + 16, PPR = NCO * 2 / 16;
    constexpr int NST = WR_TH * TW * PPR / 256;
    constexpr int CPX = 4 * 6;                             // coarse pixels per tile footprint
    constexpr int DSLOT = CPX * 32 * 4;                    // bytes of one |a - b| slot: [24 px][32 fp32]
    static_assert(NSTEP % NB == 0 && SYNC + NST + 1 < NSTEP && RH < SYNC, "pipeline shape");
    const ConvArgs& p = a.c;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* dif0 = smem;                                                       // [2 slots][DSLOT]
    unsigned char* ring = dif0 + 2 * DSLOT;
    constexpr int OT = WR_TH * TW * TPITCH;
    unsigned char* otile0 = ring + D * WR_IMG;
    float* spart0 = reinterpret_cast<float*>(otile0 + 2 * OT);

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = lane & 15, g = lane >> 4;
    const int ch = wv & 1, rh = wv >> 1;
    const int j0 = blockIdx.x;
    const int co_w = ch * 16;
    const int CHh = p.H >> 2, CWw = p.W >> 2;

    s16x8 A[9];
    {
        const unsigned char* wb = reinterpret_cast<const unsigned char*>(p.w);
#pragma unroll
        for (int t = 0; t < 9; ++t)
            A[t] = *reinterpret_cast<const s16x8*>(
                a.wfrag ? wb + ((size_t)ch * 9 + t) * 1024 + lane * 16
                        : wb + ((size_t)(t * p.CoutPad + co_w + pl) * 32 + g * 8) * 2);
    }
    float bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = p.bias ? p.bias[co_w + g * 4 + j] : 0.f;

    // this lane's three pieces of a halo image (as the loads of the kernel above bring them): (source piece, hy, hx) or -1
    int hyx[WR_NI];
#pragma unroll
    for (int k = 0; k < WR_NI; ++k) {
        const int i = (k * 4 + wv) * 64 + lane, px = i >> 2, qs = i & 3;
        const int hy = px / WR_HW, hx = px - hy * WR_HW;
        hyx[k] = px < WR_NPX ? (((qs ^ (((hx >> 2) & 1) << 1)) << 16) | (hy << 8) | hx) : -1;
    }
    int lo[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) lo[kw] = wr_off(rh * RH, pl + kw, g);

    const int K = a.nunits > j0 ? (a.nunits - j0 + a.J - 1) / a.J : 0;
    // tile T of this workgroup's stream -> (image, oy0, ox0); T >= K: a dummy inside the first image (its stages are computed
    // and never read)
    auto coords = [&](int T, int& n, int& oy0, int& ox0) {
        const int u = T < K ? j0 + T * a.J : 0;
        const int tx = u % a.tilesX, r = u / a.tilesX, uy = r % a.unitsY;
        n = r / a.unitsY; oy0 = uy * WR_TH; ox0 = tx * TW;
    };
    // stage 1: the coarse footprint of tile T into REGISTERS of lanes 0 .. 95 (piece tid of [24 px][4 pieces], a and b): plain
    // global loads, consumed one tile later -- the only loads of the stream, so the wait the compiler puts in front of their
    // use finds nothing else outstanding (the stores of a tile's epilogue are issued after it)
    uint4 fa = make_uint4(0, 0, 0, 0), fb = fa;
    auto load_coarse = [&](int T) {
        if (tid >= CPX * 4) return;
        int n, oy0, ox0;
        coords(T, n, oy0, ox0);
        const int cp = tid >> 2, q = tid & 3, r = cp / 6, c = cp - r * 6;
        int cy = (oy0 >> 2) - 1 + r, cx = (ox0 >> 2) - 1 + c;
        cy = cy < 0 ? 0 : (cy > CHh - 1 ? CHh - 1 : cy);
        cx = cx < 0 ? 0 : (cx > CWw - 1 ? CWw - 1 : cx);
        const size_t off = (((size_t)n * CHh + cy) * CWw + cx) * 32 + q * 8;
        fa = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.up4_a) + off);
        fb = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.up4_b) + off);
    };
    // stage 2: d = |a - b| (fp32) of the footprint in the registers -> difference slot T & 1
    auto absdiff = [&](int T) {
        if (tid >= CPX * 4) return;
        float u[8], v[8];
        unpack16(fa, u);
        unpack16(fb, v);
        float* d = reinterpret_cast<float*>(dif0 + (T & 1) * DSLOT) + tid * 8;
        *reinterpret_cast<float4*>(d) = make_float4(fabsf(u[0] - v[0]), fabsf(u[1] - v[1]), fabsf(u[2] - v[2]), fabsf(u[3] - v[3]));
        *reinterpret_cast<float4*>(d + 4) = make_float4(fabsf(u[4] - v[4]), fabsf(u[5] - v[5]), fabsf(u[6] - v[6]), fabsf(u[7] - v[7]));
    };
    // stage 3: piece k of this lane in the halo image of tile T (ring slot `slot`), from difference slot T & 1
    auto interp = [&](int T, int slot, int k) {
        const int code = wr_opaque(hyx[k]);
        if (code < 0) return;
        int n, oy0, ox0;
        coords(T, n, oy0, ox0);
        const int hy = (code >> 8) & 0xff, hx = code & 0xff, q = (code >> 16) & 3;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        uint4 out = make_uint4(0, 0, 0, 0);                // padding of the upsampled tensor
        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
            int y0, y1, x0, x1;
            float ly, lx;
            up4_src(iy, CHh, y0, y1, ly);
            up4_src(ix, CWw, x0, x1, lx);
            const float wy[2] = {1.f - ly, ly}, wx[2] = {1.f - lx, lx};
            const int cyb = (oy0 >> 2) - 1, cxb = (ox0 >> 2) - 1;
            const int rr[2] = {y0 - cyb, y1 - cyb}, cc[2] = {x0 - cxb, x1 - cxb};
            const float* dbase = reinterpret_cast<const float*>(dif0 + (T & 1) * DSLOT);
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const float* d = dbase + (rr[pp] * 6 + cc[qq]) * 32 + q * 8;
                    float dv[8];
                    *reinterpret_cast<float4*>(dv) = *reinterpret_cast<const float4*>(d);
                    *reinterpret_cast<float4*>(dv + 4) = *reinterpret_cast<const float4*>(d + 4);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += wy[pp] * wx[qq] * dv[j];
                }
            out = pack16<bf16>(acc);
        }
        *reinterpret_cast<uint4*>(ring + slot * WR_IMG + ((k * 4 + wv) * 64 + lane) * 16) = out;
    };

    f32x4 acc[RH];
#pragma unroll
    for (int r = 0; r < RH; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssum[4], ssq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }

    // ---- prologue: differences 0 .. 2 formed, halo image 0 published, footprint 3 in the registers ----
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" : "+v"(A[t]));
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bs[j]));
    load_coarse(0);
    absdiff(0);
    load_coarse(1);
    absdiff(1);
    load_coarse(2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // differences 0, 1 published
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < WR_NI; ++k) interp(0, 0, k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // halo image 0 published; difference slot 0 free
    asm volatile("" ::: "memory");
    absdiff(2);                                            // (published by the barrier of tile 0, read in tile 1)
    load_coarse(3);
    int s0 = 0;                                            // ring slot of the current tile

    V16u B[NB];
    auto rd = [&](int gg) {
        const int st = gg / NSTEP, i = gg - st * NSTEP, kw = i / HR, hh = i - kw * HR;
        int slot = s0 + st;
        slot = slot >= D ? slot - D : slot;
        B[gg % NB].u = *reinterpret_cast<const uint4*>(ring + slot * WR_IMG + lo[kw] + hh * (WR_HW * 64));
    };
#pragma unroll
    for (int gg = 0; gg < PFD; ++gg) rd(gg);

    int en = 0, eoy0 = 0, eox0 = 0;                        // the tile whose epilogue is pending
    unsigned char* otile = otile0;
    float* spart = spart0;
    auto epi_row = [&](int r) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = acc[r][j] + bs[j];
            ssum[j] += v[j];
            ssq[j] += v[j] * v[j];
        }
        st4(reinterpret_cast<bf16*>(otile + ((rh * RH + r) * TW + pl) * TPITCH) + co_w + g * 4, v);
    };
    float tsum[4], tsq[4];
    auto epi_stats_park = [&]() {
        if (!p.stats) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tsum[j] = row16_sum(ssum[j]); tsq[j] = row16_sum(ssq[j]);
            ssum[j] = 0.f; ssq[j] = 0.f;
        }
        if (rh == 1 && pl == 0) {
            *reinterpret_cast<float4*>(spart + ((ch * 4 + g) * 2 + 0) * 4) = make_float4(tsum[0], tsum[1], tsum[2], tsum[3]);
            *reinterpret_cast<float4*>(spart + ((ch * 4 + g) * 2 + 1) * 4) = make_float4(tsq[0], tsq[1], tsq[2], tsq[3]);
        }
    };
    auto epi_stats_write = [&](int kk) {
        if (!p.stats || rh != 0 || pl != 0) return;
        const int unit = j0 + kk * a.J;
        const float4 us = *reinterpret_cast<const float4*>(spart + ((ch * 4 + g) * 2 + 0) * 4);
        const float4 uq = *reinterpret_cast<const float4*>(spart + ((ch * 4 + g) * 2 + 1) * 4);
        const float hs[4] = {us.x, us.y, us.z, us.w}, hq[4] = {uq.x, uq.y, uq.z, uq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = co_w + wr_opaque(g) * 4 + j;
            p.stats[((size_t)0 * p.CoutPad + c) * a.nunits + unit] = tsum[j] + hs[j];
            p.stats[((size_t)1 * p.CoutPad + c) * a.nunits + unit] = tsq[j] + hq[j];
        }
    };
    uint4 ehold;
    auto epi_fetch = [&](int it) {
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        ehold = *reinterpret_cast<const uint4*>(otile + px * TPITCH + q * 16);
    };
    auto epi_store = [&](int it) {
        bf16* yout = reinterpret_cast<bf16*>(p.y) + (size_t)en * p.OH * p.OW * p.Cout;
        const int i = wr_opaque(tid) + it * 256, px = i / PPR, q = i - px * PPR;
        *reinterpret_cast<uint4*>(yout + (size_t)((eoy0 + (px >> 4)) * p.OW + eox0 + (px & 15)) * p.Cout + q * 8) = ehold;
    };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int k = 0; k < K; ++k) {
        const bool pend = k > 0;
        otile = otile0 + ((k - 1) & 1) * OT;
        spart = spart0 + ((k - 1) & 1) * 64;
        int nslot = s0 + 1;
        nslot = nslot >= D ? nslot - D : nslot;
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            const int kw = i / HR, hh = i - kw * HR;
            if (i <= RH && pend) {
                if (i < RH) epi_row(i);
                else epi_stats_park();
            }
            // the halo image of tile k + 1 from the differences of tile k + 1: one piece in each of three steps before the barrier
            if (i == 1 || i == 3 || i == 5) interp(k + 1, nslot, (i - 1) >> 1);
            if (i == SYNC) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            // footprint k + 3 (requested a tile ago) -> the slot of difference k + 1, read before this barrier; then the request
            // for footprint k + 4.  Before the epilogue's stores: what the wait in front of the registers' use finds outstanding
            if (i == SYNC + 1) { absdiff(k + 3); load_coarse(k + 4); }
            if (i == SYNC + 1 && pend) epi_stats_write(k - 1);
            if (i > SYNC && i <= SYNC + NST + 1 && pend) {
                if (i > SYNC + 1) epi_store(i - SYNC - 2);
                if (i <= SYNC + NST) epi_fetch(i - SYNC - 1);
            }
            rd(i + PFD);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int r = hh - kh;
                if (r < 0 || r >= RH) continue;
                const bool first = kw == 0 && kh == 0;
                acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kh * 3 + kw], B[i % NB].h, first ? zero4 : acc[r], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        coords(k, en, eoy0, eox0);
        s0 = nslot;
    }
    if (K > 0) {                                           // the last tile's epilogue
        otile = otile0 + ((K - 1) & 1) * OT;
        spart = spart0 + ((K - 1) & 1) * 64;
#pragma unroll
        for (int r = 0; r < RH; ++r) epi_row(r);
        epi_stats_park();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        epi_stats_write(K - 1);
#pragma unroll
        for (int it = 0; it < NST; ++it) { epi_fetch(it); epi_store(it); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int wr32_up4_launch(const ConvArgs& c, hipStream_t st, int cus) {
    constexpr int WPS = 2;
    WrArgs a;
    a.c = c;
    a.ncb = 1;
    a.subt = 1;
    a.tilesX = c.OW / TW;
    a.unitsY = c.OH / WR_TH;
    a.nunits = c.N * a.unitsY * a.tilesX;
    int J = (cus * WPS) & ~7;
    if (J > a.nunits) J = (a.nunits + 7) & ~7;
    if (J < 8) J = 8;
    a.J = J;
    a.wfrag = c.w_frag != nullptr;
    if (a.wfrag) a.c.w = c.w_frag;
    const size_t lds = (size_t)3 * WR_IMG + (size_t)2 * WR_TH * TW * (32 * 2 + 16) + 2 * 64 * 4 + 2 * (24 * 32 * 4);
    auto kern = conv3x3_up4_wreg32_kernel<WPS>;
    static bool attr_done = false;
    if (!attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wreg32 (up4): cannot raise dynamic LDS");
        }
    }
    hipLaunchKernelGGL(kern, dim3(J), dim3(256), lds, st, a);
    DH_CHECK_LAUNCH("conv_wreg32_up4");
    return 0;
}

int g_wreg_mode = -1;      // dh_conv_wreg_mode: -1 = where it is the faster kernel, 0 = never, 1 = wherever it can run


}  // namespace

// the launches this kernel serves: what conv_mfma_kernel<bf16, 3, 1, 64, *, 1, *, FAST = true, *> serves at 64 / 128 / 256
// input channels, whole 8x16 tiles, and enough tiles per persistent workgroup to amortise loading the weights
bool dh_conv_wreg_eligible(const ConvArgs& a, int ks, int stride, int dtype) {
    static const bool off = getenv("DAHITRA_NO_WREG") != nullptr;
    if (off || g_wreg_mode == 0 || dtype != DH_DTYPE_BF16 || ks != 3 || stride != 1 || a.dil != 1 || a.pad != 1) return false;
    if (a.Cin == 32) {
        // the 32 -> 32 kernel: no BatchNorm on load, whole 8x16 tiles, 8-row statistics units, enough tiles for its 768 streams
        if (a.x_split || a.y_split) return false;
        if (a.Cout != 32 || a.CoutPad != 32 || a.in_scale || a.rw != 2 || a.phase_mode || a.gate_y || a.y2 || a.y_nchw || a.w_nstride ||
            a.up4_partial || a.act == DH_ACT_GELU)
            return false;
        if (a.npix != a.OH * a.OW || a.in_npix != a.H * a.W || a.OH != a.H || a.OW != a.W || a.OH % 8 || a.OW % 16) return false;
        return (long)a.N * (a.OH / 8) * (a.OW / 16) >= (g_wreg_mode == 1 ? 16 : 4 * 512);
    }
    if (a.Cin != 64 && a.Cin != 128 && a.Cin != 256) return false;
    if (a.Cout % 64 || a.CoutPad != a.Cout || a.phase_mode || a.gate_y || a.y2 || a.y_nchw || a.w_nstride) return false;
    if (a.act == DH_ACT_GELU || a.npix != a.OH * a.OW || a.in_npix != a.H * a.W || a.OH != a.H || a.OW != a.W) return false;
    if (a.OH % (a.rw == 4 ? 16 : 8) || a.OW % 16) return false;
    if (a.in_scale && (a.in_groups > 4 || a.res || a.act == DH_ACT_RELU)) return false;
    if ((a.x_split && (a.Cin % 128 || a.in_scale)) || (a.y_split && (a.Cout % 128 || a.res))) return false;   // halves = whole chunks / blocks
    if (g_wreg_mode != 1) {
        // measured (tools/wreg_bench.py, 64 images, gpurun_out/wreg_bench_{9,10}*.txt): the shapes on which this kernel is the
        // faster one -- the 64-channel layers (x1.16 with BatchNorm on load, x1.26 - 1.30 without) and, given the
        // fragment-order weights, 128 / 256 input channels without BatchNorm on load (x1.04 - 1.13 / x1.07 - 1.11; with it
        // x0.86 - 0.90: one wave per SIMD cannot hide the in-LDS transform).  DESIGN.md section 6c.
        if (!(a.Cin == 64 || (a.w_frag && !a.in_scale))) return false;
    }
    // a persistent workgroup must see enough tiles to amortise loading its weights (74 KB at 64 input channels, two workgroups
    // per CU; 147 / 295 KB at 128 / 256, one per CU): measured down to 4 tiles per workgroup (256 -> 128 at 64 images: x1.08)
    const long work = (long)a.N * (a.OH / 8) * (a.OW / 16) * (a.Cout / 64);
    if (g_wreg_mode == 1) return work >= 2 * 256;
    return a.Cin == 64 ? work >= 4 * 512 : work >= 4 * 256;
}

// C ABI (include/dahitra_hip.h): route the eligible 3x3 convolutions through the tap-oriented kernel instead (mode 0), or
// back through this one (mode -1); returns the previous mode.  For A/B measurements and the bit-identity test.
extern "C" int dh_conv_wreg_mode(int mode) {
    const int prev = g_wreg_mode;
    g_wreg_mode = mode == 0 ? 0 : (mode == 1 ? 1 : -1);
    return prev;
}

#ifdef WR_TIMING
extern "C" int dh_debug_wreg_ts(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(wr_ts), (size_t)n * 8); }
extern "C" int dh_debug_wreg_clear() { static long long z[4096 * 32]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(wr_ts), z, sizeof(z)); }
#endif

// classifier.0 on upsample4(|a - b|) from the coarse maps (ConvArgs::up4_a / up4_b): the persistent 32 -> 32 stream with the
// interpolation in LDS.  Eligible: what that stream serves (whole 8x16 tiles, enough of them), no residual, no activation.
// NOT the default route (DAHITRA_UP4_WREG=1, or dh_conv_wreg_mode(1)): measured at 32 x 256 x 256 (tools/up4_bench.py,
// profiles/r06a_up4_fused.txt) it takes 116 us against 29.5 + 83.5 for dh_absdiff_upsample4_fwd + the plain stream -- the
// interpolation more than doubles the stream's vector instructions (rocprofv3 SQ_INSTS_VALU 4.15e7 against 1.82e7, SQ_INSTS_LDS
// 3.6e6 against 1.7e6) and two waves per SIMD are then bound by VALU issue (~630 instructions per wave and tile next to 36
// MFMAs), where the plain stream is bound by the 268 MB it moves.
bool dh_conv_wreg_up4_eligible(const ConvArgs& a) {
    static const bool off = getenv("DAHITRA_NO_WREG") != nullptr;
    static const bool on = getenv("DAHITRA_UP4_WREG") != nullptr && atoi(getenv("DAHITRA_UP4_WREG")) == 1;
    if (off || g_wreg_mode == 0 || !(on || g_wreg_mode == 1) || !a.up4_a || !a.up4_b) return false;
    if (a.Cin != 32 || a.Cout != 32 || a.CoutPad != 32 || a.res || a.act != DH_ACT_NONE || a.in_scale) return false;
    if (a.OH != a.H || a.OW != a.W || a.OH % 8 || a.OW % 16) return false;
    return (long)a.N * (a.OH / 8) * (a.OW / 16) >= (g_wreg_mode == 1 ? 16 : 4 * 512);
}
int dh_conv_wreg_up4_launch(const ConvArgs& a, hipStream_t st) {
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    return wr32_up4_launch(a, st, cus);
}

int dh_conv_wreg_launch(const ConvArgs& a, hipStream_t st) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    }
    if (a.Cin == 32) {
        const bool relu = a.act == DH_ACT_RELU;
        if (a.res) return relu ? wr32_launch<true, true>(a, st, cus) : wr32_launch<true, false>(a, st, cus);
        return relu ? wr32_launch<false, true>(a, st, cus) : wr32_launch<false, false>(a, st, cus);
    }
    //                                 NSUB NCH D PFD WPS
    if (a.Cin == 64) return wr_launch_bn<1, 2, 4, 3, 2>(a, st, cus);      // 72 weight registers: two workgroups per CU
    if (a.Cin == 256) return wr_launch_bn<1, 8, 4, 4, 1>(a, st, cus);     // 288: one wave per SIMD
    // 128 input channels.  Also measured and dropped: 32 output channels per wave (288 registers + 64 accumulators: 60 spills, x0.7), two
    // workgroups per CU at 144 registers (31 spills, x0.77), deeper rings (D = 5 / 8: no change, cache-cold inputs included).
    return wr_launch_bn<1, 4, 4, 4, 1>(a, st, cus);
}
