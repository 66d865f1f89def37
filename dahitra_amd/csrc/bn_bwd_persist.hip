// BatchNorm backward (train mode) as ONE persistent launch: reduce, combine and apply with the tensors held ON CHIP.
//
// The two-pass form (norm.hip: bn_bwd_reduce -> bn_bwd_finalize -> bn_bwd_apply) reads dout and x twice and costs three
// launches per layer: 16 BatchNorm layers of the bench step = ~0.99 ms of its 4.6 ms.  The trunk's tensors fit the
// chip: a [64, 64, 64, 64] bf16 activation is 33.5 MB, dout + x = 67 MB = 256 CUs x 256 KB, i.e. half of the register
// file (dout, as g = dout * ReLU-mask, 64 VGPRs per lane) plus 128 KB of each CU's 160 KB LDS (x).  So:
//
//   phase 1   every workgroup (one per CU, 512 threads) loads its slice of dout / x (/ out) ONCE, keeps g in
//             registers and x in LDS, accumulates (sum g, sum g * x) per channel, writes dres = g if asked, and adds
//             its [2][C] partial into FIXED-POINT accumulators (two 64-bit words per sum, integer atomics: order-independent,
//             deterministic, and exact for fp32 partials down to 2^-74);
//   barrier   one device-wide arrival counter (agent-scope release / acquire, MI355X_MICROARCH.md "barrier-counter";
//             256 workgroups = the chip's 256 CUs, one resident workgroup each: registers and LDS admit exactly one);
//   phase 2   every workgroup reads its BatchNorm group's sums, the first one also writes dgamma / dbeta, and
//             dx = gamma * invstd * (g - (s1 + xhat * s2) / M) goes out from the on-chip copies; the last workgroup to
//             leave zeroes the accumulators and counters for the next launch.
//
// HBM traffic per layer: 2 reads + 1 write (+1 write with dres) instead of 4 reads + 1 write (+1), one launch instead of
// three.  Falls back to the two-pass kernels (same results up to summation order) when the tensors do not fit, the
// dtype is fp32, or the device has fewer than 256 CUs.
// Failure is LOUD, twice over.  The barrier needs all 256 workgroups co-resident (one per CU: 148 KB of LDS each); a kernel of
// another stream holding CUs (a collective, a side-stream launch) can keep workgroups out.  The spin is bounded; a workgroup
// that gives up sets bit 0 of the error word, and a non-finite partial sum (which the fixed-point accumulators could not
// carry) sets bit 1.  Whenever the word is non-zero dgamma / dbeta are written as NaN -- the step's loss and parameters turn
// NaN exactly as an overflow in the two-pass kernels would propagate -- and dh_bn_bwd_persist_status() returns (and clears)
// the word for the host checks (dahitra_amd.graph / ops.bn_persist_check raise HipLibraryError).  dahitra_amd never
// launches this kernel where another stream may run beside it (ops.bn_bwd: persist_ok).
#include "common.h"

namespace {

constexpr int PG = 256;            // workgroups = CUs of an MI355X
constexpr int PT = 512;            // threads per workgroup (8 waves, 2 per SIMD: 256 VGPRs per lane, 64 of them hold g)
constexpr int PNP = 16;            // 16-byte pieces of each tensor per thread (16 x 512 x 16 B = 128 KB of x in LDS)
constexpr unsigned SPIN_LIMIT = 1u << 22;
// A partial sum t (fp32) is split as t = H * 2^-20 + L * 2^-74 with integers H = rint(t * 2^20) and L = (t - H * 2^-20) * 2^74
// (|L| <= 2^53: exact for every fp32 t with |t| >= 2^-50, i.e. nothing of a small late-training gradient is rounded away);
// the two words are accumulated separately.  256 workgroups x |t| < 8e9 keep sum H below 2^61, sum L below 2^62.
constexpr double FIX_HI = 1048576.0;                          // 2^20
constexpr double FIX_LO = 18889465931478580854784.0;          // 2^74

struct PersistArgs {
    const bf16* dout; const bf16* out; const bf16* x;
    const float* mean; const float* invstd; const float* gamma; const float* mscale; const float* mshift;
    bf16* dx; bf16* dres;
    float* dgamma; float* dbeta;
    unsigned* sync;        // [0] arrivals, [1] departures, [2] error word (timeout); from word 16 on: the fixed-point
                           // accumulators long long [2 words: H, L][groups][2][C] (all zero between launches)
    long pieces_per_group; // 16-byte pieces of one BatchNorm group
    int C, groups, np, accumulate;
    float inv_m;
    unsigned spin_limit;
    int expect;            // arrivals the barrier waits for (PG; tests pass PG + 1 to force the timeout)
};

template <int MASK>      // 0: no ReLU, 1: mask from `out` (post-activation tensor), 2: mask recomputed from x, 3: mask BYTES (`out` =
                         // relu_bits of dh_bn_apply_bits: one byte per 16-byte piece)
__global__ __launch_bounds__(PT) void bn_bwd_persist_kernel(PersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* xl = reinterpret_cast<uint4*>(smem);                                   // [PNP][PT] pieces of x
    float* red = reinterpret_cast<float*>(smem + (size_t)PNP * PT * 16);          // scratch: [8 waves][C] floats, later the fp64 combine
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // phase stamps of workgroup 0 (wall_clock64: 100 MHz) for tools/bn_bench.py, words 4..15 of the sync block
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(p.sync + 4);
    const bool stamper = blockIdx.x == 0 && tid == 0;
    if (stamper) stamp[0] = wall_clock64();
    const int cvn = p.C >> 3;                        // 16-byte pieces per pixel; divides 64
    const int cv = tid % cvn;
    const int wpg = PG / p.groups;                   // workgroups per BatchNorm group
    const int g = blockIdx.x / wpg, w = blockIdx.x % wpg;
    const long R = (long)p.np * PT;                  // pieces per workgroup
    long long* acc = reinterpret_cast<long long*>(p.sync + 16);
    long long* accl = acc + (size_t)p.groups * 2 * p.C;            // the low words
    const long base = (long)g * p.pieces_per_group;  // first piece of this group
    const long first = (long)w * R;

    // Phase 1 accumulates the RAW sums (sum g, sum g * x) -- no per-channel constants live beside the 64 registers of g
    // -- and the combine turns them into sum g * xhat in fp64:
    // sum g xhat = invstd * (sum g x - mean * sum g).  (A lane adds 8 values per channel; the cross-lane / cross-workgroup
    // sums run in the fixed order below.)  32-bit byte offsets from uniform per-workgroup bases keep addresses in SGPRs.
    float ms[8], mh[8];
    if (MASK == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { ms[j] = p.mscale[g * p.C + cv * 8 + j]; mh[j] = p.mshift[g * p.C + cv * 8 + j]; }
    }
    const long remain = p.pieces_per_group - first;              // pieces of this group from `first` on (<= 0: idle workgroup)
    const size_t wb = (size_t)(base + (remain > 0 ? first : 0)) * 16;     // an idle workgroup's masked loads stay in bounds
    const unsigned char* dout_w = reinterpret_cast<const unsigned char*>(p.dout) + wb;
    const unsigned char* x_w = reinterpret_cast<const unsigned char*>(p.x) + wb;
    const unsigned char* out_w = reinterpret_cast<const unsigned char*>(MASK == 1 ? p.out : p.x) + wb;
    const unsigned char* bits_w = reinterpret_cast<const unsigned char*>(p.out) + (wb >> 4);        // MASK == 3: byte = piece
    unsigned char* dres_w = reinterpret_cast<unsigned char*>(p.dres) + wb;
    unsigned char* dx_w = reinterpret_cast<unsigned char*>(p.dx) + wb;
    // ---- phase 1: load once, keep g in registers / x in LDS, accumulate ----
    uint4 gk[PNP];
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const uint4 zero = make_uint4(0, 0, 0, 0);
    constexpr int HB = 4;                                        // pieces per batch of loads
#pragma unroll
    for (int h = 0; h < PNP; h += HB) {
        uint4 xr[HB], orr[HB];
        unsigned mb[HB];
#pragma unroll
        for (int kk = 0; kk < HB; ++kk) {
            const int k = h + kk;
            const unsigned idx = (unsigned)(k * PT + tid);
            const bool ok = k < p.np && (long)idx < remain;
            const unsigned off = ok ? idx * 16u : 0u;
            gk[k] = *reinterpret_cast<const uint4*>(dout_w + off);
            xr[kk] = *reinterpret_cast<const uint4*>(x_w + off);
            if (MASK == 1) orr[kk] = *reinterpret_cast<const uint4*>(out_w + off);
            if (MASK == 3) mb[kk] = bits_w[off >> 4];
            if (!ok) { gk[k] = zero; xr[kk] = zero; if (MASK == 1) orr[kk] = zero; }
        }
#pragma unroll
        for (int kk = 0; kk < HB; ++kk) {
            const int k = h + kk;
            float d[8], xv[8];
            unpack16(gk[k], d);
            unpack16(xr[kk], xv);
            if (MASK == 1) {
                float o[8];
                unpack16(orr[kk], o);
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
            }
            if (MASK == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = (xv[j] * ms[j] + mh[j]) > 0.f ? d[j] : 0.f;
            }
            if (MASK == 3) {
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = ((mb[kk] >> j) & 1u) ? d[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s1[j] += d[j];
                s2[j] += d[j] * xv[j];
            }
            if (MASK != 0) gk[k] = pack16<bf16>(d);          // masking keeps a value or zeroes it: exact in bf16
            xl[k * PT + tid] = xr[kk];
            if (p.dres) {
                const unsigned idx = (unsigned)(k * PT + tid);
                if (k < p.np && (long)idx < remain) *reinterpret_cast<uint4*>(dres_w + idx * 16u) = gk[k];
            }
        }
        // pin the batch: without this hipcc sinks all 2 x 8 x 16 accumulations below the last batch and keeps every
        // unpacked operand alive until then (~390 registers of demand, spills); the accumulators are made opaque here
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(s1[j]), "+v"(s2[j]));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (stamper) stamp[1] = wall_clock64();
    // lanes l, l + cvn, l + 2 cvn, ... of a wave hold the same channels: butterfly over those, then the 16 waves via LDS
    // (one round per statistic: [8 waves][C] floats)
    for (int o = cvn; o < 64; o <<= 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s1[j] += __shfl_xor(s1[j], o, 64);
            s2[j] += __shfl_xor(s2[j], o, 64);
        }
    }
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (lane < cvn) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wv * p.C + lane * 8 + j] = which == 0 ? s1[j] : s2[j];
        }
        __syncthreads();
        for (int c = tid; c < p.C; c += PT) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < PT / 64; ++r) t += red[r * p.C + c];
            // cross-workgroup sum: FIXED-POINT integer atomics (integer addition is associative, so the result does not
            // depend on the arrival order: deterministic), two words per sum (see FIX_HI / FIX_LO)
            if (!(fabsf(t) < 8.0e9f))          // NaN / Inf / out of range: the integer accumulators cannot carry it
                __hip_atomic_fetch_or(&p.sync[2], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double td = (double)t;
            const long long hi = __double2ll_rn(td * FIX_HI);
            const long long lo = __double2ll_rn((td - (double)hi * (1.0 / FIX_HI)) * FIX_LO);
            __hip_atomic_fetch_add(&acc[(size_t)(g * 2 + which) * p.C + c], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lo != 0) __hip_atomic_fetch_add(&accl[(size_t)(g * 2 + which) * p.C + c], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // ---- device-wide barrier: publish (release), arrive, bounded spin, acquire ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (stamper) stamp[2] = wall_clock64();
    // The only cross-workgroup data are the fixed-point accumulators, written and read by DEVICE-SCOPE atomics (performed
    // at the memory side, coherent across XCDs): no release / acquire fence is needed around the counter -- a release
    // (buffer_wbl2) would also have to write back the megabytes of dres this workgroup just stored.  Every wave drained its
    // atomics (vmcnt(0)) before the __syncthreads above.
    // per-channel constants the coefficient step needs: loaded BEFORE the wait, they do not depend on the sums
    float k_mean = 0.f, k_isd = 0.f, k_gamma = 0.f;
    if (tid < p.C) { k_mean = p.mean[g * p.C + tid]; k_isd = p.invstd[g * p.C + tid]; k_gamma = p.gamma[tid]; }
    if (tid == 0) {
        __hip_atomic_fetch_add(&p.sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(&p.sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.expect) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > p.spin_limit) {
                __hip_atomic_store(&p.sync[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (stamper) stamp[3] = wall_clock64();
    // ---- phase 2: per-channel coefficients from the group's sums, dgamma / dbeta once, dx from the on-chip copies ----
    //   s1 = sum g, sgx = sum g x, s2 = sum g xhat = invstd * (sgx - mean * s1)
    //   dx = A * (g - (s1 + xhat * s2) / M) = A g + B x + K,  A = gamma * invstd,  xhat = (x - mean) * invstd
    float* coef = red;                                       // [3][C]: A, B, K
    auto total = [&](int i) -> double {      // the sum behind accumulator i
        return (double)__hip_atomic_load(&acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (1.0 / FIX_HI) +
               (double)__hip_atomic_load(&accl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (1.0 / FIX_LO);
    };
    if (tid < p.C) {
        const double t1 = total((g * 2 + 0) * p.C + tid);
        const double tgx = total((g * 2 + 1) * p.C + tid);
        const double mean = (double)k_mean, isd = (double)k_isd;
        const double t2 = isd * (tgx - mean * t1);
        const double A = (double)k_gamma * isd, im = (double)p.inv_m;
        coef[tid] = (float)A;
        coef[p.C + tid] = (float)(-A * isd * t2 * im);
        coef[2 * p.C + tid] = (float)(A * (mean * isd * t2 - t1) * im);
    }
    __syncthreads();
    if (stamper) stamp[4] = wall_clock64();
    float cA[8], cB[8], cK[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        cA[j] = coef[cv * 8 + j];
        cB[j] = coef[p.C + cv * 8 + j];
        cK[j] = coef[2 * p.C + cv * 8 + j];
    }
#pragma unroll
    for (int k = 0; k < PNP; ++k) {
        const unsigned idx = (unsigned)(k * PT + tid);
        if (k < p.np && (long)idx < remain) {
            float d[8], xv[8], r[8];
            unpack16(gk[k], d);
            unpack16(xl[k * PT + tid], xv);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = cA[j] * d[j] + (cB[j] * xv[j] + cK[j]);
            *reinterpret_cast<uint4*>(dx_w + idx * 16u) = pack16<bf16>(r);
        }
    }
    if (stamper) stamp[5] = wall_clock64();
    // ---- off the critical path: dgamma / dbeta (first workgroup), departure count, re-arming by the last to leave ----
    if (blockIdx.x == 0 && tid < p.C) {
        double tb = 0.0, tg = 0.0;
        // a timed-out barrier or a non-finite partial: poison the parameter gradients (in-band, no host round trip)
        if (__hip_atomic_load(&p.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) tb = __builtin_nan("");
        for (int gg = 0; gg < p.groups; ++gg) {
            const double u1 = total((gg * 2 + 0) * p.C + tid);
            const double ux = total((gg * 2 + 1) * p.C + tid);
            tb += u1;
            tg += (double)p.invstd[gg * p.C + tid] * (ux - (double)p.mean[gg * p.C + tid] * u1);
        }
        if (tb != tb) tg = tb;
        if (p.accumulate) { p.dgamma[tid] += (float)tg; p.dbeta[tid] += (float)tb; }
        else { p.dgamma[tid] = (float)tg; p.dbeta[tid] = (float)tb; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's reads of the accumulators have completed
    __syncthreads();
    unsigned* last = reinterpret_cast<unsigned*>(red);     // (the coefficients were consumed above)
    if (tid == 0)
        *last = __hip_atomic_fetch_add(&p.sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)PG - 1;
    __syncthreads();
    if (*last) {       // the last workgroup to leave: all its threads zero the accumulators, then the counters
        for (int i = tid; i < p.groups * 4 * p.C; i += PT)         // high and low words
            __hip_atomic_store(&acc[i], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&p.sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace

// 1 when the persistent form serves this layer (bf16, the tensors fit the chip, C / 8 divides 64, C <= 256, 256 CUs)
extern "C" int dh_bn_bwd_persist_supported(int dtype, long npix, int C, int groups) {
    if (dtype != DH_DTYPE_BF16 || C % 8 || C < 8 || C > 256 || 64 % (C / 8) || groups < 1 || PG % groups || npix % groups)
        return 0;
    if ((long)groups * 4 * C * 8 + 64 > 8192 * 4) return 0;      // the accumulators live in the caller's 8192-word sync block
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            cus = 0;
    }
    if (cus < PG) return 0;
    const long ppg = npix / groups * (C / 8);                 // pieces per group
    const long wpg = PG / groups;
    const long np = (ppg + wpg * PT - 1) / (wpg * PT);
    return np >= 1 && np <= PNP;
}
// 1 when it is also FASTER than the two-pass kernels: the launch + device-wide barrier cost ~20 us whatever the size, so
// only tensors of >= ~24 MB gain (measured on MI355X: 16.8 MB tensors tie, 33.5 MB: 28 - 39 us against 35 - 53 us)
extern "C" int dh_bn_bwd_persist_preferred(int dtype, long npix, int C, int groups) {
    if (!dh_bn_bwd_persist_supported(dtype, npix, C, groups)) return 0;
    const long ppg = npix / groups * (C / 8);
    const long wpg = PG / groups;
    return (ppg + wpg * PT - 1) / (wpg * PT) >= 12;
}

// C ABI: see include/dahitra_hip.h.  Returns the error word of a sync block (bit 0: barrier timeout, bit 1: non-finite partial
// sum) and clears it; < 0: the copy failed.  Synchronises the stream (4-byte device-to-host copy): call it every N steps.
extern "C" int dh_bn_bwd_persist_status(unsigned* sync, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned word = 0;
    if (hipMemcpyAsync(&word, sync + 2, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    if (word && hipMemsetAsync(sync + 2, 0, 4, st) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return (int)word;
}
// test hook: arms / disarms a short spin limit so that tests can force the timeout path (limit 0: the normal 2^22 spins)
static unsigned g_spin_limit = 0;
extern "C" int dh_bn_bwd_persist_test_spin_limit(unsigned limit) { g_spin_limit = limit; return 0; }

// C ABI: see include/dahitra_hip.h (dh_bn_bwd with `sync`)
static int bn_bwd_persist_impl(const void* dout, const void* out_relu, bool bits, const void* x, const float* mean,
                               const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                               void* dres, float* dgamma, float* dbeta, int accumulate, const float* mask_scale,
                               const float* mask_shift, void* workspace, unsigned* sync, void* stream) {
    DH_REQUIRE(dh_bn_bwd_persist_supported(DH_DTYPE_BF16, npix, C, groups), "bn_bwd_persist: unsupported shape npix=%ld C=%d groups=%d",
               npix, C, groups);
    DH_REQUIRE(!(out_relu && mask_scale), "bn_bwd_persist: give the ReLU mask either as out_relu or as mask_scale/shift");
    DH_REQUIRE(sync, "bn_bwd_persist: sync words missing");
    PersistArgs a;
    a.dout = (const bf16*)dout; a.out = (const bf16*)out_relu; a.x = (const bf16*)x;
    a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.mscale = mask_scale; a.mshift = mask_shift;
    a.dx = (bf16*)dx; a.dres = (bf16*)dres; a.dgamma = dgamma; a.dbeta = dbeta;
    (void)workspace; a.sync = sync;
    a.pieces_per_group = npix / groups * (C / 8);
    a.C = C; a.groups = groups; a.accumulate = accumulate;
    const long wpg = PG / groups;
    a.np = (int)((a.pieces_per_group + wpg * PT - 1) / (wpg * PT));
    a.inv_m = 1.0f / (float)(npix / groups);
    a.spin_limit = g_spin_limit ? g_spin_limit : SPIN_LIMIT;
    a.expect = g_spin_limit ? PG + 1 : PG;         // (test hook: one arrival that never comes)
    const size_t lds = (size_t)PNP * PT * 16 + 20 * 1024;      // x pieces + reduction scratch (<= 16 KB + 10 KB used in turn)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto go = [&](auto kern) -> int {
        static bool attr_done = false;
        if (!attr_done) {
            attr_done = true;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                (void)hipGetLastError();
                DH_FAIL("bn_bwd_persist: cannot raise dynamic LDS to %zu", lds);
            }
        }
        hipLaunchKernelGGL(kern, dim3(PG), dim3(PT), lds, st, a);
        DH_CHECK_LAUNCH("bn_bwd_persist");
        return 0;
    };
    if (out_relu && bits) return go(bn_bwd_persist_kernel<3>);
    if (out_relu) return go(bn_bwd_persist_kernel<1>);
    if (mask_scale) return go(bn_bwd_persist_kernel<2>);
    return go(bn_bwd_persist_kernel<0>);
}
extern "C" int dh_bn_bwd_persist(const void* dout, const void* out_relu, const void* x, const float* mean,
                                 const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                                 void* dres, float* dgamma, float* dbeta, int accumulate, const float* mask_scale,
                                 const float* mask_shift, void* workspace, unsigned* sync, void* stream) {
    return bn_bwd_persist_impl(dout, out_relu, false, x, mean, invstd, gamma, npix, C, groups, dx, dres, dgamma, dbeta, accumulate,
                               mask_scale, mask_shift, workspace, sync, stream);
}
// C ABI: dh_bn_bwd_bits with `sync` (the ReLU mask as the bytes of dh_bn_apply_bits)
extern "C" int dh_bn_bwd_persist_bits(const void* dout, const unsigned char* relu_bits, const void* x, const float* mean,
                                      const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                                      void* dres, float* dgamma, float* dbeta, int accumulate, void* workspace, unsigned* sync,
                                      void* stream) {
    DH_REQUIRE(relu_bits, "bn_bwd_persist_bits: relu_bits missing");
    return bn_bwd_persist_impl(dout, relu_bits, true, x, mean, invstd, gamma, npix, C, groups, dx, dres, dgamma, dbeta, accumulate,
                               nullptr, nullptr, workspace, sync, stream);
}
