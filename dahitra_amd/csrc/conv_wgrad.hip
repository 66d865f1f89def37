// Weight gradient of an NHWC convolution / linear layer on the CDNA4 matrix cores.
//
//   dW[tap][co][ci] = sum over (n, oy, ox) of dY[n,oy,ox,co] * X[n, oy*s - pad + kh, ox*s - pad + kw, ci]
//
// is a GEMM whose reduction dimension is the PIXEL index, so both MFMA operands need k (= pixel)
// contiguous per lane while memory is channel-contiguous.  The 8x16-pixel dY tile and the haloed
// X tile are staged in LDS as [pixel][channel]; fragments are then gathered
//   fp32: one ds_read_b32 per operand and v_mfma_f32_16x16x4_f32 (4 pixels per MFMA);
//   bf16: two ds_read_b64_tr_b16 (the gfx950 LDS transpose read) per operand, or 8 ds_read_u16
//         when TR = false, feeding v_mfma_f32_16x16x32_bf16 (32 pixels per MFMA).
// A workgroup owns a CTT(co) x IT(ci) x all-taps slab of dW, walks a strided subset of the pixel tiles
// (split-K) and writes an fp32 partial slab; dh_wgrad_reduce sums the partials deterministically straight
// into the OIHW master-gradient layout.  CTT = 64: wave w = co sub-tile w.  CTT = 32 (layers with <= 32
// output channels -- the 32-wide transformer side, the heads: half of a 64-wide tile would be zero padding):
// waves (0,1) and (2,3) take the two co sub-tiles for the first / second half of every tile's pixels and
// are added through LDS at the end.
// groups == N gives one dW per image (per-image attention products, see tokens.hip).
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;
static inline int co_tile(int Cout) { return Cout <= 16 ? 16 : (Cout <= 32 ? 32 : 64); }

struct WgArgs {
    const void* x;
    const void* dy;
    float* part;       // [groups][splitk][taps][Cout][Cin]
    int N, H, W, Cin, OH, OW, Cout, pad;
    int tilesX, tilesY, splitk, groups, npix, in_npix;
    int ci_tiles;
    int CinPitch;      // elements between consecutive pixels of x (>= Cin; the stem reads a padded image)
    int dil;
    int CoutUse;       // output channels that carry a gradient (<= Cout: dy may be zero-padded to a K-chunk)
    int direct;        // 0: write the partial slab; 1 / 2: split-K is 1 and the slab layout IS the destination layout
                       // (1x1, one group), so assign (1) or accumulate (2) straight into dW and skip the reduce launch
    // BatchNorm-apply + ReLU on load (see ConvArgs::in_scale): x is the pre-normalisation tensor of the previous layer
    const float* in_scale;      // [in_groups][Cin] or null
    const float* in_shift;
    int in_groups;
    // phase mode (KS = 2; see ConvArgs::phase_mode): blockIdx.z = output parity (a, b); dY is gathered from the fine grid at
    // (2 oy + a, 2 ox + b) of [N][2 OH][2 OW][Cout], x is read with pad (1 - a, 1 - b); one slab set per phase
    int phase_mode;
    int no_xcd_remap;           // DAHITRA_NO_XCD_REMAP=1
    // DYT (the stem, KS = 4): dy is the masked gradient d of the BatchNorm OUTPUT; the gradient of the convolution output is
    // formed on load as A * d + B * y + C per channel (BatchNorm backward, coefficients from dh_stem_pool_bn_bwd)
    const void* dyt_y;          // pre-normalisation convolution output, same layout as dy
    const float* dytoef;      // [dyt_groups][3][Cout]
    int dyt_groups;
    // x = cat([A, B], channel) of two [N][H][W][Cin / 2] tensors, B at x + x_split bytes (0: off), never materialised
    // (ConvArgs::x_split; wave-specialised kernel only: a 64-channel ci tile lies in one of the two)
    long x_split;
};

constexpr int lds_pitch(int row_bytes) { return ((row_bytes / 32) & 1) ? row_bytes : row_bytes + 32; }

union F8 {
    s16x8 v;
    s16x4 h[2];
    unsigned short s[8];
};

// CIG = 2 (3x3 layers with >= 64 input channels and a 64-wide co tile): a 512-thread workgroup whose second group of four
// waves takes the NEXT 32 input channels of the same pixels -- a 64co x 64ci slab per workgroup.  The dY tile is staged
// once for both halves, and a layer writes (and dh_wgrad_reduce reads) half as many partial slabs at the same number of
// resident waves: the split-K slab traffic of the bench step was ~500 MB written + ~500 MB read per step.
// (the body: workgroup (bx, kz, bz) of an nbx x nkz x . grid; conv_wgrad_kernel / conv_wgrad_multi_kernel below are the entry points)
template <typename T, int KS, int STRIDE, int IT, bool TR, int DIL, int CTT, int CIG, bool DYT>
__device__ __forceinline__ void wg_body(const WgArgs& p, int bx, int kz, const int nbx, const int nkz, const int bz,
                                        unsigned char* smem) {
    constexpr int CT = CTT;
    constexpr int NTHR = 256 * CIG;
    constexpr int ITT = IT * CIG;                  // input channels per workgroup
    constexpr int CW = CTT / 16;                   // co sub-tiles (waves along co)
    constexpr int KSPLIT = 4 / CW;                 // wave groups along the pixel (K) dimension of a tile
    static_assert(CIG == 1 || KSPLIT == 1, "the ci wave groups exist for the 64-wide co tile only");
    constexpr int KPW = TH * TW / KSPLIT;          // pixels of a tile per wave group
    constexpr int HH = (TH - 1) * STRIDE + (KS - 1) * DIL + 1;
    constexpr int HWD = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    constexpr int TAPS = KS * KS;
    constexpr int NI = IT / 16;
    // bf16x3 (T = f32x3, common.h): fp32 tensors in memory, staged as TWO bf16 planes (hi | lo) of the bf16 layout; the
    // bf16 fragment code below runs the three products ah * bh + al * bh + ah * bl
    constexpr bool X3 = Prec<T>::X3;
    constexpr int LE = X3 ? 2 : (int)sizeof(T);    // bytes per LDS element
    constexpr int NPL = X3 ? 2 : 1;                // LDS planes
    constexpr int LV = X3 ? 2 : 1;                 // 16-byte global loads per staged 16-byte LDS piece
    static_assert(!(X3 && (DYT || !TR)), "bf16x3: transpose reads, no BatchNorm-backward-on-load form");
    constexpr bool WCI = CIG == 2 && LE == 2 && IT == 32 && CTT == 64;       // see the MFMA loop
    // LDS pitches are ODD multiples of 32 B: a half-wave of ds_read_b64_tr_b16 then touches 8 consecutive
    // pixel rows x 32 B = 8 distinct bank windows of the 256-byte bank row (conflict-free)
    constexpr int XP = lds_pitch(ITT * LE);    // halo pitch (bytes)
    constexpr int DP = lds_pitch(CT * LE);     // dY tile pitch (bytes)
    constexpr int HPB = HH * HWD * XP, DPB = TH * TW * DP;       // bytes per halo / dY plane
    // (Two staging buffers -- commit of tile t+1 right after the MFMAs of tile t, one barrier per tile -- were measured and
    // change nothing: layer3 90.3 vs 89.5 us, and the 256-thread forms lose a resident workgroup to the second buffer.)
    unsigned char* halo = smem;                      // [NPL][HH*HWD][XP]
    unsigned char* dyt = smem + NPL * HPB;           // [NPL][128][DP]
    float* bnp = reinterpret_cast<float*>(dyt + NPL * DPB);      // in_scale: [in_groups][2][ITT] scale | shift

    const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) & 3, cig = tid >> 8;
    const int pl = lane & 15, g = lane >> 4;
    const int cw = wv % CW, kq = wv / CW;
    // Workgroups are dispatched x-fastest, round-robin over the 8 XCDs (one L2 each).  Remapped so that all (co, ci) tile
    // pairs of one pixel split kz -- they read the SAME x / dY tiles -- run back to back on ONE XCD (its L2 serves the re-reads).
    if (nbx > 1 && (nkz & 7) == 0 && !p.no_xcd_remap) {
        const unsigned lin = (unsigned)(kz * nbx + bx), xcd = lin & 7, sq = lin >> 3;
        bx = (int)(sq % nbx);
        kz = (int)((sq / nbx) * 8 + xcd);
    }
    const int cot = bx / p.ci_tiles, cit = bx % p.ci_tiles;
    const int co0 = cot * CT, ci0 = cit * ITT;
    const int grp = bz;
    const int ph_a = (KS == 2 && p.phase_mode) ? (bz >> 1) : 0, ph_b = (KS == 2 && p.phase_mode) ? (bz & 1) : 0;
    const int imgs_per_group = p.N / p.groups;
    const int tiles_per_img = p.tilesX * p.tilesY;
    const int ntiles = imgs_per_group * tiles_per_img;

    f32x4 acc[TAPS][NI];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline: the global loads of tile t+1 are issued into registers before the MFMAs of tile t
    // and committed to LDS after them, so HBM/L2 latency hides under the matrix work (T14-style split stage).
    constexpr int XQ = ITT * LE / 16, DQ = CT * LE / 16;
    constexpr int NXV = (HH * HWD * XQ + NTHR - 1) / NTHR, NDV = (TH * TW * DQ + NTHR - 1) / NTHR;
    uint4 rx[NXV * LV], rd[NDV * LV];
    uint4 ry[DYT ? NDV : 1];
    unsigned d_okmask = 0;    // DYT: pieces of the fetched dY tile that exist (the others stay zero)
    // Everything about a 16-byte piece that does not depend on the tile is computed ONCE: its halo / tile position,
    // its element offset from the tile origin and how it is loaded (0: zeros, 1: one 16-byte load, 2: ragged channel
    // tail).  Per tile only the scalar tile origin and the border tests remain (no divisions by runtime tile counts,
    // no 64-bit products per piece).
    int x_hy[NXV], x_hx[NXV], x_mode[NXV];
    int d_py[NDV], d_px[NDV], d_mode[NDV];
    constexpr int EPV = 16 / LE;
    const bool xal_ = ((p.Cin * (int)sizeof(T)) & 15) == 0 && ((p.CinPitch * (int)sizeof(T)) & 15) == 0;
    const bool dal_ = ((p.Cout * (int)sizeof(T)) & 15) == 0;
    {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int idx = tid + i * NTHR;
            const int px = idx / XQ, q = idx % XQ;
            x_hy[i] = px / HWD; x_hx[i] = px % HWD;
            const int c = ci0 + q * EPV;
            // (bf16x3: a piece is two 16-byte halves of 4 fp32 channels, each loaded or zeroed on its own -- no ragged mode)
            x_mode[i] = (idx >= HH * HWD * XQ || c >= p.Cin) ? 0 : ((X3 || (xal_ && c + EPV <= p.Cin)) ? 1 : 2);
        }
#pragma unroll
        for (int i = 0; i < NDV; ++i) {
            const int idx = tid + i * NTHR;
            const int px = idx / DQ, q = idx % DQ;
            d_py[i] = px / TW; d_px[i] = px % TW;
            const int c = co0 + q * EPV;
            d_mode[i] = (idx >= TH * TW * DQ || c >= p.Cout) ? 0 : ((X3 || (dal_ && c + EPV <= p.Cout)) ? 1 : 2);
        }
    }
    // ragged channel tail of one piece: element loads, zeros past `nvalid` elements
    auto load_tail = [&](const T* src, int nvalid) {
        __attribute__((aligned(16))) T tmp[EPV];
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
            if (e < nvalid) tmp[e] = src[e];
            else stf(&tmp[e], 0.f);
        }
        return *reinterpret_cast<uint4*>(tmp);
    };
    // tile = (image, ty, tx) advances by splitk tiles per step: a three-digit counter instead of divisions
    const int dn = p.splitk / tiles_per_img, drem = p.splitk % tiles_per_img;
    const int dty = drem / p.tilesX, dtx = drem % p.tilesX;
    int f_n = kz / tiles_per_img, f_ty = (kz % tiles_per_img) / p.tilesX, f_tx = (kz % tiles_per_img) % p.tilesX;
    // Common case (16-byte aligned channel counts, no ragged channel tail in this workgroup's slab): branch-free loads
    // at 32-bit byte offsets from the image base -- an out-of-range piece reads offset 0 and is zeroed by a select.
    // At two to three waves per SIMD this per-tile code is one dependent chain, so its length is time.
    const int xcb = (ci0 + (tid % XQ) * EPV), dcb = (co0 + (tid % DQ) * EPV);            // channel of this thread's pieces
    const bool fast = X3 || (xal_ && dal_ && (ci0 + ITT <= p.Cin || (p.Cin - ci0) % EPV == 0) &&
                             (co0 + CT <= p.Cout || (p.Cout - co0) % EPV == 0));
    unsigned x_okmask = 0;    // pieces of the fetched halo that lie inside the image (BatchNorm-on-load leaves padding zero)
    int c_bng = 0;            // BatchNorm group of the fetched tile's image
    auto fetch = [&]() {      // loads the tile the counter points at, then advances the counter
        const int n = ((KS == 2 && p.phase_mode) ? 0 : grp * imgs_per_group) + f_n;
        x_okmask = 0;
        d_okmask = 0;
        c_bng = p.in_scale ? n / (p.N / p.in_groups) : (DYT ? n / (p.N / p.dyt_groups) : 0);
        const int oy0 = f_ty * TH, ox0 = f_tx * TW;
        const int iy0 = oy0 * STRIDE - p.pad + ph_a, ix0 = ox0 * STRIDE - p.pad + ph_b;
        if (fast) {
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(
                reinterpret_cast<const T*>(p.x) + (size_t)n * p.H * p.W * p.CinPitch);
            const unsigned char* db = reinterpret_cast<const unsigned char*>(
                reinterpret_cast<const T*>(p.dy) + (size_t)n * p.OH * p.OW * p.Cout * ((KS == 2 && p.phase_mode) ? 4 : 1));
            const unsigned xps = (unsigned)p.CinPitch * (unsigned)sizeof(T), dps = (unsigned)p.Cout * (unsigned)sizeof(T);
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                const int iy = iy0 + x_hy[i], ix = ix0 + x_hx[i], lin = iy * p.W + ix;
                const bool ok = x_mode[i] != 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && lin < p.in_npix;
#pragma unroll
                for (int l = 0; l < LV; ++l) {
                    const bool okl = ok && (!X3 || xcb + 4 * l < p.Cin);       // (bf16x3: the half past a Cin % 8 == 4 tail is zero)
                    const uint4 v = *reinterpret_cast<const uint4*>(xb + (okl ? (unsigned)lin * xps + (unsigned)xcb * (unsigned)sizeof(T) + 16 * l : 0u));
                    rx[i * LV + l] = okl ? v : make_uint4(0, 0, 0, 0);
                }
                x_okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const int oy = oy0 + d_py[i], ox = ox0 + d_px[i];
                const int lin = (KS == 2 && p.phase_mode) ? (2 * oy + ph_a) * (2 * p.OW) + 2 * ox + ph_b : oy * p.OW + ox;
                const bool ok = d_mode[i] != 0 && oy < p.OH && ox < p.OW && ((KS == 2 && p.phase_mode) || lin < p.npix);
                const unsigned off = ok ? (unsigned)lin * dps + (unsigned)dcb * (unsigned)sizeof(T) : 0u;
#pragma unroll
                for (int l = 0; l < LV; ++l) {
                    const bool okl = ok && (!X3 || dcb + 4 * l < p.Cout);
                    const uint4 v = *reinterpret_cast<const uint4*>(db + (okl ? off + 16 * l : 0u));
                    rd[i * LV + l] = okl ? v : make_uint4(0, 0, 0, 0);
                }
                if constexpr (DYT) {
                    ry[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(
                        reinterpret_cast<const T*>(p.dyt_y) + (size_t)n * p.OH * p.OW * p.Cout) + off);
                    d_okmask |= ok ? (1u << i) : 0u;
                }
            }
        } else if constexpr (!X3) {       // (bf16x3 launches have whole, aligned pieces: wgrad_x3_eligible)
            const T* xin = reinterpret_cast<const T*>(p.x) + (size_t)n * p.H * p.W * p.CinPitch +
                           ((long)iy0 * p.W + ix0) * p.CinPitch;
            const T* dyin = reinterpret_cast<const T*>(p.dy) + (size_t)n * p.OH * p.OW * p.Cout +
                            ((long)oy0 * p.OW + ox0) * p.Cout;
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                rx[i] = make_uint4(0, 0, 0, 0);
                const int iy = iy0 + x_hy[i], ix = ix0 + x_hx[i];
                const bool ok = x_mode[i] != 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && iy * p.W + ix < p.in_npix;
                if (ok) {
                    x_okmask |= 1u << i;
                    const T* src = xin + (x_hy[i] * p.W + x_hx[i]) * p.CinPitch + xcb;
                    if (x_mode[i] == 1) rx[i] = *reinterpret_cast<const uint4*>(src);
                    else rx[i] = load_tail(src, p.Cin - xcb);
                }
            }
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                rd[i] = make_uint4(0, 0, 0, 0);
                const int oy = oy0 + d_py[i], ox = ox0 + d_px[i];
                const bool ok = d_mode[i] != 0 && oy < p.OH && ox < p.OW && oy * p.OW + ox < p.npix;
                if (ok) {
                    const T* src = dyin + (d_py[i] * p.OW + d_px[i]) * p.Cout + dcb;
                    if (d_mode[i] == 1) rd[i] = *reinterpret_cast<const uint4*>(src);
                    else rd[i] = load_tail(src, p.Cout - dcb);
                }
            }
        }
        f_tx += dtx;
        if (f_tx >= p.tilesX) { f_tx -= p.tilesX; ++f_ty; }
        f_ty += dty;
        if (f_ty >= p.tilesY) { f_ty -= p.tilesY; ++f_n; }
        f_n += dn;
    };
    auto commit = [&]() {
        if constexpr (X3) {
            float sc[8], sh[8];
            if (p.in_scale) {
                const float* sp = bnp + c_bng * 2 * ITT + (tid % XQ) * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                    *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + ITT + j);
                }
            }
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                const int idx = tid + i * NTHR;
                if (idx >= HH * HWD * XQ) continue;
                float v[8];
                unpack16(rx[2 * i], reinterpret_cast<float(&)[4]>(v[0]));
                unpack16(rx[2 * i + 1], reinterpret_cast<float(&)[4]>(v[4]));
                if (p.in_scale && ((x_okmask >> i) & 1u)) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                }
                uint4 hi, lo;
                split_bf16x3(v, hi, lo);
                *reinterpret_cast<uint4*>(halo + (idx / XQ) * XP + (idx % XQ) * 16) = hi;
                *reinterpret_cast<uint4*>(halo + HPB + (idx / XQ) * XP + (idx % XQ) * 16) = lo;
            }
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const int idx = tid + i * NTHR;
                if (idx >= TH * TW * DQ) continue;
                uint4 hi, lo;
                split_bf16x3(rd[2 * i], rd[2 * i + 1], hi, lo);
                *reinterpret_cast<uint4*>(dyt + (idx / DQ) * DP + (idx % DQ) * 16) = hi;
                *reinterpret_cast<uint4*>(dyt + DPB + (idx / DQ) * DP + (idx % DQ) * 16) = lo;
            }
        } else {
        if (p.in_scale) {          // x = relu(x * scale + shift) on its way into LDS; this thread's pieces share their channels
            float sc[EPV], sh[EPV];
            const float* sp = bnp + c_bng * 2 * ITT + (tid % XQ) * EPV;
#pragma unroll
            for (int j = 0; j < EPV; j += 4) {
                *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + ITT + j);
            }
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                if (!((x_okmask >> i) & 1u)) continue;
                float v[EPV];
                unpack16(rx[i], v);
#pragma unroll
                for (int j = 0; j < EPV; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                rx[i] = pack16<T>(v);
            }
        }
        if constexpr (DYT) {       // dy = A * d + B * y + C on its way into LDS; this thread's pieces share their channels
            float ca[EPV], cb[EPV], cc[EPV];
            const float* sp = bnp + c_bng * 3 * CT + (tid % DQ) * EPV;
#pragma unroll
            for (int j = 0; j < EPV; j += 4) {
                *reinterpret_cast<float4*>(ca + j) = *reinterpret_cast<const float4*>(sp + j);
                *reinterpret_cast<float4*>(cb + j) = *reinterpret_cast<const float4*>(sp + CT + j);
                *reinterpret_cast<float4*>(cc + j) = *reinterpret_cast<const float4*>(sp + 2 * CT + j);
            }
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                if (!((d_okmask >> i) & 1u)) continue;
                float dv[EPV], yv[EPV];
                unpack16(rd[i], dv);
                unpack16(ry[i], yv);
#pragma unroll
                for (int j = 0; j < EPV; ++j) dv[j] = ca[j] * dv[j] + (cb[j] * yv[j] + cc[j]);
                rd[i] = pack16<T>(dv);
            }
        }
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int idx = tid + i * NTHR;
            if (idx < HH * HWD * XQ) *reinterpret_cast<uint4*>(halo + (idx / XQ) * XP + (idx % XQ) * 16) = rx[i];
        }
#pragma unroll
        for (int i = 0; i < NDV; ++i) {
            const int idx = tid + i * NTHR;
            if (idx < TH * TW * DQ) *reinterpret_cast<uint4*>(dyt + (idx / DQ) * DP + (idx % DQ) * 16) = rd[i];
        }
        }      // !X3
    };

    if (p.in_scale) {
        for (int i = tid; i < p.in_groups * ITT; i += NTHR) {
            const int gi = i / ITT, c = i % ITT;
            const bool ok = ci0 + c < p.Cin;
            bnp[(gi * 2 + 0) * ITT + c] = ok ? p.in_scale[gi * p.Cin + ci0 + c] : 0.f;
            bnp[(gi * 2 + 1) * ITT + c] = ok ? p.in_shift[gi * p.Cin + ci0 + c] : 0.f;
        }
        __syncthreads();
    }
    if constexpr (DYT) {
        for (int i = tid; i < p.dyt_groups * 3 * CT; i += NTHR) {
            const int gk = i / CT, c = i % CT;
            bnp[i] = co0 + c < p.Cout ? p.dytoef[(size_t)gk * p.Cout + co0 + c] : 0.f;
        }
        __syncthreads();
    }
    if (kz < ntiles) fetch();
    for (int tile = kz; tile < ntiles; tile += p.splitk) {
        commit();
        __syncthreads();
        if (tile + p.splitk < ntiles) fetch();

        if constexpr (LE == 4) {
            // 4 pixels per MFMA: lane (pl, g) supplies pixel k0+g, channel pl
            for (int k0 = kq * KPW; k0 < (kq + 1) * KPW; k0 += 4) {
                const int k = k0 + g, row = k / TW, col = k % TW;
                const float a = *reinterpret_cast<const float*>(dyt + k * DP + (cw * 16 + pl) * 4);
#pragma unroll
                for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                    for (int kw = 0; kw < KS; ++kw) {
                        const int hp = (row * STRIDE + kh * DIL) * HWD + col * STRIDE + kw * DIL;
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const float b = *reinterpret_cast<const float*>(halo + hp * XP + (cig * IT + i * 16 + pl) * 4);
                            acc[kh * KS + kw][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[kh * KS + kw][i], 0, 0, 0);
                        }
                    }
            }
        } else {
            // 32 pixels (two tile rows) per MFMA: lane group g supplies columns 4g..4g+3 of rows r0 and r0+1,
            // so the 32 lanes of a half-wave read 8 consecutive pixels per transpose-read (see pitch note).
            // The k-steps are fully unrolled from per-lane base addresses: every LDS read is base + immediate.
            const int c0 = g * 4;
            // WCI (the 512-thread 64co x 64ci form): a wave owns ONE 16-wide ci sub-tile (wv) and TWO co sub-tiles (2 cig,
            // 2 cig + 1) instead of one co sub-tile and two ci sub-tiles -- per k-step it reads 2 dY + 9 x fragments for its 18
            // MFMAs instead of 1 + 18 (the x fragments were read by all four co waves): 22 instead of 38 transpose reads per
            // 18 MFMAs, and the LDS pipe was as busy as the matrix pipe.
            const int a_co = WCI ? 2 * cig * 16 : cw * 16, b_ci = WCI ? wv * 16 : cig * IT;
            const unsigned char* a_base = TR ? dyt + (kq * KPW + c0 + (pl >> 2)) * DP + (a_co + (pl & 3) * 4) * 2
                                             : dyt + (kq * KPW + c0) * DP + (a_co + pl) * 2;
            // (stride 2: the four pixels of a transpose read are every other halo pixel -- each lane supplies its own address,
            // so the read serves any stride; the former eight ds_read_u16 per fragment made the stride-2 launches LDS-issue bound)
            const unsigned char* b_base = (TR
                ? halo + (kq * (KPW / TW) * STRIDE * HWD + (c0 + (pl >> 2)) * STRIDE) * XP + ((pl & 3) * 4) * 2
                : halo + (kq * (KPW / TW) * STRIDE * HWD + c0 * STRIDE) * XP + pl * 2) + b_ci * 2;
#pragma unroll
            for (int kk = 0; kk < KPW; kk += 32) {
                auto load_a = [&](int cofs, int plane = 0) {          // dY fragment of the co sub-tile `cofs` bytes further
                    F8 a;
                    if constexpr (TR) {
                        // lane p of each 16-lane group points at the 8-byte piece (pixel p/4, channels 4*(p%4)..+3)
                        a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(a_base + plane * DPB + kk * DP + cofs));
                        a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(a_base + plane * DPB + (kk + TW) * DP + cofs));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            a.s[j] = *reinterpret_cast<const unsigned short*>(a_base + (kk + (j >> 2) * TW + (j & 3)) * DP + cofs);
                    }
                    return a;
                };
                auto load_b = [&](int hp, int cofs, int plane = 0) {  // x fragment at halo pixel offset hp, ci sub-tile `cofs` bytes further
                    F8 b;
                    if constexpr (TR) {
                        b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(b_base + plane * HPB + hp * XP + cofs));
                        b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4*)(b_base + plane * HPB + (hp + STRIDE * HWD) * XP + cofs));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            b.s[j] = *reinterpret_cast<const unsigned short*>(
                                b_base + (hp + (j >> 2) * STRIDE * HWD + (j & 3) * STRIDE) * XP + cofs);
                    }
                    return b;
                };
                if constexpr (X3 && WCI) {
                    const F8 a0 = load_a(0), a1 = load_a(32), a0l = load_a(0, 1), a1l = load_a(32, 1);
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                        for (int kw = 0; kw < KS; ++kw) {
                            const int hp = ((kk / TW) * STRIDE + kh * DIL) * HWD + kw * DIL;      // compile-time
                            const F8 b = load_b(hp, 0), bl = load_b(hp, 0, 1);
                            f32x4& c0_ = acc[kh * KS + kw][0];
                            f32x4& c1_ = acc[kh * KS + kw][1];
                            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v, b.v, c0_, 0, 0, 0);
                            c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, b.v, c1_, 0, 0, 0);
                            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l.v, b.v, c0_, 0, 0, 0);
                            c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l.v, b.v, c1_, 0, 0, 0);
                            c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v, bl.v, c0_, 0, 0, 0);
                            c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, bl.v, c1_, 0, 0, 0);
                        }
                } else if constexpr (X3) {
                    const F8 a = load_a(0), al = load_a(0, 1);
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                        for (int kw = 0; kw < KS; ++kw) {
                            const int hp = ((kk / TW) * STRIDE + kh * DIL) * HWD + kw * DIL;      // compile-time
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const F8 b = load_b(hp, i * 32), bl = load_b(hp, i * 32, 1);
                                f32x4& c_ = acc[kh * KS + kw][i];
                                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c_, 0, 0, 0);
                                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, b.v, c_, 0, 0, 0);
                                c_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bl.v, c_, 0, 0, 0);
                            }
                        }
                } else if constexpr (WCI) {
                    const F8 a0 = load_a(0), a1 = load_a(32);
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                        for (int kw = 0; kw < KS; ++kw) {
                            const int hp = ((kk / TW) * STRIDE + kh * DIL) * HWD + kw * DIL;      // compile-time
                            const F8 b = load_b(hp, 0);
                            acc[kh * KS + kw][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v, b.v, acc[kh * KS + kw][0], 0, 0, 0);
                            acc[kh * KS + kw][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, b.v, acc[kh * KS + kw][1], 0, 0, 0);
                        }
                } else {
                    const F8 a = load_a(0);
#pragma unroll
                    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                        for (int kw = 0; kw < KS; ++kw) {
                            const int hp = ((kk / TW) * STRIDE + kh * DIL) * HWD + kw * DIL;      // compile-time
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const F8 b = load_b(hp, i * 32);
                                acc[kh * KS + kw][i] =
                                    __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[kh * KS + kw][i], 0, 0, 0);
                            }
                        }
                }
            }
        }
        __syncthreads();
    }

    if constexpr (KSPLIT > 1) {
        // add the wave groups that split the pixels (fixed order): group q > 0 parks its accumulators in LDS
        float* red = reinterpret_cast<float*>(smem);            // [CW][TAPS*NI][64 lanes][4]   (host sized the LDS)
        for (int q = 1; q < KSPLIT; ++q) {
            if (kq == q) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        *reinterpret_cast<f32x4*>(red + (((size_t)cw * TAPS * NI + t * NI + i) * 64 + lane) * 4) = acc[t][i];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t)
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const f32x4 o = *reinterpret_cast<const f32x4*>(red + (((size_t)cw * TAPS * NI + t * NI + i) * 64 + lane) * 4);
                        acc[t][i][0] += o[0]; acc[t][i][1] += o[1]; acc[t][i][2] += o[2]; acc[t][i][3] += o[3];
                    }
            }
            __syncthreads();
        }
        if (kq != 0) return;
    }
    // partial slab: [grp][kz][tap][Cout][Cin]
    // (one 64-bit base + 32-bit offsets, bounds hoisted: the slab of one workgroup is far below 2^31 elements)
    // accumulator tile i of a lane: rows co = cob + j (j < 4), column ci = cib, advancing by `istep` elements per i:
    // 16 input channels (the ci sub-tiles of a wave) or, WCI, 16 output channels (its two co sub-tiles)
    const int cob = co0 + (WCI ? 2 * cig * 16 : cw * 16) + g * 4, cib = ci0 + (WCI ? wv * 16 : cig * IT) + pl;
    float* out = p.part + ((size_t)grp * p.splitk + kz) * TAPS * p.Cout * p.Cin + (size_t)cob * p.Cin + cib;
    const int tstride = p.Cout * p.Cin;
    const int istep = WCI ? 16 * p.Cin : 16;
    bool okij[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            okij[i][j] = WCI ? (cob + i * 16 + j < p.CoutUse && cib < p.Cin) : (cob + j < p.CoutUse && cib + i * 16 < p.Cin);
    if (p.direct == 2) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (okij[i][j]) out[t * tstride + j * p.Cin + i * istep] += acc[t][i][j];
    } else {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (okij[i][j]) out[t * tstride + j * p.Cin + i * istep] = acc[t][i][j];
    }
}
template <typename T, int KS, int STRIDE, int IT, bool TR, int DIL, int CTT, int CIG, bool DYT = false>
__global__ __launch_bounds__(256 * CIG) void conv_wgrad_kernel(WgArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    wg_body<T, KS, STRIDE, IT, TR, DIL, CTT, CIG, DYT>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, blockIdx.z, smem);
}
// several layers of ONE instantiation in one launch (see conv_wgrad_ws_multi_kernel, whose argument block this shares)
constexpr int WS_MAXJ = 16;
struct WsMulti {
    int njobs;
    int first[WS_MAXJ + 1];       // multiples of 8 (the XCD-aware order inside a layer relies on it)
    WgArgs a[WS_MAXJ];
};
static_assert(sizeof(WsMulti) <= 4096, "the kernel argument block");
template <typename T, int KS, int STRIDE, int IT, bool TR, int DIL, int CTT, int CIG>
__global__ __launch_bounds__(256 * CIG) void conv_wgrad_multi_kernel(WsMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int j = 0;
    while (j + 1 < m.njobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const WgArgs p = m.a[j];
    const int nbx = ((p.CoutUse + CTT - 1) / CTT) * p.ci_tiles, local = (int)blockIdx.x - m.first[j];
    if (local >= nbx * p.splitk) return;              // padding up to the next multiple of 8
    wg_body<T, KS, STRIDE, IT, TR, DIL, CTT, CIG, false>(p, local % nbx, local / nbx, nbx, p.splitk, 0, smem);
}

// dw_oihw[g][o][i][tap] (+)= sum_kz part[g][kz][tap][o][i].  A workgroup = 8 split-K phases x 32 lanes; a lane owns FOUR
// consecutive slab elements (one 16-byte load per slab; Cin % 4 == 0) or one (ragged Cin), i.e. 128 / 32 outputs per
// workgroup -- wgrad_reduce_epb() is the host's side of that contract.  The slab reads of a step are ~250-500 MB: the
// former 4-byte-per-lane form ran at ~3.3 TB/s.
static inline int wgrad_reduce_epb(int I) { return (I & 3) == 0 ? 128 : 32; }
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ part, int splitk, int taps, int Oslab, int O,
                                                   int I, float* __restrict__ dw, int accumulate, long block, float (*red)[132]) {
    const long n = (long)O * I * taps;
    const int lane = threadIdx.x & 31, ph = threadIdx.x >> 5;
    const bool vec = (I & 3) == 0;
    const long i = vec ? (block * 32 + lane) * 4 : block * 32 + lane;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // lanes walk the SLAB order (tap, o, ci) so the splitk reads are coalesced; the single OIHW write scatters
    int tap = 0, ci = 0, o = 0;
    if (i < n) {
        ci = (int)(i % I);
        o = (int)((i / I) % O);
        tap = (int)(i / ((long)I * O));
        const size_t slab = (size_t)taps * Oslab * I;
        const float* src = part + ((size_t)tap * Oslab + o) * I + ci;
        if (vec) {
            // four independent partial sums (otherwise the slab loop is a chain of dependent-latency loads)
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            auto add = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
            int k = ph;
            for (; k + 24 < splitk; k += 32) {
                add(a0, *reinterpret_cast<const float4*>(src + (size_t)k * slab));
                add(a1, *reinterpret_cast<const float4*>(src + (size_t)(k + 8) * slab));
                add(a2, *reinterpret_cast<const float4*>(src + (size_t)(k + 16) * slab));
                add(a3, *reinterpret_cast<const float4*>(src + (size_t)(k + 24) * slab));
            }
            for (; k < splitk; k += 8) add(a0, *reinterpret_cast<const float4*>(src + (size_t)k * slab));
            acc[0] = (a0.x + a1.x) + (a2.x + a3.x); acc[1] = (a0.y + a1.y) + (a2.y + a3.y);
            acc[2] = (a0.z + a1.z) + (a2.z + a3.z); acc[3] = (a0.w + a1.w) + (a2.w + a3.w);
        } else {
            float a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = ph;
            for (; k + 24 < splitk; k += 32) {
                acc[0] += src[(size_t)k * slab];
                a1 += src[(size_t)(k + 8) * slab];
                a2 += src[(size_t)(k + 16) * slab];
                a3 += src[(size_t)(k + 24) * slab];
            }
            for (; k < splitk; k += 8) acc[0] += src[(size_t)k * slab];
            acc[0] = (acc[0] + a1) + (a2 + a3);
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ph][lane * 4 + e] = acc[e];
    __syncthreads();
    // 128 (vec) / 32 outputs: thread t < 128 finishes element t = lane' * 4 + e
    const int t = threadIdx.x;
    if (t < (vec ? 128 : 32)) {
        const int l2 = vec ? t >> 2 : t, e = vec ? t & 3 : 0;
        const long i2 = vec ? (block * 32 + l2) * 4 + e : block * 32 + l2;
        if (i2 < n) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) s += red[r][l2 * 4 + e];
            const int ci2 = (int)(i2 % I), o2 = (int)((i2 / I) % O), tap2 = (int)(i2 / ((long)I * O));
            const size_t d = ((size_t)o2 * I + ci2) * taps + tap2;
            if (accumulate) dw[d] += s; else dw[d] = s;
        }
    }
}
__global__ void wgrad_reduce_oihw_kernel(const float* __restrict__ part, int splitk, int taps, int Oslab, int O, int I,
                                         float* __restrict__ dw, int accumulate) {      // blockIdx.y = group
    __shared__ float red[8][132];
    const long n = (long)O * I * taps;
    wgrad_reduce_block(part + (size_t)blockIdx.y * splitk * taps * Oslab * I, splitk, taps, Oslab, O, I,
                       dw + (size_t)blockIdx.y * n, accumulate, blockIdx.x, red);
}
// every deferred split-K reduce of a backward pass in ONE launch (dh_wgrad_reduce_multi): record k is served by
// workgroups [first_block, first_block + nblocks)
struct WgReduceJob {
    const float* part;
    float* dw;
    int splitk, taps, Oslab, O, I, accumulate, first_block, nblocks;
};
__global__ void wgrad_reduce_multi_kernel(const WgReduceJob* __restrict__ jobs, int njobs) {
    __shared__ float red[8][132];
    int j = 0;
    while (j + 1 < njobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
    const WgReduceJob job = jobs[j];
    wgrad_reduce_block(job.part, job.splitk, job.taps, job.Oslab, job.O, job.I, job.dw, job.accumulate,
                       (long)blockIdx.x - job.first_block, red);
}

// ---- wave-specialised 64co x 64ci weight gradient of the 3x3 stride-1 layers (bf16, Cin % 64 == Cout % 64 == 0) ----------
// The per-workgroup clock of conv_wgrad_kernel (tools/wgrad_timeline.py) shows 2.24 us per 128-pixel tile against 0.96 us of MFMA
// issue: 0.5 us of it is the address arithmetic + load issue of the next tile, which every wave runs BEFORE its MFMAs (both
// waves of a SIMD in lock step, the matrix pipe idle), 0.7 us are two barriers, 0.2 us the LDS commit.  Here the eight waves
// split the roles: waves 4-7 are PRODUCERS (all global loads, BatchNorm-on-load, LDS commits; no accumulators), waves 0-3 are
// CONSUMERS (one per SIMD: ci sub-tile wv x all four co sub-tiles = 36 accumulator tiles, 144 MFMAs per tile back to back, 13
// fragment loads per 36 MFMAs).  Two LDS stages, ONE raw s_barrier per tile (no vmcnt drain): the producers' VALU / memory work
// runs on the SIMD's other issue slots while its consumer wave keeps the matrix pipe busy.
#ifdef DH_WS_TIMING
__device__ long long g_ws[4096 * 16];
#define WS_NOW() ((long long)wall_clock64())
#define WS_T(...) __VA_ARGS__
#else
#define WS_T(...)
#endif
// the workgroup (bx, kz) of an nbx x nkz grid (nkz = p.splitk): 64co x 64ci block bx, K slice kz
template <int DIL>
__device__ __forceinline__ void ws_body(const WgArgs& p, int bx, int kz, const int nbx, const int nkz, unsigned char* smem) {
    constexpr int KS = 3, TAPS = 9;
    constexpr int HH = (TH - 1) + (KS - 1) * DIL + 1, HWD = (TW - 1) + (KS - 1) * DIL + 1;
    constexpr int XP = lds_pitch(128), DP = lds_pitch(128);           // 64 channels x 2 B, odd multiple of 32 B
    constexpr int STAGE = HH * HWD * XP + TH * TW * DP;
    float* bnp = reinterpret_cast<float*>(smem + 2 * STAGE);          // in_scale: [in_groups][2][64]

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int pl = lane & 15, g = lane >> 4;
    if (nbx > 1 && (nkz & 7) == 0 && !p.no_xcd_remap) {
        const unsigned lin = (unsigned)(kz * nbx + bx), xcd = lin & 7, sq = lin >> 3;
        bx = (int)(sq % nbx);
        kz = (int)((sq / nbx) * 8 + xcd);
    }
    const int cot = bx / p.ci_tiles, cit = bx % p.ci_tiles;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int tiles_per_img = p.tilesX * p.tilesY;
    const int ntiles = p.N * tiles_per_img;
    const int my_tiles = kz < ntiles ? (ntiles - kz + p.splitk - 1) / p.splitk : 0;

    if (p.in_scale) {
        for (int i = tid; i < p.in_groups * 128; i += 512) {
            const int gi = i >> 7, k = (i >> 6) & 1, c = i & 63;
            bnp[i] = (k ? p.in_shift : p.in_scale)[gi * p.Cin + ci0 + c];
        }
        __syncthreads();
    }

    if (wv >= 4) {
        // ================= producers =================
        const int pt = tid - 256, q = pt & 7, prow = pt >> 3;          // this thread's pieces: channel piece q of pixels prow + 32 i
        constexpr int NXV = (HH * HWD + 31) / 32, NDV = TH * TW / 32;
        // two register sets: the loads of tile it + 2 are issued right after tile it is committed, i.e. they have two tile times
        // (~2 us) to land -- with one set the producers waited on HBM latency every tile
        struct Regs { uint4 rx[NXV], rd[NDV]; unsigned okmask; int bng; };
        Regs RA, RB;
        int x_h[NXV];                                                 // hy << 8 | hx, or -1 past the halo
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int px = prow + 32 * i;
            x_h[i] = px < HH * HWD ? ((px / HWD) << 8) | (px % HWD) : -1;
        }
        const unsigned xps = (unsigned)p.CinPitch * 2u, dps = (unsigned)p.Cout * 2u;
        // split input: this ci tile's 64 channels lie in ONE of the two tensors (CinPitch = Cin / 2 channels per pixel)
        const bool upper = p.x_split && ci0 >= p.Cin / 2;
        const long xsel = upper ? p.x_split : 0L;
        const unsigned xcb = (unsigned)((upper ? ci0 - p.Cin / 2 : ci0) + q * 8) * 2u, dcb = (unsigned)(co0 + q * 8) * 2u;
        const int dn = p.splitk / tiles_per_img, drem = p.splitk % tiles_per_img;
        const int dty = drem / p.tilesX, dtx = drem % p.tilesX;
        int f_n = kz / tiles_per_img, f_ty = (kz % tiles_per_img) / p.tilesX, f_tx = (kz % tiles_per_img) % p.tilesX;
        auto fetch = [&](Regs& R) {
            const int n = f_n;
            R.bng = p.in_scale ? n / (p.N / p.in_groups) : 0;
            const int oy0 = f_ty * TH, ox0 = f_tx * TW, iy0 = oy0 - p.pad, ix0 = ox0 - p.pad;
            const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + xsel + (size_t)n * p.H * p.W * xps;
            const unsigned char* db = reinterpret_cast<const unsigned char*>(p.dy) + (size_t)n * p.OH * p.OW * dps;
            R.okmask = 0;
#pragma unroll
            for (int i = 0; i < NXV; ++i) {
                const int iy = iy0 + (x_h[i] >> 8), ix = ix0 + (x_h[i] & 0xff);
                const bool ok = x_h[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint4 v = *reinterpret_cast<const uint4*>(xb + (ok ? (unsigned)(iy * p.W + ix) * xps + xcb : 0u));
                R.rx[i] = ok ? v : make_uint4(0, 0, 0, 0);
                R.okmask |= ok ? (1u << i) : 0u;
            }
#pragma unroll
            for (int i = 0; i < NDV; ++i) {
                const int px = prow + 32 * i, oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
                const bool ok = oy < p.OH && ox < p.OW;
                const uint4 v = *reinterpret_cast<const uint4*>(db + (ok ? (unsigned)(oy * p.OW + ox) * dps + dcb : 0u));
                R.rd[i] = ok ? v : make_uint4(0, 0, 0, 0);
            }
            f_tx += dtx;
            if (f_tx >= p.tilesX) { f_tx -= p.tilesX; ++f_ty; }
            f_ty += dty;
            if (f_ty >= p.tilesY) { f_ty -= p.tilesY; ++f_n; }
            f_n += dn;
        };
        auto commit = [&](Regs& R, int stage) {
            unsigned char* halo = smem + stage * STAGE;
            unsigned char* dyt = halo + HH * HWD * XP;
            if (p.in_scale) {          // x = relu(x * scale + shift) on its way into LDS (padding stays zero)
                float sc[8], sh[8];
                const float* sp = bnp + R.bng * 128 + q * 8;
#pragma unroll
                for (int j = 0; j < 8; j += 4) {
                    *reinterpret_cast<float4*>(sc + j) = *reinterpret_cast<const float4*>(sp + j);
                    *reinterpret_cast<float4*>(sh + j) = *reinterpret_cast<const float4*>(sp + 64 + j);
                }
#pragma unroll
                for (int i = 0; i < NXV; ++i) {
                    if (!((R.okmask >> i) & 1u)) continue;
                    float v[8];
                    unpack16(R.rx[i], v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaxf(v[j] * sc[j] + sh[j], 0.f);
                    R.rx[i] = pack16<bf16>(v);
                }
            }
#pragma unroll
            for (int i = 0; i < NXV; ++i)
                if (x_h[i] >= 0) *reinterpret_cast<uint4*>(halo + (prow + 32 * i) * XP + q * 16) = R.rx[i];
#pragma unroll
            for (int i = 0; i < NDV; ++i) *reinterpret_cast<uint4*>(dyt + (prow + 32 * i) * DP + q * 16) = R.rd[i];
        };
        WS_T(long long ts[4] = {0, 0, 0, 0}; const long long tbeg = WS_NOW();)
        if (my_tiles > 0) fetch(RA);
        if (my_tiles > 1) fetch(RB);
        for (int it = 0; it < my_tiles; it += 2) {
            WS_T(long long t0 = WS_NOW();)
            commit(RA, 0);
            WS_T(long long t1 = WS_NOW(); ts[0] += t1 - t0;)
            if (it + 2 < my_tiles) fetch(RA);
            WS_T(t0 = WS_NOW(); ts[1] += t0 - t1;)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            WS_T(ts[2] += WS_NOW() - t0;)
            if (it + 1 < my_tiles) {
                WS_T(t0 = WS_NOW();)
                commit(RB, 1);
                WS_T(t1 = WS_NOW(); ts[0] += t1 - t0;)
                if (it + 3 < my_tiles) fetch(RB);
                WS_T(t0 = WS_NOW(); ts[1] += t0 - t1;)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                WS_T(ts[2] += WS_NOW() - t0;)
            }
        }
        WS_T(if (tid == 256) { long long* o = g_ws + (blockIdx.y * gridDim.x + blockIdx.x) % 4096 * 16 + 8; o[0] = ts[0]; o[1] = ts[1]; o[2] = ts[2]; o[3] = WS_NOW() - tbeg; })
        return;
    }

    // ================= consumers: ci sub-tile wv, co sub-tiles 0..3 =================
    f32x4 acc[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c0 = g * 4;
    WS_T(long long tc[3] = {0, 0, 0}; const long long tbeg = WS_NOW();)
    for (int it = 0; it < my_tiles; ++it) {
        WS_T(long long t0 = WS_NOW();)
        __builtin_amdgcn_s_barrier();             // stage it & 1 is committed; the producers now refill the other one
        WS_T(long long t1 = WS_NOW(); tc[0] += t1 - t0;)
        const unsigned char* halo = smem + (it & 1) * STAGE;
        const unsigned char* dyt = halo + HH * HWD * XP;
        // lane p of each 16-lane group points at the 8-byte piece (pixel p/4, channels 4*(p%4)..+3): ds_read_b64_tr_b16
        const unsigned char* a_base = dyt + (c0 + (pl >> 2)) * DP + ((pl & 3) * 4) * 2;
        const unsigned char* b_base = halo + (c0 + (pl >> 2)) * XP + ((pl & 3) * 4) * 2 + wv * 32;
        // (Left to the compiler's scheduling: a hand-pipelined version -- x fragment two steps ahead, dY fragments of the next
        // k-step at tap 5, sched_barriers around each MFMA group -- was slower, 1.66 vs 1.44 us per tile.  144 MFMAs are
        // 0.96 us at 2.4 GHz; the chip runs this phase at ~80 % of that on random operands.)
#pragma unroll
        for (int kk = 0; kk < TH * TW; kk += 32) {
            F8 a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a[s].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a_base + kk * DP + s * 32));
                a[s].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a_base + (kk + TW) * DP + s * 32));
            }
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const int hp = ((kk / TW) + kh * DIL) * HWD + kw * DIL;      // compile-time
                    F8 b;
                    b.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b_base + hp * XP));
                    b.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b_base + (hp + HWD) * XP));
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[kh * KS + kw][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s].v, b.v, acc[kh * KS + kw][s], 0, 0, 0);
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS reads of the stage are complete
        WS_T(tc[1] += WS_NOW() - t1;)
    }
    WS_T(const long long tloop = WS_NOW();)
    // partial slab [kz][tap][Cout][Cin]: lane (pl, g) holds rows co = 16 s + 4 g + j, column ci = 16 wv + pl
    float* out = p.part + (size_t)kz * TAPS * p.Cout * p.Cin + (size_t)(co0 + g * 4) * p.Cin + ci0 + wv * 16 + pl;
    const int tstride = p.Cout * p.Cin;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[t * tstride + (s * 16 + j) * p.Cin] = acc[t][s][j];
    WS_T(if (tid == 0) { long long* o = g_ws + (blockIdx.y * gridDim.x + blockIdx.x) % 4096 * 16; o[0] = tc[0]; o[1] = tc[1]; o[2] = tloop - tbeg; o[3] = WS_NOW() - tloop; o[4] = my_tiles; })
}
template <int DIL>
__global__ __launch_bounds__(512) void conv_wgrad_ws_kernel(WgArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ws_body<DIL>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y, smem);
}
// The weight gradients of SEVERAL layers in one launch (dh_wgrad_batch_*): the layers' arguments travel by value in the kernel
// argument block (a recorded HIP graph keeps them; no device table whose activation pointers would differ between the warm-up
// pass and the capture), workgroups [first[j], first[j + 1]) serve layer j.  One launch per backward pass instead of one per
// layer lets every layer take FEWER, LONGER K slices (a launch of its own needs 256 workgroups to fill the chip: 8 pixel tiles
// per workgroup on the 64-channel layers, and a 147 KB partial slab each) -- the chip is filled by the sum of the layers.
__global__ __launch_bounds__(512) void conv_wgrad_ws_multi_kernel(WsMulti m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int j = 0;
    while (j + 1 < m.njobs && (int)blockIdx.x >= m.first[j + 1]) ++j;
    const WgArgs p = m.a[j];
    const int nbx = (p.Cout / 64) * p.ci_tiles, local = (int)blockIdx.x - m.first[j];
    if (local >= nbx * p.splitk) return;              // padding up to the next multiple of 8
    ws_body<1>(p, local % nbx, local / nbx, nbx, p.splitk, smem);
}
#ifdef DH_WS_TIMING
extern "C" int dh_debug_ws(long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ws), (size_t)n * 8); }
#endif
template <int DIL>
int launch_ws(const WgArgs& a, hipStream_t st) {
    constexpr int HH = (TH - 1) + 2 * DIL + 1, HWD = (TW - 1) + 2 * DIL + 1;
    const size_t lds = 2 * ((size_t)HH * HWD * lds_pitch(128) + (size_t)TH * TW * lds_pitch(128)) +
                       (a.in_scale ? (size_t)a.in_groups * 128 * sizeof(float) : 0);
    static bool attr_done = false;
    if (!attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ws_kernel<DIL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds + 4096) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wgrad_ws: cannot raise dynamic LDS to %zu", lds);
        }
    }
    dim3 grid((a.Cout / 64) * a.ci_tiles, a.splitk, 1);
    hipLaunchKernelGGL(conv_wgrad_ws_kernel<DIL>, grid, dim3(512), lds, st, a);
    DH_CHECK_LAUNCH("conv_wgrad_ws");
    return 0;
}
// eligibility of the wave-specialised form (DAHITRA_WGRAD_NO_WS=1: A/B switch back to conv_wgrad_kernel)
static inline bool ws_eligible(const WgArgs& a, int ks, int stride, bool bf16, bool tr) {
    static const bool off = getenv("DAHITRA_WGRAD_NO_WS") != nullptr;
    return !off && bf16 && tr && ks == 3 && stride == 1 && a.groups == 1 && !a.phase_mode && !a.dyt_y && a.Cin % 64 == 0 &&
           a.Cout % 64 == 0 && a.CoutUse == a.Cout && a.npix == a.OH * a.OW && a.in_npix == a.H * a.W &&
           (a.x_split ? (a.CinPitch * 2 == a.Cin && a.Cin % 128 == 0 && a.in_scale == nullptr) : a.CinPitch == a.Cin) &&
           (a.in_scale == nullptr || a.in_groups <= 8);
}

// ---- batched form (dh_wgrad_batch_begin / _launch / _end): eligible layers are collected, one conv_wgrad_ws_multi_kernel serves them
struct WsBatch {
    bool on = false;
    int n = 0;
    size_t lds = 0;
    WsMulti m;
};
static thread_local WsBatch g_wsb;       // wave-specialised 64co x 64ci layers (`on` = a batch is open, for both)
static thread_local WsBatch g_c32b;      // 3x3 stride-1 bf16 layers with a 32-wide co tile (conv_wgrad_kernel<bf16, 3, 1, 32, true, 1, 32, 1>)
// K slices of a layer inside a batch: ~32 pixel tiles per workgroup (DAHITRA_WGRAD_TPW), 64 for the layers of >= 16 blocks
// (DAHITRA_WGRAD_TPW_BIG: they come first in a backward pass, the shorter workgroups of the later layers fill the tail; measured
// 32 / 32: 9017 + 5620 pairs/s (s4 + newUNetTrans), 32 / 64: 9085 + 5621, 32 / 128: 9117 + 5565, 64 / 64: 9018 + 5553, 16 / 16:
// 8923), never more slices than the layer's own launch would take, a multiple of 8 from 8 on (XCD-aware order)
static int ws_batch_splitk(const WgArgs& a) {
    static const int tpw = getenv("DAHITRA_WGRAD_TPW") ? atoi(getenv("DAHITRA_WGRAD_TPW")) : 32;
    static const int tpw_big = getenv("DAHITRA_WGRAD_TPW_BIG") ? atoi(getenv("DAHITRA_WGRAD_TPW_BIG")) : 2 * tpw;
    const long tiles = (long)a.N * a.tilesX * a.tilesY;
    const int t = (a.Cout / 64) * a.ci_tiles >= 16 ? tpw_big : tpw;
    long sk = tiles / (t > 0 ? t : 32);
    if (sk > a.splitk) sk = a.splitk;
    if (sk >= 8) sk &= ~7L;
    return (int)(sk < 1 ? 1 : sk);
}
// longest workgroups first (the dispatcher hands out workgroups in index order: the short ones then fill the tail of the launch)
static void batch_sort(WsBatch& b, int co_tile_) {
    if (getenv("DAHITRA_WGRAD_NO_SORT")) return;
    auto tpw = [](const WgArgs& a) { return ((long)a.N * a.tilesX * a.tilesY + a.splitk - 1) / a.splitk; };
    for (int i = 1; i < b.n; ++i)                  // insertion sort, stable: at most 16 entries
        for (int j = i; j > 0 && tpw(b.m.a[j]) > tpw(b.m.a[j - 1]); --j) { const WgArgs t = b.m.a[j]; b.m.a[j] = b.m.a[j - 1]; b.m.a[j - 1] = t; }
    b.m.first[0] = 0;
    for (int i = 0; i < b.n; ++i) {
        const int nblocks = dh_cdiv(b.m.a[i].CoutUse, co_tile_) * b.m.a[i].ci_tiles * b.m.a[i].splitk;
        b.m.first[i + 1] = b.m.first[i] + ((nblocks + 7) & ~7);
    }
}
static int ws_batch_flush(hipStream_t st) {
    WsBatch& b = g_wsb;
    if (b.n == 0) return 0;
    b.m.njobs = b.n;
    batch_sort(b, 64);
    const int total = b.m.first[b.n];
    static bool attr_done = false;      // once, for the largest form (ws_eligible: at most 8 BatchNorm groups): never inside a capture
    if (!attr_done) {
        attr_done = true;
        constexpr int HH = (TH - 1) + 2 + 1, HWD = (TW - 1) + 2 + 1;
        const size_t max_lds = 2 * ((size_t)HH * HWD * lds_pitch(128) + (size_t)TH * TW * lds_pitch(128)) + 8 * 128 * sizeof(float);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_ws_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)max_lds + 4096) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wgrad_ws_multi: cannot raise dynamic LDS to %zu", max_lds);
        }
    }
    hipLaunchKernelGGL(conv_wgrad_ws_multi_kernel, dim3(total), dim3(512), b.lds, st, b.m);
    b.n = 0; b.lds = 0;
    DH_CHECK_LAUNCH("conv_wgrad_ws_multi");
    return 0;
}
// takes the layer into the open batch (a.splitk becomes its in-batch slice count); a full batch is launched first
static int ws_batch_add(WgArgs& a, hipStream_t st) {
    WsBatch& b = g_wsb;
    if (b.n == WS_MAXJ) { const int rc = ws_batch_flush(st); if (rc) return rc; }
    a.splitk = ws_batch_splitk(a);
    if (b.n == 0) b.m.first[0] = 0;
    const int nblocks = (a.Cout / 64) * a.ci_tiles * a.splitk;
    b.m.a[b.n] = a;
    b.m.first[b.n + 1] = b.m.first[b.n] + ((nblocks + 7) & ~7);
    constexpr int HH = (TH - 1) + 2 + 1, HWD = (TW - 1) + 2 + 1;
    const size_t lds = 2 * ((size_t)HH * HWD * lds_pitch(128) + (size_t)TH * TW * lds_pitch(128)) +
                       (a.in_scale ? (size_t)a.in_groups * 128 * sizeof(float) : 0);
    if (lds > b.lds) b.lds = lds;
    ++b.n;
    return 0;
}

static inline bool c32_batch_on() { static const bool off = getenv("DAHITRA_WGRAD_BATCH32") && atoi(getenv("DAHITRA_WGRAD_BATCH32")) == 0; return !off; }
// the second family a batch takes: the 32-channel 3x3 layers (DAHiTra's top-down path: seven launches of ~32 us per step)
static inline bool c32_eligible(const WgArgs& a, int ks, int stride, bool bf16, bool tr) {
    return bf16 && tr && ks == 3 && stride == 1 && a.dil == 1 && a.groups == 1 && !a.phase_mode && !a.dyt_y && !a.direct &&
           co_tile(a.CoutUse) == 32 && a.npix == a.OH * a.OW && a.in_npix == a.H * a.W;
}
static size_t c32_lds(const WgArgs& a) {
    constexpr int HH = (TH - 1) + 2 + 1, HWD = (TW - 1) + 2 + 1;
    size_t lds = (size_t)HH * HWD * lds_pitch(32 * 2) + (size_t)TH * TW * lds_pitch(32 * 2) +
                 (a.in_scale ? (size_t)a.in_groups * 2 * 32 * sizeof(float) : 0);
    const size_t red = (size_t)(32 / 16) * 9 * (32 / 16) * 64 * 16;      // the end-of-kernel wave-group combine (launch_ct)
    return lds < red ? red : lds;
}
static int c32_batch_flush(hipStream_t st) {
    WsBatch& b = g_c32b;
    if (b.n == 0) return 0;
    b.m.njobs = b.n;
    batch_sort(b, 32);
    const int total = b.m.first[b.n];
    auto kern = conv_wgrad_multi_kernel<bf16, 3, 1, 32, true, 1, 32, 1>;
    static size_t attr_lds = 64 * 1024;
    if (b.lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b.lds) != hipSuccess) {
            (void)hipGetLastError();
            DH_FAIL("conv_wgrad_multi: cannot raise dynamic LDS to %zu", b.lds);
        }
        attr_lds = b.lds;
    }
    hipLaunchKernelGGL(kern, dim3(total), dim3(256), b.lds, st, b.m);
    b.n = 0; b.lds = 0;
    DH_CHECK_LAUNCH("conv_wgrad_multi");
    return 0;
}
static int c32_batch_add(WgArgs& a, hipStream_t st) {
    WsBatch& b = g_c32b;
    if (b.n == WS_MAXJ) { const int rc = c32_batch_flush(st); if (rc) return rc; }
    static const int tpw = getenv("DAHITRA_WGRAD_TPW32") ? atoi(getenv("DAHITRA_WGRAD_TPW32")) : 32;
    long sk = ((long)a.N * a.tilesX * a.tilesY) / (tpw > 0 ? tpw : 32);
    if (sk > a.splitk) sk = a.splitk;
    if (sk >= 8) sk &= ~7L;
    a.splitk = (int)(sk < 1 ? 1 : sk);
    if (b.n == 0) b.m.first[0] = 0;
    const int nblocks = dh_cdiv(a.CoutUse, 32) * a.ci_tiles * a.splitk;
    b.m.a[b.n] = a;
    b.m.first[b.n + 1] = b.m.first[b.n] + ((nblocks + 7) & ~7);
    const size_t lds = c32_lds(a);
    if (lds > b.lds) b.lds = lds;
    ++b.n;
    return 0;
}

template <typename T, int KS, int STRIDE, int IT, int DIL, int CT, int CIG = 1>
int launch_ct(const WgArgs& a, bool tr, hipStream_t st) {
    constexpr int HH = (TH - 1) * STRIDE + (KS - 1) * DIL + 1, HWD = (TW - 1) * STRIDE + (KS - 1) * DIL + 1;
    constexpr int LE = Prec<T>::X3 ? 2 : (int)sizeof(T), NPL = Prec<T>::X3 ? 2 : 1;      // LDS element bytes, planes (wg_body)
    size_t lds = NPL * ((size_t)HH * HWD * lds_pitch(IT * CIG * LE) + (size_t)TH * TW * lds_pitch(CT * LE)) +
                 (a.in_scale ? (size_t)a.in_groups * 2 * IT * CIG * sizeof(float) : 0) +
                 (a.dyt_y ? (size_t)a.dyt_groups * 3 * CT * sizeof(float) : 0);
    if (CT < 64) {                                     // the end-of-kernel wave-group combine parks accumulators here
        const size_t red = (size_t)(CT / 16) * KS * KS * (IT / 16) * 64 * 16;
        if (lds < red) lds = red;
    }
    dim3 grid(dh_cdiv(a.CoutUse, CT) * a.ci_tiles, a.splitk, a.phase_mode ? 4 : a.groups);
    auto go = [&](auto kern) -> int {
        static bool attr_done = false;      // once per instantiation (and never inside a graph capture)
        if (lds > 64 * 1024 && !attr_done) {
            attr_done = true;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                (void)hipGetLastError();      // do not leave a sticky error for the next launch check
                DH_FAIL("conv_wgrad: cannot raise dynamic LDS to %zu", lds);
            }
        }
        hipLaunchKernelGGL(kern, grid, dim3(256 * CIG), lds, st, a);
        DH_CHECK_LAUNCH("conv_wgrad");
        return 0;
    };
    if constexpr (KS == 4 && CT == 64 && sizeof(T) == 2) {
        if (a.dyt_y) {
            if (tr) return go(conv_wgrad_kernel<T, KS, STRIDE, IT, true, DIL, CT, CIG, true>);
            return go(conv_wgrad_kernel<T, KS, STRIDE, IT, false, DIL, CT, CIG, true>);
        }
    }
    if (a.dyt_y) DH_FAIL("conv_wgrad: the BatchNorm-backward-on-load form is built for the bf16 stem (4x4, 64 output channels)");
    if constexpr (Prec<T>::X3) return go(conv_wgrad_kernel<T, KS, STRIDE, IT, true, DIL, CT, CIG>);
    else {
        if (tr) return go(conv_wgrad_kernel<T, KS, STRIDE, IT, true, DIL, CT, CIG>);
        return go(conv_wgrad_kernel<T, KS, STRIDE, IT, false, DIL, CT, CIG>);
    }
}
// 3x3 layers with >= 64 input channels and a 64-wide co tile run the 512-thread, 64co x 64ci variant (CIG = 2)
static inline bool wide_ci3x3(int Cin, int CoutUse, int ks) {
    static const bool off = getenv("DAHITRA_WGRAD_CIG1") != nullptr;       // A/B switch for tools/kbench.py
    return !off && ks == 3 && Cin >= 64 && co_tile(CoutUse) == 64;
}
template <typename T, int KS, int STRIDE, int IT, int DIL = 1>
int launch(const WgArgs& a, bool tr, hipStream_t st) {
    if (co_tile(a.CoutUse) == 16) return launch_ct<T, KS, STRIDE, IT, DIL, 16>(a, tr, st);
    if (co_tile(a.CoutUse) == 32) return launch_ct<T, KS, STRIDE, IT, DIL, 32>(a, tr, st);
    if constexpr (KS == 3 && IT == 32 && STRIDE == 1) {      // (stride 2: the 64-channel halo does not fit / is slower)
        if (wide_ci3x3(a.Cin, a.CoutUse, KS)) return launch_ct<T, KS, STRIDE, IT, DIL, 64, 2>(a, tr, st);
    }
    return launch_ct<T, KS, STRIDE, IT, DIL, 64>(a, tr, st);
}

// ---- 1x1 / stride 1, bf16, large layers: a workgroup owns a CT x IT block of dW (256 x 128 and its relatives) -----------------
// dW[co][ci] = sum over pixels of dY[p][co] X[p][ci] streams both tensors once per (co tile, ci tile) PAIR: with the 64 x 64
// slabs of wg_body the 1024 x 256 layers of a ResNet-50 trunk are 64 pairs -- 2.1 GB of operand reads for a 335 MB operand
// pair, 333 us per launch at the 1024 x 1024 bench size, HBM-bound.  Here the four waves of a workgroup are a 2 x 2 grid over a
// block CT x IT = 256 x 128 / 128 x 256 / 256 x 64 / 64 x 256 (32 accumulator tiles per wave at most): 8 pairs for that layer.
// Pixels are flat (p = ((n H) + y) W + x: a 1x1 stride-1 layer has no halo), 64 per stage: the next stage's 16-byte pieces are
// requested into registers before the MFMAs of this one and committed to LDS after them (pitches odd multiples of 32 bytes,
// fragments by ds_read_b64_tr_b16 as in wg_body); same partial-slab layout [splitk][Cout][Cin], same reduce.
struct W1Args {
    const bf16* x;
    const bf16* dy;
    float* part;
    long P;                   // pixels
    int Cin, CinPitch, Cout, CoutUse, splitk, ci_tiles, no_xcd_remap;
    int stride, OW, OHW, W, HW;   // stride 2: output pixel p = (n, oy, ox) reads x at (n, 2 oy, 2 ox) of an H x W image
};
constexpr int W1_PS = 64;     // pixels per stage

template <int CT, int IT>
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(W1Args p) {
    constexpr int DP = lds_pitch(CT * 2), XP = lds_pitch(IT * 2);
    constexpr int DQ = CT * 2 / 16, XQ = IT * 2 / 16;                   // 16-byte pieces per pixel
    constexpr int ND = W1_PS * DQ / 256, NX = W1_PS * XQ / 256;         // pieces per thread and stage
    constexpr int NA = CT / 32, NB = IT / 32;                           // 16-wide sub-tiles per wave (2 x 2 waves)
    static_assert(ND >= 1 && NX >= 1 && NA * NB <= 32, "block shape");
    __shared__ __attribute__((aligned(16))) unsigned char dyt[W1_PS * DP];
    __shared__ __attribute__((aligned(16))) unsigned char xt[W1_PS * XP];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;
    // Workgroups are dispatched x-fastest, round-robin over the 8 XCDs (one L2 each): remapped so that all blocks of one pixel
    // split -- they read the same dY / X stages at about the same time -- run on ONE XCD, whose L2 then serves the re-reads
    int bx = blockIdx.x, kz = blockIdx.y;
    if (gridDim.x > 1 && (gridDim.y & 7) == 0 && !p.no_xcd_remap) {
        const unsigned lin = (unsigned)(kz * gridDim.x + bx), xcd = lin & 7, sq = lin >> 3;
        bx = (int)(sq % gridDim.x);
        kz = (int)((sq / gridDim.x) * 8 + xcd);
    }
    const int cot = bx / p.ci_tiles, cit = bx - cot * p.ci_tiles;
    const int co0 = cot * CT, ci0 = cit * IT;
    const int cow = (wv >> 1) * (CT / 2), ciw = (wv & 1) * (IT / 2);    // this wave's corner inside the block
    f32x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread's pieces of a stage: pixel row and channel piece, fixed for the launch
    const int dq = tid % DQ, dpx = tid / DQ, xq = tid % XQ, xpx = tid / XQ;            // piece i: pixel dpx + i * (256 / DQ)
    const bool dch = co0 + dq * 8 < p.Cout, xch = ci0 + xq * 8 < p.Cin;
    const long nchunk = (p.P + W1_PS - 1) / W1_PS;
    uint4 rd[ND], rx[NX];
    auto fetch = [&](long chunk) {
        const long p0 = chunk * W1_PS;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const long px = p0 + dpx + i * (256 / DQ);
            const bool ok = dch && px < p.P;
            const uint4 v = *reinterpret_cast<const uint4*>(p.dy + (ok ? px * p.Cout + co0 + dq * 8 : 0));
            rd[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const long px = p0 + xpx + i * (256 / XQ);
            const bool ok = xch && px < p.P;
            long xpix = px;
            if (p.stride == 2) {            // (uniform; 32-bit divisions: P < 2^31)
                const unsigned up = (unsigned)px, n = up / (unsigned)p.OHW, rem = up - n * (unsigned)p.OHW;
                const unsigned oy = rem / (unsigned)p.OW, ox = rem - oy * (unsigned)p.OW;
                xpix = (long)n * p.HW + (long)(2 * oy) * p.W + 2 * ox;
            }
            const uint4 v = *reinterpret_cast<const uint4*>(p.x + (ok ? xpix * p.CinPitch + ci0 + xq * 8 : 0));
            rx[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < ND; ++i) *reinterpret_cast<uint4*>(dyt + (dpx + i * (256 / DQ)) * DP + dq * 16) = rd[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<uint4*>(xt + (xpx + i * (256 / XQ)) * XP + xq * 16) = rx[i];
    };
    // fragment of 32 pixels (k) x 16 channels: lane (pl, g) supplies pixels 4 g .. 4 g + 3 and 16 + 4 g .. + 3 (wg_body's form)
    const unsigned char* a_base = dyt + (4 * g + (pl >> 2)) * DP + (cow + (pl & 3) * 4) * 2;
    const unsigned char* b_base = xt + (4 * g + (pl >> 2)) * XP + (ciw + (pl & 3) * 4) * 2;
    auto frag = [&](const unsigned char* base, int pitch, int kk, int ch) {
        F8 f;
        f.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + kk * pitch + ch * 2));
        f.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + (kk + 16) * pitch + ch * 2));
        return f;
    };
    long chunk = kz;
    if (chunk < nchunk) fetch(chunk);
    for (; chunk < nchunk; chunk += p.splitk) {
        commit();
        __syncthreads();
        if (chunk + p.splitk < nchunk) fetch(chunk + p.splitk);
#pragma unroll
        for (int kk = 0; kk < W1_PS; kk += 32) {
            F8 bf[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = frag(b_base, XP, kk, b * 16);
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const F8 af = frag(a_base, DP, kk, a * 16);
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf[b].v, acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // partial slab [kz][Cout][Cin]: lane (pl, g) holds rows co = 4 g + j of column ci = pl of every sub-tile
    float* out = p.part + (size_t)kz * p.Cout * p.Cin;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int ci = ci0 + ciw + b * 16 + pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + cow + a * 16 + 4 * g + j;
                if (co < p.Cout && ci < p.Cin) out[(size_t)co * p.Cin + ci] = acc[a][b][j];
            }
        }
}

// block shape by operand traffic: dY is read once per ci tile, X once per co tile
static inline void w1_pick(long Cout, long Cin, int& ct, int& it) {
    const int cand[4][2] = {{256, 128}, {128, 256}, {256, 64}, {64, 256}};
    double best = 1e30;
    for (auto& c : cand) {
        const double t = (double)Cout * dh_cdiv((int)Cin, c[1]) + (double)Cin * dh_cdiv((int)Cout, c[0]) +
                         0.25 * ((double)dh_cdiv((int)Cout, c[0]) * c[0] - Cout + (double)dh_cdiv((int)Cin, c[1]) * c[1] - Cin);
        if (t < best) { best = t; ct = c[0]; it = c[1]; }
    }
}
static inline bool w1_shape(int Cin, int Cout, int ks) {
    static const bool on = getenv("DAHITRA_WGRAD_1X1_BIG") ? atoi(getenv("DAHITRA_WGRAD_1X1_BIG")) != 0 : true;
    return on && ks == 1 && Cin % 8 == 0 && Cout % 8 == 0 && (long)Cin * Cout >= 128 * 128 && (Cin >= 256 || Cout >= 256);
}
static inline bool w1_eligible(const WgArgs& a, int ks, int stride, bool bf16_, bool tr) {
    const bool geo = stride == 1 ? (a.H == a.OH && a.W == a.OW) : (stride == 2 && a.OH == (a.H + 1) / 2 && a.OW == (a.W + 1) / 2);
    return bf16_ && tr && geo && w1_shape(a.Cin, a.Cout, ks) && a.groups == 1 && !a.in_scale && !a.x_split && !a.direct &&
           !a.dyt_y && !a.phase_mode && a.CoutUse == a.Cout && a.CinPitch % 8 == 0 && a.pad == 0 &&
           a.npix == a.OH * a.OW && a.in_npix == a.H * a.W && (long)a.N * a.OH * a.OW < (1L << 31);
}
// ... and only where its few fat blocks still fill the chip (blocks x pixel splits >= 256 workgroups; the split count is capped by
// 8 pixel tiles per workgroup, so a 256 x 128 layer at 64 x 32 x 32 pixels would be 64 workgroups: that one stays with wg_body)
static inline bool w1_fills(const WgArgs& a) {
    int ct = 256, it = 128;
    w1_pick(a.Cout, a.Cin, ct, it);
    return (long)dh_cdiv(a.Cout, ct) * dh_cdiv(a.Cin, it) * a.splitk >= 256;
}
static int launch_w1(const WgArgs& a, hipStream_t st) {
    int ct = 256, it = 128;
    w1_pick(a.Cout, a.Cin, ct, it);
    W1Args w;
    w.x = reinterpret_cast<const bf16*>(a.x); w.dy = reinterpret_cast<const bf16*>(a.dy); w.part = a.part;
    w.P = (long)a.N * a.OH * a.OW; w.stride = (a.H == a.OH && a.W == a.OW) ? 1 : 2; w.OW = a.OW; w.OHW = a.OH * a.OW; w.W = a.W; w.HW = a.H * a.W;
    w.Cin = a.Cin; w.CinPitch = a.CinPitch; w.Cout = a.Cout; w.CoutUse = a.CoutUse;
    w.splitk = a.splitk; w.ci_tiles = dh_cdiv(a.Cin, it); w.no_xcd_remap = a.no_xcd_remap;
    dim3 grid(dh_cdiv(a.Cout, ct) * w.ci_tiles, a.splitk);
    if (ct == 256 && it == 128) hipLaunchKernelGGL((wgrad1x1_kernel<256, 128>), grid, dim3(256), 0, st, w);
    else if (ct == 128) hipLaunchKernelGGL((wgrad1x1_kernel<128, 256>), grid, dim3(256), 0, st, w);
    else if (it == 64) hipLaunchKernelGGL((wgrad1x1_kernel<256, 64>), grid, dim3(256), 0, st, w);
    else hipLaunchKernelGGL((wgrad1x1_kernel<64, 256>), grid, dim3(256), 0, st, w);
    DH_CHECK_LAUNCH("conv_wgrad 1x1");
    return 0;
}

// 1x1 layers with >= 128 input channels and a 64-wide co tile: the 512-thread form, 64co x 128ci per workgroup (CIG = 2 on the
// 64-wide ci tile) -- the dY slice of a co tile is staged once for 128 input channels instead of twice.  The Bottleneck layers of
// the ResNet-50 trunk (1024 x 256 at 128 x 128 x 8 pixels: 16 x 4 tile pairs re-read a 335 MB operand pair)
static inline bool wide_ci1x1(int Cin, int CoutUse, int ks) {
    static const bool on = getenv("DAHITRA_WGRAD_1X1_CIG2") ? atoi(getenv("DAHITRA_WGRAD_1X1_CIG2")) != 0 : true;
    return on && ks == 1 && Cin >= 128 && Cin % 128 == 0 && co_tile(CoutUse) == 64;
}
template <typename T>
int launch_all(WgArgs& a, int ks, int stride, bool tr, hipStream_t st) {
    const bool wide = a.Cin > 32 && ks == 1 && stride == 1;   // 64-wide ci tiles only where accumulators / LDS fit
    const bool wide2 = wide && wide_ci1x1(a.Cin, a.CoutUse, ks);
    const int it = ks == 4 ? 16 : (wide2 ? 128 : ((wide || (stride == 1 && wide_ci3x3(a.Cin, a.CoutUse, ks))) ? 64 : 32));
    a.ci_tiles = dh_cdiv(a.Cin, it);
    if (w1_eligible(a, ks, stride, std::is_same<T, bf16>::value, tr) && w1_fills(a)) return launch_w1(a, st);
    if (ws_eligible(a, ks, stride, sizeof(T) == 2, tr) && wide_ci3x3(a.Cin, a.CoutUse, ks)) {
        if (g_wsb.on && a.dil == 1 && !a.direct) return ws_batch_add(a, st);
        return a.dil == 2 ? launch_ws<2>(a, st) : launch_ws<1>(a, st);
    }
    if (ks == 3 && stride == 1 && a.dil == 2) return launch<T, 3, 1, 32, 2>(a, tr, st);
    if (g_wsb.on && c32_batch_on() && c32_eligible(a, ks, stride, sizeof(T) == 2, tr)) return c32_batch_add(a, st);
    if (ks == 3 && stride == 1) return launch<T, 3, 1, 32>(a, tr, st);
    if (ks == 3 && stride == 2) return launch<T, 3, 2, 32>(a, tr, st);
    if (ks == 1 && stride == 1 && wide2) return launch_ct<T, 1, 1, 64, 1, 64, 2>(a, tr, st);
    if (ks == 1 && stride == 1) return wide ? launch<T, 1, 1, 64>(a, tr, st) : launch<T, 1, 1, 32>(a, tr, st);
    if (ks == 1 && stride == 2) return launch<T, 1, 2, 32>(a, tr, st);
    if (ks == 4 && stride == 1) return launch<T, 4, 1, 16>(a, tr, st);     // space-to-depth stem (12 real channels)
    if (ks == 2 && stride == 1 && a.phase_mode) return launch<T, 2, 1, 32>(a, tr, st);
    DH_FAIL("conv_wgrad: unsupported kernel %d stride %d", ks, stride);
}

}  // namespace

// dh_set_f32_mma_mode(1 / 2) (conv_mfma.hip): fp32 launches whose channel counts are multiples of 4 (16-byte aligned half
// pieces: the class head's one-piece-per-pixel dlogits have 4) take the split-bf16 three-product form (wg_body<f32x3>); the
// others keep the exact fp32 MFMA
extern "C" int dh_get_f32_mma_mode(void);
static inline bool wgrad_x3(int dtype, const WgArgs& a) {
    static const bool skip = getenv("DAHITRA_X3_NO_WGRAD") != nullptr;      // experiment switch
    return !skip && dtype == DH_DTYPE_F32 && dh_get_f32_mma_mode() != 0 && a.Cin % 4 == 0 && a.Cout % 4 == 0 && a.CinPitch % 4 == 0;
}

// split-K factor: one resident round of workgroups, never more slabs than pixel tiles
extern "C" int dh_conv2d_wgrad_splitk(int N, int OH, int OW, int Cin, int Cout, int ks, int groups) {
    const long tiles = (long)(N / (groups > 0 ? groups : 1)) * dh_cdiv(OW, TW) * dh_cdiv(OH, TH);
    // NOTE: stride is not known here; the 64-wide ci tile is only used at stride 1, where this
    // estimate is exact; at stride 2 it under-estimates the slab count (harmless: more workgroups)
    const bool big = wide_ci3x3(Cin, Cout, ks);       // 512-thread workgroups, one per CU
    const int it = ks == 4 ? 16 : (wide_ci1x1(Cin, Cout, ks) ? 128 : (((Cin > 32 && ks == 1) || big) ? 64 : 32));
    long slabs = (long)dh_cdiv(Cout, co_tile(Cout)) * dh_cdiv(Cin, it) * (groups > 0 ? groups : 1);
    bool w1 = false;
    if (groups <= 1 && w1_shape(Cin, Cout, ks)) {      // (dtype / stride unknown here: a launch that takes another kernel just gets fewer, fatter slabs)
        int ct = 256, it1 = 128;
        w1_pick(Cout, Cin, ct, it1);
        slabs = (long)dh_cdiv(Cout, ct) * dh_cdiv(Cin, it1);
        w1 = true;
    }
    // workgroups in flight: the 3x3 / 4x4 kernels hold two workgroups per CU (168+ registers per lane), so 512 fill the
    // chip in ONE round -- a second round only doubles the partial-slab traffic and the per-workgroup prologue / slab
    // write (measured: layer3 115.6 -> 107.6 us, classifier 75.7 -> 65.1 us); the light 1x1 kernels fit four per CU
    static const long w1_target = getenv("DAHITRA_W1_TARGET") ? atol(getenv("DAHITRA_W1_TARGET")) : 512;
    static const long w1_target1 = getenv("DAHITRA_W1_TARGET1") ? atol(getenv("DAHITRA_W1_TARGET1")) : 512;
    const long target = w1 ? (slabs == 1 ? w1_target1 : w1_target) : (ks == 1 ? 1024 : (big ? 256 : 512));
    long sk = (target + slabs - 1) / slabs;      // ... however small Cout x Cin is ...
    if (sk > tiles / 8) sk = tiles / 8;        // ... but at least 8 pixel tiles per workgroup (slab write amortised)
    if (sk > 1024) sk = 1024;
    return (int)(sk < 1 ? 1 : sk);
}

// whether a plain bf16 1x1 / stride-1 weight gradient of this shape (one group, no BatchNorm on load) runs as wgrad1x1_kernel
// blocks -- for tests and tools: the rule of launch_all, w1_eligible and w1_fills
extern "C" int dh_conv2d_wgrad_1x1_blocks(int N, int H, int W, int Cin, int Cout) {
    if (!w1_shape(Cin, Cout, 1)) return 0;
    int ct = 256, it = 128;
    w1_pick(Cout, Cin, ct, it);
    const long blocks = (long)dh_cdiv(Cout, ct) * dh_cdiv(Cin, it);
    return blocks * dh_conv2d_wgrad_splitk(N, H, W, Cin, Cout, 1, 1) >= 256 ? (int)blocks : 0;
}

extern "C" long dh_conv2d_wgrad_workspace_size(int N, int OH, int OW, int Cin, int Cout, int ks, int groups) {
    return (long)groups * dh_conv2d_wgrad_splitk(N, OH, OW, Cin, Cout, ks, groups) * ks * ks * Cout * Cin * 4;
}

// x: [N,H,W,Cin], dy: [N,OH,OW,Cout]; groups == 1: dw_oihw (+)= gradient in torch OIHW layout;
// groups == N : dw_oihw is [N][Cout][Cin] (ks must be 1) -- one gradient per image.
// defer != 0: only the partial slabs are written; returns the split-K factor (0: the result went straight into dW)
static int conv2d_wgrad_impl(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N,
                             int H, int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad,
                             int groups, int npix_valid, int use_tr, int Cout_real, int cin_pitch, int dilation,
                             void* workspace, void* stream, int defer, int* splitk_out, const float* in_scale = nullptr,
                             const float* in_shift = nullptr, int in_groups = 1, const void* dyt_y = nullptr,
                             const float* dytoef = nullptr, int dyt_groups = 1, long x_split = 0) {
    DH_REQUIRE(groups == 1 || (groups == N && ks == 1), "conv2d_wgrad: groups must be 1 or N (with ks=1)");
    WgArgs a;
    a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(workspace);
    a.x_split = x_split;
    DH_REQUIRE(dilation == 1 || (dilation == 2 && ks == 3 && stride == 1), "conv2d_wgrad: dilation %d unsupported here", dilation);
    a.dil = dilation;
    a.phase_mode = 0;
    static const int no_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    a.no_xcd_remap = no_remap;
    a.dyt_y = dyt_y; a.dytoef = dytoef; a.dyt_groups = dyt_groups;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_groups = in_groups > 0 ? in_groups : 1;
    if (in_scale) DH_REQUIRE(in_shift && groups == 1 && N % a.in_groups == 0 && (Cin * (dtype == DH_DTYPE_BF16 ? 2 : 4)) % 16 == 0,
                             "conv2d_wgrad: BatchNorm-on-load needs in_shift, one weight group, N %% in_groups == 0, 16-byte channel pieces");
    a.CinPitch = cin_pitch > 0 ? cin_pitch : Cin;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.pad = pad;
    a.tilesX = dh_cdiv(OW, TW); a.tilesY = dh_cdiv(OH, TH);
    a.CoutUse = Cout_real > 0 ? Cout_real : Cout;
    a.groups = groups; a.splitk = dh_conv2d_wgrad_splitk(N, OH, OW, Cin, Cout, ks, groups);
    // 1x1 with a single K slab: [tap = 1][Cout][Cin] is exactly dW's [Cout][Cin] (also per image: [N][Cout][Cin])
    a.direct = (ks == 1 && a.splitk == 1 && a.CoutUse == Cout) ? (accumulate ? 2 : 1) : 0;
    if (a.direct) a.part = dw_oihw;
    a.npix = npix_valid > 0 ? npix_valid : OH * OW;
    a.in_npix = npix_valid > 0 ? npix_valid : H * W;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (x_split) {
        a.ci_tiles = dh_cdiv(Cin, 64);
        DH_REQUIRE(dtype == DH_DTYPE_BF16 && ws_eligible(a, ks, stride, true, use_tr != 0) && wide_ci3x3(Cin, a.CoutUse, ks),
                   "conv2d_wgrad: a split input is served by the wave-specialised 3x3 kernel only (Cin=%d Cout=%d ks=%d)", Cin, Cout, ks);
    }
    const bool batching = g_wsb.on;
    if (!defer) g_wsb.on = false;           // this call reduces right after its launch: never recorded into a batch
    int rc = dtype == DH_DTYPE_BF16 ? launch_all<bf16>(a, ks, stride, use_tr != 0, st)
             : wgrad_x3(dtype, a)   ? launch_all<f32x3>(a, ks, stride, true, st)
                                    : launch_all<float>(a, ks, stride, false, st);
    g_wsb.on = batching;
    if (rc) return rc;
    if (splitk_out) *splitk_out = a.direct ? 0 : a.splitk;
    if (a.direct || defer) return 0;
    const int taps = ks * ks;
    const int oreal = Cout_real > 0 ? Cout_real : Cout;     // dy may carry zero-padded channels
    const long n = (long)oreal * Cin * taps;
    hipLaunchKernelGGL(wgrad_reduce_oihw_kernel, dim3(dh_cdiv(n, wgrad_reduce_epb(Cin)), groups), dim3(256), 0, st, a.part, a.splitk,
                       taps, Cout, oreal, Cin, dw_oihw, accumulate);
    DH_CHECK_LAUNCH("wgrad_reduce");
    return 0;
}

// Batched weight gradients (one backward pass = one launch for the wave-specialised 3x3 layers).  Between _begin and _end,
// dh_conv2d_wgrad_partial / dh_conv2d_wgrad_bn_in calls (deferred form: splitk_out given) for eligible layers only RECORD
// their launch; dh_wgrad_batch_launch issues everything recorded so far (their x / dy / workspace must be alive and unchanged
// until then), dh_wgrad_batch_pending tells how many layers wait.  Thread-local state; a 17th layer launches the first 16.
extern "C" int dh_wgrad_batch_begin() { g_wsb.on = true; g_wsb.n = 0; g_wsb.lds = 0; g_c32b.n = 0; g_c32b.lds = 0; return 0; }
extern "C" int dh_wgrad_batch_pending() { return g_wsb.n + g_c32b.n; }
// closes the batch WITHOUT launching what it recorded (an aborted pass: the recorded operands may be gone)
extern "C" int dh_wgrad_batch_abort() { g_wsb.on = false; g_wsb.n = 0; g_wsb.lds = 0; g_c32b.n = 0; g_c32b.lds = 0; return 0; }
extern "C" int dh_wgrad_batch_launch(void* stream) {
    const int rc = ws_batch_flush(reinterpret_cast<hipStream_t>(stream));
    return rc ? rc : c32_batch_flush(reinterpret_cast<hipStream_t>(stream));
}
extern "C" int dh_wgrad_batch_end(void* stream) {
    const int rc = dh_wgrad_batch_launch(stream);
    g_wsb.on = false;
    return rc;
}

extern "C" int dh_conv2d_wgrad(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N,
                               int H, int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad,
                               int groups, int npix_valid, int use_tr, int Cout_real, int cin_pitch, int dilation,
                               void* workspace, void* stream) {
    return conv2d_wgrad_impl(dtype, x, dy, dw_oihw, accumulate, N, H, W, Cin, OH, OW, Cout, ks, stride, pad, groups,
                             npix_valid, use_tr, Cout_real, cin_pitch, dilation, workspace, stream, 0, nullptr);
}

// The same launch without its reduce: the partial slabs stay in `workspace` (which must then outlive the batched
// reduce) and *splitk_out receives the number of slabs; 0 means the single-slab 1x1 case already wrote dW.
extern "C" int dh_conv2d_wgrad_partial(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N,
                                       int H, int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad,
                                       int groups, int npix_valid, int use_tr, int Cout_real, int cin_pitch,
                                       int dilation, void* workspace, int* splitk_out, void* stream) {
    DH_REQUIRE(groups == 1 && splitk_out, "conv2d_wgrad_partial: one group only");
    return conv2d_wgrad_impl(dtype, x, dy, dw_oihw, accumulate, N, H, W, Cin, OH, OW, Cout, ks, stride, pad, groups,
                             npix_valid, use_tr, Cout_real, cin_pitch, dilation, workspace, stream, 1, splitk_out);
}
// dh_conv2d_wgrad_partial for a 3x3 / stride 1 / pad 1 layer whose input is cat([A, B], channel) of two [N][H][W][Cin / 2]
// tensors, A at x and B at x + x_split_bytes, never materialised (see dh_conv3x3_split_fwd; models/networks.py:1344).
bool dh_wgrad_split_supported(int N, int H, int W, int Cin, int Cout) {
    WgArgs a;
    memset(&a, 0, sizeof(a));
    a.dil = 1; a.in_groups = 1; a.dyt_groups = 1;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.Cout = Cout; a.CoutUse = Cout; a.pad = 1; a.groups = 1;
    a.CinPitch = Cin / 2; a.x_split = 16; a.npix = H * W; a.in_npix = H * W;
    return ws_eligible(a, 3, 1, true, true) && wide_ci3x3(Cin, Cout, 3);
}
extern "C" int dh_conv2d_wgrad_split(const void* x, long x_split_bytes, const void* dy, float* dw_oihw, int accumulate, int N, int H,
                                     int W, int Cin, int Cout, void* workspace, int* splitk_out, void* stream) {
    DH_REQUIRE(x_split_bytes > 0 && x_split_bytes % 16 == 0 && Cin % 128 == 0 && splitk_out, "conv2d_wgrad_split: bad arguments (Cin=%d)", Cin);
    return conv2d_wgrad_impl(DH_DTYPE_BF16, x, dy, dw_oihw, accumulate, N, H, W, Cin, H, W, Cout, 3, 1, 1, 1, 0, 1, 0, Cin / 2, 1,
                             workspace, stream, 1, splitk_out, nullptr, nullptr, 1, nullptr, nullptr, 1, x_split_bytes);
}
// dh_conv2d_wgrad / dh_conv2d_wgrad_partial (splitk_out != NULL: deferred) with BatchNorm-apply + ReLU on the load of x:
// x is the PRE-normalisation output of the previous convolution, the gradient is taken against
// relu(x * in_scale[g][ci] + in_shift[g][ci]) (g = image / (N / in_groups)); padding stays zero.
extern "C" int dh_conv2d_wgrad_bn_in(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N,
                                     int H, int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad,
                                     int use_tr, int Cout_real, int dilation, const float* in_scale, const float* in_shift,
                                     int in_groups, void* workspace, int* splitk_out, void* stream) {
    DH_REQUIRE(in_scale && in_shift, "conv2d_wgrad_bn_in: scale / shift missing");
    return conv2d_wgrad_impl(dtype, x, dy, dw_oihw, accumulate, N, H, W, Cin, OH, OW, Cout, ks, stride, pad, 1, 0, use_tr,
                             Cout_real, 0, dilation, workspace, stream, splitk_out ? 1 : 0, splitk_out, in_scale, in_shift,
                             in_groups);
}
// The stem's weight gradient with BatchNorm backward applied on load (see WgArgs::dyt_y): xs16 [N][OH][OW][16] bf16 (the
// space-to-depth by-product of dh_stem7_fwd), d / y [N][OH][OW][64] bf16, coef [groups][3][64] from dh_stem_pool_bn_bwd;
// dw2 [64][16][4][4] fp32 is ASSIGNED (dh_stem_unpack_grad folds it into the 7x7 gradient).
extern "C" int dh_stem_wgrad_bn(const void* xs16, const void* d, const void* y, const float* coef, int groups, int N, int OH,
                                int OW, float* dw2, int use_tr, void* workspace, void* stream) {
    DH_REQUIRE(xs16 && d && y && coef && groups >= 1 && N % groups == 0, "stem_wgrad_bn: bad arguments (N=%d groups=%d)", N, groups);
    return conv2d_wgrad_impl(DH_DTYPE_BF16, xs16, d, dw2, 0, N, OH, OW, 16, OH, OW, 64, 4, 1, 2, 1, 0, use_tr, 0, 16, 1, workspace,
                             stream, 0, nullptr, nullptr, nullptr, 1, y, coef, groups);
}
extern "C" int dh_wgrad_reduce_job_size(void) { return (int)sizeof(WgReduceJob); }
// outputs served by one workgroup of dh_wgrad_reduce_multi for a layer with `Cin` input channels (job.nblocks =
// ceil(O * I * taps / this))
extern "C" int dh_wgrad_reduce_outputs_per_block(int Cin) { return wgrad_reduce_epb(Cin); }
extern "C" int dh_wgrad_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream) {
    if (njobs <= 0) return 0;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const WgReduceJob*>(jobs_dev), njobs);
    DH_CHECK_LAUNCH("wgrad_reduce_multi");
    return 0;
}

// ---- 2x2 phase form of conv3x3(nearest-upsample-x2(x)) with 32 output channels: weight gradient per output parity ----
static int phase_splitk(int N, int H, int W, int Cin) {
    const long tiles = (long)N * dh_cdiv(W, TW) * dh_cdiv(H, TH);
    const long slabs = (long)dh_cdiv(Cin, 32);                 // one 32-wide co tile x ci tiles, per phase
    long sk = (128 + slabs - 1) / slabs;                       // 4 phases x slabs x sk ~ 512 workgroups
    if (sk > tiles / 8) sk = tiles / 8;
    return (int)(sk < 1 ? 1 : sk);
}
extern "C" long dh_conv2d_wgrad_phase_workspace_size(int N, int H, int W, int Cin) {
    return (long)4 * phase_splitk(N, H, W, Cin) * 4 * 32 * Cin * 4;
}
extern "C" int dh_conv2d_wgrad_phase(int dtype, const void* x, const void* dy, int N, int H, int W, int Cin, int use_tr,
                                     void* workspace, int* splitk_out, void* stream) {
    DH_REQUIRE(splitk_out && workspace, "conv2d_wgrad_phase: workspace / splitk_out missing");
    DH_REQUIRE((Cin * (dtype == DH_DTYPE_BF16 ? 2 : 4)) % 16 == 0, "conv2d_wgrad_phase: Cin=%d not 16-byte aligned", Cin);
    WgArgs a;
    a.x = x; a.dy = dy; a.part = reinterpret_cast<float*>(workspace);
    a.dil = 1; a.phase_mode = 1;
    a.no_xcd_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    a.dyt_y = nullptr; a.dytoef = nullptr; a.dyt_groups = 1;
    a.in_scale = nullptr; a.in_shift = nullptr; a.in_groups = 1;
    a.CinPitch = Cin;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.Cout = 32; a.pad = 1;
    a.tilesX = dh_cdiv(W, TW); a.tilesY = dh_cdiv(H, TH);
    a.CoutUse = 32;
    a.x_split = 0;
    a.groups = 1; a.splitk = phase_splitk(N, H, W, Cin);
    a.direct = 0;
    a.npix = H * W; a.in_npix = H * W;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int rc = dtype == DH_DTYPE_BF16 ? launch_all<bf16>(a, 2, 1, use_tr != 0, st)
                   : wgrad_x3(dtype, a)   ? launch_all<f32x3>(a, 2, 1, true, st) : launch_all<float>(a, 2, 1, false, st);
    if (rc) return rc;
    *splitk_out = a.splitk;
    return 0;
}

namespace {
// dw[co][ci][kh][kw] (+)= sum over the phase taps (a, t) that read source row kh: kh 0 <- (0,0) (1,0); 1 <- (0,1) (1,0);
// 2 <- (0,1) (1,1); columns alike.  dwab: [4 phases][32][Cin][2][2]
__global__ void phase_wgrad_combine_kernel(const float* __restrict__ dwab, float* __restrict__ dw, int Cin, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (co, ci)
    if (i >= 32 * Cin) return;
    const size_t pstride = (size_t)32 * Cin * 4;
    const float* src = dwab + (size_t)i * 4;
    float v[4][4];                                               // [phase][t * 2 + u]
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const float4 q = *reinterpret_cast<const float4*>(src + ph * pstride);
        v[ph][0] = q.x; v[ph][1] = q.y; v[ph][2] = q.z; v[ph][3] = q.w;
    }
    // row sets: kh -> {(a, t)}
    const int ra[3][2] = {{0, 1}, {0, 1}, {0, 1}}, rt[3][2] = {{0, 0}, {1, 0}, {1, 1}};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) s += v[ra[kh][m] * 2 + ra[kw][n]][rt[kh][m] * 2 + rt[kw][n]];
            float* d = dw + (size_t)i * 9 + kh * 3 + kw;
            if (accumulate) *d += s; else *d = s;
        }
}

// W_ab[t][u] = sum of the 3x3 taps (kh, kw) whose source row / column under nearest-x2 upsampling is the same:
// a = 0: t = 0 <- kh {0}, t = 1 <- kh {1, 2};  a = 1: t = 0 <- kh {0, 1}, t = 1 <- kh {2}
template <typename T>
__global__ void pack_phase_weights_kernel(const float* __restrict__ w, const float* __restrict__ bias, int Cin,
                                          T* __restrict__ fwd, T* __restrict__ dgrad, float* __restrict__ bias4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (co, ci)
    if (i < 128 && bias4) bias4[i] = bias ? bias[i & 31] : 0.f;
    if (i >= 32 * Cin) return;
    const int co = i / Cin, ci = i % Cin;
    float k[3][3];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j / 3][j % 3] = w[(size_t)i * 9 + j];
    const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};      // [a][t] -> kh range [lo, hi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float s = 0.f;
                    for (int kh = lo[a][t]; kh <= hi[a][t]; ++kh)
                        for (int kw = lo[b][u]; kw <= hi[b][u]; ++kw) s += k[kh][kw];
                    const int ph = a * 2 + b;
                    // forward: [tap = t * 2 + u][ph * 32 + co][ci]
                    if (fwd) stf(fwd + ((size_t)(t * 2 + u) * 128 + ph * 32 + co) * Cin + ci, s);
                    // data gradient (a 2x2 pad-1 conv over the phase-gathered gradient): tap (kh', kw') = (1 - t, 1 - u),
                    // [tap][ci][ph * 32 + co]
                    if (dgrad) stf(dgrad + ((size_t)((1 - t) * 2 + (1 - u)) * Cin + ci) * 128 + ph * 32 + co, s);
                }
}
}  // namespace

namespace {
// Data gradient of a 3x3 / stride 2 / pad 1 convolution [Co][Ci] as a phase convolution over dY (ConvArgs::phase_mode 1):
// output parity (a, b) of dX reads dY rows i - 1 + a + t (t = 0, 1) with
//   a = 0: t = 1 <- kh 1 (t = 0: no tap);   a = 1: t = 0 <- kh 2, t = 1 <- kh 0        (columns alike)
// out: [tap = t * 2 + u][(a * 2 + b) * Ci + ci][co] T, zeros where a phase has no tap
template <typename T>
__global__ void pack_s2_dgrad_phase_kernel(const float* __restrict__ w, int Co, int Ci, T* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;        // (co, ci)
    if (i >= Co * Ci) return;
    const int co = i / Ci, ci = i % Ci;
    float k[3][3];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j / 3][j % 3] = w[(size_t)i * 9 + j];
    const int kmap[2][2] = {{-1, 1}, {2, 0}};        // [a][t] -> kh (or none)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int kh = kmap[a][t], kw = kmap[b][u];
                    const float v = (kh >= 0 && kw >= 0) ? k[kh][kw] : 0.f;
                    stf(out + ((size_t)(t * 2 + u) * 4 * Ci + (a * 2 + b) * Ci + ci) * Co + co, v);
                }
}
}  // namespace
extern "C" int dh_pack_s2_dgrad_phase_weights(int dtype, const float* w_oihw, int Co, int Ci, void* out, void* stream) {
    DH_REQUIRE((Ci == 32 || Ci == 64) && Co % 16 == 0, "pack_s2_dgrad_phase_weights: Ci=%d (32 or 64) Co=%d", Ci, Co);
    const dim3 grid(dh_cdiv(Co * Ci, 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DH_DTYPE_BF16) hipLaunchKernelGGL(pack_s2_dgrad_phase_kernel<bf16>, grid, dim3(256), 0, st, w_oihw, Co, Ci, (bf16*)out);
    else hipLaunchKernelGGL(pack_s2_dgrad_phase_kernel<float>, grid, dim3(256), 0, st, w_oihw, Co, Ci, (float*)out);
    DH_CHECK_LAUNCH("pack_s2_dgrad_phase_weights");
    return 0;
}
extern "C" int dh_phase_wgrad_combine(const float* dwab, float* dw_oihw, int Cin, int accumulate, void* stream) {
    hipLaunchKernelGGL(phase_wgrad_combine_kernel, dim3(dh_cdiv(32 * Cin, 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       dwab, dw_oihw, Cin, accumulate);
    DH_CHECK_LAUNCH("phase_wgrad_combine");
    return 0;
}
extern "C" int dh_pack_phase_weights(int dtype, const float* w_oihw, const float* bias, int Cin, void* fwd, void* dgrad,
                                     float* bias4, void* stream) {
    DH_REQUIRE(Cin % 16 == 0, "pack_phase_weights: Cin=%d", Cin);
    const dim3 grid(dh_cdiv(32 * Cin, 256));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DH_DTYPE_BF16)
        hipLaunchKernelGGL(pack_phase_weights_kernel<bf16>, grid, dim3(256), 0, st, w_oihw, bias, Cin, (bf16*)fwd, (bf16*)dgrad, bias4);
    else
        hipLaunchKernelGGL(pack_phase_weights_kernel<float>, grid, dim3(256), 0, st, w_oihw, bias, Cin, (float*)fwd, (float*)dgrad, bias4);
    DH_CHECK_LAUNCH("pack_phase_weights");
    return 0;
}
