// The ResNet stem nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False) (reference models/resnet.py:150) as ONE kernel on the
// NCHW fp32 images the boundary hands over: no space-to-depth pass, no 32-channel padded image, the conv itself HBM-bound
// (50 MB of images in, 134 MB of bf16 activations out at the bench size) instead of a 4x4 conv over a zero-padded tensor.
//
//   GEMM view per 8x16-pixel output tile: D[64 couts][128 pixels] = W[64][K] * P[K][128], K = (c, kh, kw') with kw' = 0..7
//   (kw' = 7 is a zero weight) = 21 rows of 8 -> padded to 24 rows = 6 MFMA k-steps of 32.  A k-step covers 4 (c, kh) rows;
//   lane group g of a wave owns row 4*step + g, its 8 k-values are 8 CONSECUTIVE input columns 2*ox + 0..7 of input row
//   2*oy + kh -- one 16-byte window of the bf16 image patch in LDS (4-byte aligned: four ds_read_b32).
//   The 64 x 192 weights live in registers (24 fragments per lane) for the life of a persistent workgroup.
//
// By-product for the weight gradient (which keeps the 4x4 space-to-depth form, conv_wgrad.hip): the tile's own 8x16 block of
// the space-to-depth image [N][H/2][W/2][16] (12 real channels (ry*2+rx)*3+c), written from the patch already in LDS.
#include "common.h"

namespace {

union Frag {
    uint4 u;
    s16x8 h;
};

constexpr int TH = 8, TW = 16;               // output tile
constexpr int PR = 2 * TH + 5;               // 21 patch rows
constexpr int PC = 2 * TW + 5;               // 37 patch columns
constexpr int PP = 40;                       // patch row pitch (bf16 elements)
constexpr int PN = 3 * PR * PC;              // 2331 patch elements
constexpr int NPRE = (PN + 255) / 256;       // 10 per thread
constexpr int TPB = 144;                     // transposed output tile: 128 bytes per pixel + 16 (bank spread)

struct Stem7Args {
    const float* xa;        // images [0, B): NCHW fp32, 3 channels
    const float* xb;        // images [B, N)
    int B, N, H, W, OH, OW, tilesX, tilesY, ntiles;
    const float* w;         // OIHW fp32 [64][3][7][7]
    const float* oscale;    // optional per-cout scale folded into the weights (eval-mode BatchNorm)
    const float* bias;      // optional
    int relu;
    bf16* y;                // [N][OH][OW][64]
    float* stats;           // optional [2][64][gridDim.x]: sum, sum of squares per WORKGROUP (of the stored bf16 values)
    int groups;             // a workgroup stays inside one group's tiles (the first N/groups images are group 0, ...)
    bf16* xs;               // optional [N][OH][OW][16] space-to-depth image
};

template <bool EPI>      // EPI: bias + ReLU epilogue (eval form); otherwise the raw convolution (+ statistics)
__global__ __launch_bounds__(256, 2) void stem7_fwd_kernel(Stem7Args p) {
    __shared__ __attribute__((aligned(16))) unsigned short patch[3 * PR * PP + 8];     // + a dump slot for the idle lanes
    __shared__ __attribute__((aligned(16))) unsigned char tr[4][32 * TPB];
    __shared__ float red[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, pl = lane & 15, g = lane >> 4;

    // the pitch padding (columns 37..39) is read as the k-value behind kw' = 7 of the last pixel: its weight is zero, the
    // padding must merely be finite (never written by a commit, zeroed once; the first commit's barrier orders it)
    for (int i = tid; i < 3 * PR * (PP - PC); i += 256) patch[(i / (PP - PC)) * PP + PC + i % (PP - PC)] = 0;
    // ---- weights: 4 cout sub-tiles x 6 k-steps, straight from the fp32 OIHW tensor ----
    Frag A[4][6];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int co = s * 16 + pl, r = 4 * k + g;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r < 21) {
                const float* q = p.w + co * 147 + r * 7;
                const float sc = p.oscale ? p.oscale[co] : 1.f;
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = q[j] * sc;
            }
            A[s][k].u = pack16<bf16>(v);
        }
    int roff[6];             // patch offset of this lane group's (c, kh) row per k-step (rows >= 21 carry zero weights)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int r = min(4 * k + g, 20);
        roff[k] = ((r / 7) * PR + (r % 7)) * PP;
    }

    // ---- patch staging: tile-independent per-thread element coordinates ----
    int lrc[NPRE];           // LDS offset | row << 12 | col << 20 | channel << 28
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
        const int idx = tid + i * 256;
        const int c = idx / (PR * PC), rem = idx - c * (PR * PC), r = rem / PC, col = rem - r * PC;
        lrc[i] = idx < PN ? ((c * PR + r) * PP + col) | (r << 12) | (col << 20) | (c << 28) : (3 * PR * PP) | (0xff << 12);   // row 255: never inside
    }
    float pre[NPRE];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
        const float* img = n < p.B ? p.xa + (size_t)n * 3 * p.H * p.W : p.xb + (size_t)(n - p.B) * 3 * p.H * p.W;
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
        const int base = iy0 * p.W + ix0;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int r = (lrc[i] >> 12) & 0xff, col = (lrc[i] >> 20) & 0xff, c = (lrc[i] >> 28) & 3;
            const bool ok = (unsigned)(iy0 + r) < (unsigned)p.H && (unsigned)(ix0 + col) < (unsigned)p.W;
            const float v = img[ok ? base + (c * p.H + r) * p.W + col : 0];
            pre[i] = ok ? v : 0.f;
        }
    };

    // tiles of this workgroup: a strided walk through its group's tiles (statistics are kept per workgroup)
    const int wpg = gridDim.x / p.groups, tpg = p.ntiles / p.groups;
    const int grp = blockIdx.x / wpg, tend = (grp + 1) * tpg;
    int tile = grp * tpg + (int)blockIdx.x - grp * wpg;
    float ssum[8], ssq[8];       // of the stored (bf16) values: channels (lane & 7) * 8 .. + 8 of the pixels this lane stores
#pragma unroll
    for (int j = 0; j < 8; ++j) ssum[j] = ssq[j] = 0.f;
    float bs[4][4];
    if constexpr (EPI) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[s][j] = p.bias ? p.bias[s * 16 + g * 4 + j] : 0.f;
    }
    if (tile < tend) fetch(tile);
    for (; tile < tend; tile += wpg) {
#pragma unroll
        for (int i = 0; i < NPRE; ++i) patch[lrc[i] & 0xfff] = f2bf(pre[i]);
        __syncthreads();
        if (tile + wpg < tend) fetch(tile + wpg);

        int t = tile;
        const int tx = t % p.tilesX; t /= p.tilesX;
        const int ty = t % p.tilesY;
        const int n = t / p.tilesY;
        const int oy0 = ty * TH, ox0 = tx * TW;

        if (p.xs) {          // this tile's block of the space-to-depth image: thread = (pixel, half of its 16 channels)
            const int px = tid >> 1, half = tid & 1, py = px >> 4, pxx = px & 15;
            if (oy0 + py < p.OH && ox0 + pxx < p.OW) {
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = half * 8 + j;       // (ry*2 + rx)*3 + c
                    const int c = ch % 3, r = ch / 3;
                    e[j] = ch < 12 ? patch[(c * PR + 3 + 2 * py + (r >> 1)) * PP + 3 + 2 * pxx + (r & 1)] : (unsigned short)0;
                }
                uint4 v;
                v.x = e[0] | ((unsigned)e[1] << 16); v.y = e[2] | ((unsigned)e[3] << 16);
                v.z = e[4] | ((unsigned)e[5] << 16); v.w = e[6] | ((unsigned)e[7] << 16);
                *reinterpret_cast<uint4*>(p.xs + (((size_t)n * p.OH + oy0 + py) * p.OW + ox0 + pxx) * 16 + half * 8) = v;
            }
        }

        f32x4 acc[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s][0] = acc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t* pw = reinterpret_cast<const uint32_t*>(patch);
#pragma unroll
        for (int k = 0; k < 6; ++k)
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint32_t* q = pw + ((roff[k] + 2 * (2 * wv + rr) * PP + 2 * pl) >> 1);
                Frag b;
                b.u = make_uint4(q[0], q[1], q[2], q[3]);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[s][rr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][k].h, b.h, acc[s][rr], 0, 0, 0);
            }

        // ---- epilogue: bias / ReLU, statistics, transpose through LDS, 16-byte stores ----
        unsigned char* mytr = tr[wv];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float v[4] = {acc[s][rr][0], acc[s][rr][1], acc[s][rr][2], acc[s][rr][3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (EPI) {
                        v[j] += bs[s][j];
                        if (p.relu) v[j] = fmaxf(v[j], 0.f);
                    }
                }
                uint2 o;
                o.x = f2bf2(v[0], v[1]);
                o.y = f2bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(mytr + (rr * 16 + pl) * TPB + s * 32 + g * 8) = o;
            }
        }
        __syncthreads();          // also: every wave is past its patch reads, the next commit may overwrite the patch
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q, px = i >> 3, piece = i & 7;
            const int oy = oy0 + 2 * wv + (px >> 4), ox = ox0 + (px & 15);
            if (oy < p.OH && ox < p.OW) {
                const uint4 v = *reinterpret_cast<const uint4*>(mytr + px * TPB + piece * 16);
                *reinterpret_cast<uint4*>(p.y + (((size_t)n * p.OH + oy) * p.OW + ox) * 64 + piece * 8) = v;
                if constexpr (!EPI) {
                    float f[8];
                    unpack16(v, f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        ssum[j] += f[j];
                        ssq[j] += f[j] * f[j];
                    }
                }
            }
        }
    }
    if constexpr (!EPI) {
        if (p.stats) {       // one reduction per workgroup: the 8 lanes that share a piece index, then the four waves through LDS
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = ssum[j], b = ssq[j];
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    a += __shfl_xor(a, o, 64);
                    b += __shfl_xor(b, o, 64);
                }
                if (lane < 8) {
                    red[wv][0][lane * 8 + j] = a;
                    red[wv][1][lane * 8 + j] = b;
                }
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, c = tid & 63;
                p.stats[((size_t)which * 64 + c) * gridDim.x + blockIdx.x] =
                    red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
            }
        }
    }
}

}  // namespace

// workgroups of the launch = statistic slots per channel (a multiple of `groups`, every group gets the same number)
extern "C" int dh_stem7_fwd_num_slots(int N, int H, int W, int groups) {
    if (groups < 1 || N % groups) return 0;
    const long tpg = (long)(N / groups) * dh_cdiv(H / 2, TH) * dh_cdiv(W / 2, TW);
    long wpg = 512 / groups;          // two persistent workgroups per CU
    if (wpg < 1) wpg = 1;
    if (wpg > tpg) wpg = tpg;
    return (int)(wpg * groups);
}

// C ABI: see include/dahitra_hip.h
extern "C" int dh_stem7_fwd(const float* xa, const float* xb, int B, int N, int H, int W, const float* w_oihw,
                            const float* out_scale, const float* bias, int relu, void* y, float* stats, int groups,
                            void* xs16, void* stream) {
    DH_REQUIRE(xa && w_oihw && y && N >= 1 && B >= 1 && B <= N && (B == N || xb), "stem7_fwd: bad image arguments (B=%d N=%d)", B, N);
    DH_REQUIRE(H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && (long)3 * H * W < (1L << 30), "stem7_fwd: H, W must be even (got %d x %d)", H, W);
    Stem7Args a;
    a.xa = xa; a.xb = xb; a.B = B; a.N = N; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2;
    a.tilesX = dh_cdiv(a.OW, TW); a.tilesY = dh_cdiv(a.OH, TH); a.ntiles = N * a.tilesX * a.tilesY;
    a.w = w_oihw; a.oscale = out_scale; a.bias = bias; a.relu = relu;
    a.y = (bf16*)y; a.stats = stats; a.xs = (bf16*)xs16;
    if (!stats) groups = 1;
    const int grid = dh_stem7_fwd_num_slots(N, H, W, groups);
    DH_REQUIRE(grid > 0, "stem7_fwd: %d images do not split into %d statistic groups", N, groups);
    DH_REQUIRE(!(stats && (out_scale || bias || relu)), "stem7_fwd: statistics are those of the raw convolution (no scale / bias / ReLU)");
    a.groups = groups;
    if (bias || relu) hipLaunchKernelGGL(stem7_fwd_kernel<true>, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(stem7_fwd_kernel<false>, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    DH_CHECK_LAUNCH("stem7_fwd");
    return 0;
}
