// C ABI of the MFMA direct convolution (kernel: conv_mfma_impl.h).  The bf16 and fp32 instantiations are compiled in
// separate translation units (conv_mfma_bf16.hip / conv_mfma_f32.hip): each is minutes of device code generation.
#include "conv_mfma_impl.h"

int dh_conv_launch_bf16(const ConvArgs& a, int ks, int stride, hipStream_t st);
int dh_conv_launch_f32(const ConvArgs& a, int ks, int stride, hipStream_t st);
int dh_conv_launch_x3(const ConvArgs& a, int ks, int stride, hipStream_t st);      // conv_mfma_x3.hip
int dh_conv_launch_x6(const ConvArgs& a, int ks, int stride, hipStream_t st);      // conv_mfma_x6.hip
int dh_conv_launch_h3(const ConvArgs& a, int ks, int stride, hipStream_t st);      // conv_mfma_h3.hip

// How the matrix products of fp32 (DH_DTYPE_F32) launches are computed -- per host thread, read at launch time (so a recorded
// graph keeps the form it was captured with):
//   0  exact fp32 on v_mfma_f32_16x16x4_f32 (default);
//   1  split-bf16, two planes (common.h f32x3): three v_mfma_f32_16x16x32_bf16 products per operand pair on (hi, lo) bf16
//      splits formed while staging, unit roundoff 2^-17 -- tensors, accumulation and everything that is not a matrix product
//      stay fp32;
//   2  split-bf16, three planes (f32x6): six products, unit roundoff 2^-23 (convolutions / linears only; weight-gradient
//      launches issued in this mode run form 1).
// Launches whose input-channel count is not a multiple of 32 (the 16-channel space-to-depth stem) or whose planes do not fit
// the LDS (form 2 at stride 2 with 3x3 taps) keep the exact form.
// (DAHITRA_F32_MMA=bf16x3 makes mode 1 the initial value: the whole fp32 test suite can then run on the split form)
// Split-bf16 launches take 16-row tiles for every 3x3 stride-1 layer with enough tiles, not only from 128 input channels on:
// they run one workgroup per CU whatever the tile (the planes), and a 16-row tile stages -- and splits -- the weight tile once
// per 256 pixels instead of once per 128 (bf16x3 step 3043 / 3020 -> 3100 pairs/s same-box; from 64 channels on only: 3064).
// DAHITRA_X_RW4=<min Cin> moves the threshold, 0 switches it off.
static int x_rw4_cin_min() {
    const char* e = getenv("DAHITRA_X_RW4");
    return e ? atoi(e) : 32;
}
// modes: 0 exact fp32 MFMA; 1 split-bf16, two planes / three products (2^-17: the backward of compute_dtype "bf16x3"); 2 split-bf16,
// three planes / six products (2^-23: the forward's fallback); 3 split-fp16, two planes / three products (~2^-21, fp16's range:
// the forward's default).  DAHITRA_F32_MMA sets the thread default (bf16x3 -> 1, bf16x6 -> 2); engines set the mode per pass.
static int f32_mma_env_default() {
    const char* e = getenv("DAHITRA_F32_MMA");
    return e && !strcmp(e, "bf16x3") ? 1 : (e && !strcmp(e, "bf16x6") ? 2 : 0);
}
static thread_local int g_f32_mma_mode = f32_mma_env_default();
extern "C" int dh_set_f32_mma_mode(int mode) {
    DH_REQUIRE(mode >= 0 && mode <= 3, "set_f32_mma_mode: mode %d (0 = exact fp32 MFMA, 1 / 2 = split-bf16 three- / six-product form, 3 = split-fp16 three-product form)", mode);
    g_f32_mma_mode = mode;
    return 0;
}
extern "C" int dh_get_f32_mma_mode(void) { return g_f32_mma_mode; }
static inline int pick_rw_mode(int dtype, int N, int OH, int OW, int Cin, int ks, int stride) {
    static const int cmin = x_rw4_cin_min();
    if (cmin && dtype == DH_DTYPE_F32 && g_f32_mma_mode != 0 && Cin % 32 == 0 && Cin >= cmin && ks == 3 && stride == 1 && OH >= 16 &&
        (long)N * dh_cdiv(OH, 16) * dh_cdiv(OW, TW) >= 256)
        return 4;
    return pick_rw(N, OH, OW, Cin, ks, stride);
}
// fp32 launch -> its kernel family under the current mode
static int launch_f32_family(const ConvArgs& a, int ks, int stride, hipStream_t st) {
    if (g_f32_mma_mode != 0 && a.Cin % 32 == 0) {
        const int rc = g_f32_mma_mode == 3 ? dh_conv_launch_h3(a, ks, stride, st)
                     : g_f32_mma_mode == 2 ? dh_conv_launch_x6(a, ks, stride, st) : dh_conv_launch_x3(a, ks, stride, st);
        if (rc != DH_CONV_NO_FIT) return rc;
    }
    return dh_conv_launch_f32(a, ks, stride, st);
}
// K-deep GEMM form of the 1x1 / stride-1 convolutions with >= 64 input channels (conv1x1_gemm.hip)
bool dh_conv1x1_gemm_eligible(const ConvArgs& a, int ks, int stride, int dtype);
int dh_conv1x1_gemm_launch(const ConvArgs& a, hipStream_t st);
// 3x3 stride-1 convolutions with the weights resident in registers, persistent workgroups (conv_wreg.hip)
bool dh_conv_wreg_eligible(const ConvArgs& a, int ks, int stride, int dtype);
int dh_conv_wreg_launch(const ConvArgs& a, hipStream_t st);
bool dh_conv_wreg_up4_eligible(const ConvArgs& a);
int dh_conv_wreg_up4_launch(const ConvArgs& a, hipStream_t st);

// C ABI: see include/dahitra_hip.h
extern "C" int dh_conv2d_fwd(int dtype, const void* x, const void* w_packed, void* y, const float* bias,
                             const void* residual, float* stats_partial, int N, int H, int W, int Cin,
                             int OH, int OW, int Cout, int CoutPad, int ks, int stride, int pad, int act,
                             int npix_valid, long w_image_stride, void* y_preact, int dilation, const void* gate_out,
                             const void* gate_y, const float* gate_mean, const float* gate_invstd, int gate_groups,
                             const float* in_scale, const float* in_shift, int in_groups, int phase_mode, const void* w_frag,
                             void* stream) {
    const int esz = dtype == DH_DTYPE_BF16 ? 2 : 4;
    DH_REQUIRE(dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16, "conv2d_fwd: bad dtype %d", dtype);
    DH_REQUIRE((Cin * esz) % 64 == 0, "conv2d_fwd: Cin=%d must be a multiple of %d", Cin, 64 / esz);
    DH_REQUIRE(CoutPad % 16 == 0 && CoutPad >= Cout, "conv2d_fwd: CoutPad=%d invalid for Cout=%d", CoutPad, Cout);
    DH_REQUIRE(N > 0 && OH > 0 && OW > 0, "conv2d_fwd: empty output");
    ConvArgs a;
    a.x = x; a.w = w_packed; a.y = y; a.bias = bias; a.res = residual; a.stats = stats_partial; a.y2 = y_preact;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout; a.CoutPad = CoutPad;
    a.pad = pad; a.act = act; a.npix = npix_valid > 0 ? npix_valid : OH * OW;
    a.in_npix = npix_valid > 0 ? npix_valid : H * W; a.w_nstride = w_image_stride;
    DH_REQUIRE(dilation == 1 || (dilation == 2 && ks == 3 && stride == 1), "conv2d_fwd: dilation %d unsupported here", dilation);
    a.dil = dilation;
    a.gate_out = gate_out; a.gate_y = gate_y; a.gate_mean = gate_mean; a.gate_invstd = gate_invstd;
    a.gate_groups = gate_groups > 0 ? gate_groups : 1;
    if (gate_y) {
        DH_REQUIRE(stats_partial && gate_mean && gate_invstd && Cout % 4 == 0 && act == DH_ACT_NONE && N % a.gate_groups == 0,
                   "conv2d_fwd: BN-backward gating needs stats_partial, mean/invstd, Cout %% 4 == 0, no activation");
    }
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_groups = in_groups > 0 ? in_groups : 1;
    if (in_scale) DH_REQUIRE(in_shift && N % a.in_groups == 0 && w_image_stride == 0,
                             "conv2d_fwd: BatchNorm-on-load needs in_shift and N %% in_groups == 0");
    a.phase_mode = phase_mode;
    a.y_nchw = nullptr;
    a.up4_partial = nullptr;
    a.x_split = a.y_split = 0;
    a.up4_a = a.up4_b = nullptr;
    a.w_frag = w_frag;
    DH_REQUIRE(!w_frag || (ks == 3 && w_image_stride == 0 && !phase_mode && dtype == DH_DTYPE_BF16 && Cin % 32 == 0 && CoutPad % 16 == 0),
               "conv2d_fwd: fragment-order weights exist for the bf16 3x3 layers only");
    static const int no_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    a.no_xcd_remap = no_remap;
    if (phase_mode) {
        DH_REQUIRE(ks == 2 && stride == 1 && pad == 1 && dilation == 1 && (!residual || phase_mode == 1) && !stats_partial && !y_preact && !gate_y &&
                   !in_scale && w_image_stride == 0 && npix_valid == 0 && H == OH && W == OW && act != DH_ACT_GELU,
                   "conv2d_fwd: phase mode is a plain 2x2 pad-1 convolution on equal input / output grids");
        DH_REQUIRE(phase_mode == 1 ? ((Cout == 128 || Cout == 256) && CoutPad == Cout) : (phase_mode == 2 && Cin == 128 && (Cout % 64 == 0 || Cout == 32)),
                   "conv2d_fwd: phase mode %d with Cin=%d Cout=%d", phase_mode, Cin, Cout);
    }
    a.rw = pick_rw_mode(dtype, N, OH, OW, Cin, ks, stride);
    a.tilesX = dh_cdiv(OW, TW); a.tilesY = dh_cdiv(OH, 4 * a.rw);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dh_conv1x1_gemm_eligible(a, ks, stride, dtype)) return dh_conv1x1_gemm_launch(a, st);
    if (dh_conv_wreg_eligible(a, ks, stride, dtype)) return dh_conv_wreg_launch(a, st);
    if (dtype == DH_DTYPE_BF16) return dh_conv_launch_bf16(a, ks, stride, st);
    return launch_f32_family(a, ks, stride, st);
}

// 3x3 / stride 1 / pad 1 convolution (bf16) over a channel concatenation that is never materialised, see ConvArgs::x_split /
// y_split: conv_layer2_0 of the hierarchical model reads torch.cat([a_128, b_128], 1) (models/networks.py:1344) -- the two
// temporal streams' stem outputs, which here are the two halves of one [2 N]-image tensor -- and its data gradient writes
// the two halves of that tensor's gradient.  Without this the concatenation and its gradient cost four channel-copy passes
// of 134 MB each per step at batch 32.  x_split_bytes / y_split_bytes: byte distance from the first to the second tensor
// (0: plain tensor on that side).  No bias, residual or activation; stats_partial as dh_conv2d_fwd.  Served by the
// register-resident-weights kernel only: dh_conv3x3_split_supported says whether a shape is (callers fall back to
// dh_copy_channels + dh_conv2d_fwd otherwise).
static bool split_conv_args(ConvArgs& a, const void* x, long x_split, const void* w_packed, const void* w_frag, void* y, long y_split,
                            float* stats, int N, int H, int W, int Cin, int Cout) {
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w_packed; a.w_frag = w_frag; a.y = y; a.stats = stats;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.Cout = Cout; a.CoutPad = Cout;
    a.pad = 1; a.act = DH_ACT_NONE; a.npix = H * W; a.in_npix = H * W; a.dil = 1; a.gate_groups = 1; a.in_groups = 1;
    a.x_split = x_split; a.y_split = y_split;
    a.rw = pick_rw(N, H, W, Cin, 3, 1);
    a.tilesX = dh_cdiv(W, TW); a.tilesY = dh_cdiv(H, 4 * a.rw);
    return dh_conv_wreg_eligible(a, 3, 1, DH_DTYPE_BF16);
}
bool dh_wgrad_split_supported(int N, int H, int W, int Cin, int Cout);      // conv_wgrad.hip
extern "C" int dh_conv3x3_split_supported(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 128 || Cout % 64) return 0;
    ConvArgs a;
    static unsigned char dummy[16];
    // forward (split input), data gradient (split output, channel counts exchanged; both need the fragment-order weights)
    if (!split_conv_args(a, dummy, 16, dummy, dummy, dummy, 0, nullptr, N, H, W, Cin, Cout)) return 0;
    if (!split_conv_args(a, dummy, 0, dummy, dummy, dummy, 16, nullptr, N, H, W, Cout, Cin)) return 0;
    return dh_wgrad_split_supported(N, H, W, Cin, Cout) ? 1 : 0;
}
extern "C" int dh_conv3x3_split_fwd(const void* x, long x_split_bytes, const void* w_packed, const void* w_frag, void* y,
                                    long y_split_bytes, float* stats_partial, int N, int H, int W, int Cin, int Cout, void* stream) {
    DH_REQUIRE(x && w_packed && y && N > 0 && (x_split_bytes || y_split_bytes), "conv3x3_split_fwd: bad arguments");
    DH_REQUIRE((!x_split_bytes || Cin % 128 == 0) && (!y_split_bytes || Cout % 128 == 0) && x_split_bytes % 16 == 0 && y_split_bytes % 16 == 0,
               "conv3x3_split_fwd: a split side needs a multiple of 128 channels (Cin=%d Cout=%d) and 16-byte aligned tensors", Cin, Cout);
    ConvArgs a;
    if (!split_conv_args(a, x, x_split_bytes, w_packed, w_frag, y, y_split_bytes, stats_partial, N, H, W, Cin, Cout))
        DH_FAIL("conv3x3_split_fwd: %d x %dx%d, %d -> %d channels is outside the register-resident-weights kernel "
                "(dh_conv3x3_split_supported)", N, H, W, Cin, Cout);
    return dh_conv_wreg_launch(a, reinterpret_cast<hipStream_t>(stream));
}

// The class head (3x3, pad 1, <= 16 classes) with fp32 NCHW logits written by the convolution itself: see ConvArgs::y_nchw.
extern "C" int dh_conv3x3_head_fwd(int dtype, const void* x, const void* w_packed, const float* bias, int N, int H, int W, int Cin,
                                   int Cout, const float* in_scale, const float* in_shift, int in_groups,
                                   float* logits_nchw, void* stream) {
    const int esz = dtype == DH_DTYPE_BF16 ? 2 : 4;
    DH_REQUIRE(dtype == DH_DTYPE_F32 || dtype == DH_DTYPE_BF16, "conv3x3_head_fwd: bad dtype %d", dtype);
    DH_REQUIRE((Cin * esz) % 64 == 0 && Cout >= 1 && Cout <= 16 && logits_nchw && N > 0 && H > 0 && W > 0,
               "conv3x3_head_fwd: Cin=%d Cout=%d", Cin, Cout);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.w = w_packed; a.bias = bias;
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.OH = H; a.OW = W; a.Cout = Cout; a.CoutPad = 16;
    a.pad = 1; a.act = DH_ACT_NONE; a.npix = H * W; a.in_npix = H * W; a.dil = 1; a.gate_groups = 1;
    a.in_scale = in_scale; a.in_shift = in_shift; a.in_groups = in_groups > 0 ? in_groups : 1;
    if (in_scale) DH_REQUIRE(in_shift && N % a.in_groups == 0, "conv3x3_head_fwd: BatchNorm-on-load needs in_shift and N %% in_groups == 0");
    static const int no_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    a.no_xcd_remap = no_remap;
    a.y_nchw = logits_nchw;
    a.rw = pick_rw_mode(dtype, N, H, W, Cin, 3, 1);
    a.tilesX = dh_cdiv(W, TW); a.tilesY = dh_cdiv(H, 4 * a.rw);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == DH_DTYPE_BF16) return dh_conv_launch_bf16(a, 3, 1, st);
    return launch_f32_family(a, 3, 1, st);
}

// Data gradient of a 3x3 / pad-1 convolution whose INPUT is a bilinear x4 upsampled map (models/networks.py:387-389:
// classifier.0 behind nn.Upsample(4, 'bilinear')), taken straight to the COARSE grid: dy [N][H][W][K] (K = the conv's output
// channels, a multiple of the 64-byte chunk), w_packed = the data-gradient pack [9][32][K]; the kernel reduces every 8x16 tile
// of the fine-grid gradient to the 4 x 6 coarse pixels it touches and writes partial [N * (H/8) * (W/16)][4][6][32] fp32.
// The fine-grid gradient (32 channels x H x W) is never stored.  dh_absdiff_up4_combine finishes (sum of <= 4 tiles, sign).
extern "C" long dh_conv3x3_dgrad_up4_partial_floats(int N, int H, int W) { return (long)N * (H / 8) * (W / 16) * 4 * 6 * 32; }
extern "C" int dh_conv3x3_dgrad_up4(int dtype, const void* dy, const void* w_packed, int N, int H, int W, int K, float* partial,
                                    void* stream) {
    DH_REQUIRE(dtype == DH_DTYPE_BF16, "conv3x3_dgrad_up4: bf16 only (the fp32 mode keeps the two-kernel path)");
    DH_REQUIRE(K % 32 == 0 && H % 8 == 0 && W % 16 == 0 && partial && N > 0, "conv3x3_dgrad_up4: K=%d H=%d W=%d", K, H, W);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = dy; a.w = w_packed;
    a.N = N; a.H = H; a.W = W; a.Cin = K; a.OH = H; a.OW = W; a.Cout = 32; a.CoutPad = 32;
    a.pad = 1; a.act = DH_ACT_NONE; a.npix = H * W; a.in_npix = H * W; a.dil = 1; a.gate_groups = 1; a.in_groups = 1;
    static const int no_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    a.no_xcd_remap = no_remap;
    a.up4_partial = partial;
    a.rw = 2;
    a.tilesX = W / TW; a.tilesY = H / 8;
    return dh_conv_launch_bf16(a, 3, 1, reinterpret_cast<hipStream_t>(stream));
}

// classifier.0 on the bilinear-x4 upsampled |A - B| map WITHOUT that map (ConvArgs::up4_a; models/networks.py:383-389,
// models/help_funcs.py:9): a, b [N][H / 4][W / 4][32] bf16 (the two streams' decoder outputs), y [N][H][W][32] bf16 =
// act(conv3x3(upsample4(|a - b|)) + bias); stats_partial as dh_conv2d_fwd with dh_conv2d_fwd_num_tiles(N, H, W, 32, 3, 1) rows.
// The interpolation is dh_absdiff_upsample4_fwd's (same terms, same order, rounded to bf16 as that kernel's output is), so the
// result equals dh_conv2d_fwd on its output bit for bit.
extern "C" int dh_conv3x3_up4_fwd(const void* a, const void* b, const void* w_packed, const float* bias, int act, void* y,
                                  float* stats_partial, int N, int H, int W, void* stream) {
    DH_REQUIRE(a && b && w_packed && y && N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "conv3x3_up4_fwd: N=%d H=%d W=%d", N, H, W);
    DH_REQUIRE(act == DH_ACT_NONE || act == DH_ACT_RELU, "conv3x3_up4_fwd: activation %d", act);
    ConvArgs c;
    memset(&c, 0, sizeof(c));
    c.w = w_packed; c.y = y; c.bias = bias; c.stats = stats_partial;
    c.N = N; c.H = H; c.W = W; c.Cin = 32; c.OH = H; c.OW = W; c.Cout = 32; c.CoutPad = 32;
    c.pad = 1; c.act = act; c.npix = H * W; c.in_npix = H * W; c.dil = 1; c.gate_groups = 1; c.in_groups = 1;
    static const int no_remap = getenv("DAHITRA_NO_XCD_REMAP") ? 1 : 0;
    c.no_xcd_remap = no_remap;
    c.up4_a = a; c.up4_b = b;
    c.rw = 2;
    c.tilesX = dh_cdiv(W, TW); c.tilesY = dh_cdiv(H, 8);
    // the persistent register-resident-weights stream with the interpolation in LDS (csrc/conv_wreg.hip) where it serves the shape;
    // else (ragged tiles, ReLU, few tiles; DAHITRA_UP4_TAP=1) the tap kernel with the interpolation on its load path
    if (dh_conv_wreg_up4_eligible(c)) return dh_conv_wreg_up4_launch(c, reinterpret_cast<hipStream_t>(stream));
    return dh_conv_launch_bf16(c, 3, 1, reinterpret_cast<hipStream_t>(stream));
}

// number of workgroup tiles along the pixel dimension (= rows of the stats_partial buffer)
extern "C" int dh_conv2d_fwd_num_tiles(int dtype, int N, int OH, int OW, int Cin, int ks, int stride) {
    // (the fp32 split forms may pick other tiles than a bf16 launch of the same shape: the rule of the launch that will WRITE the
    // buffer -- dtype, and for fp32 the thread's current MMA mode -- sizes it)
    static const int cmin = x_rw4_cin_min();
    if (dtype == DH_DTYPE_F32 && cmin && g_f32_mma_mode != 0) return N * dh_cdiv(OW, TW) * dh_cdiv(OH, 4 * pick_rw_mode(DH_DTYPE_F32, N, OH, OW, Cin, ks, stride));
    return N * dh_cdiv(OW, TW) * dh_cdiv(OH, 4 * pick_rw(N, OH, OW, Cin, ks, stride));
}
