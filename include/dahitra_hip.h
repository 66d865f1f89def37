/* dahitra_hip.h -- C ABI of libdahitra_hip.so, the MI355X (gfx950) kernel library behind
 * dahitra_amd's drop-in replacement of the DAHiTra change-detection hot path.
 *
 * The reference (nka77/DAHiTra) owns no native code: its hot path issues stock torch operators.
 * Each entry point below therefore cites the reference call site(s) whose torch operator it
 * replaces (paths relative to the reference root).  A maintainer binds these with ctypes exactly as
 * dahitra_amd/_lib.py does.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; dh_last_error() gives the message
 *   - all pointers are DEVICE pointers (hipMalloc / torch CUDA tensors) unless stated otherwise;
 *     buffers are caller-allocated, no ownership is transferred, nothing is allocated inside
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); calls are async
 *   - `dtype`: DH_F32 (parity mode, exact-fp32 MFMA) or DH_BF16 (throughput mode, bf16 MFMA with
 *     fp32 accumulation) selects the ACTIVATION type "T"; parameters, statistics, optimizer state
 *     and weight gradients are always fp32
 *   - activations are NHWC: [N][H][W][C], C contiguous; "rows x C" tensors are the same thing
 *   - `*_workspace_size` twins return the bytes of scratch the call needs (contents undefined)
 */
#ifndef DAHITRA_HIP_H
#define DAHITRA_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define DH_F32 0
#define DH_BF16 1
#define DH_ACT_NONE 0
#define DH_ACT_RELU 1
#define DH_ACT_GELU 2

int dh_abi_version(void);
const char* dh_last_error(void);
void dh_set_error(const char* msg);

/* ---- convolution / linear on the matrix cores ------------------------------------------------
 * replaces F.conv2d / nn.Linear at: models/resnet.py:24-32,150 (trunk), models/networks.py:215
 * (conv_pred), models/help_funcs.py:7-15 (TwoLayerConv2d), :52-63 (FeedForward), :86-88,111
 * (to_q/k/v/out), models/networks.py:1178-1185,1196-1199,1235-1243 (squeeze / decode / top-down
 * convs); with transposed+flipped weights it is also their input gradient (autograd
 * convolution_backward), and with w_image_stride != 0 the per-image attention products
 * (help_funcs.py:92,109) in re-associated form.
 * y[n,oy,ox,co] = act( sum_{kh,kw,ci} x[n, oy*stride-pad+kh, ox*stride-pad+kw, ci] * w[kh*ks+kw][co][ci]
 *                      + bias[co] + residual[n,oy,ox,co] )
 * w_packed: [ks*ks][CoutPad][Cin] T (dh_pack_weight).  Cin*sizeof(T) must be a multiple of 64.
 * y_preact (optional): receives the value before `act`.  stats_partial (optional):
 * [2][CoutPad][dh_conv2d_fwd_num_tiles(dtype,N,OH,OW,Cin,ks,stride)] fp32 per-tile (sum, sum of squares) of y for BatchNorm
 * (channel-major: the per-channel combine reads contiguous runs).
 * npix_valid > 0: treat each image as a row list with that many valid rows (H*W >= npix_valid). */
int dh_conv2d_fwd(int dtype, const void* x, const void* w_packed, void* y, const float* bias,
                  const void* residual, float* stats_partial, int N, int H, int W, int Cin, int OH, int OW,
                  int Cout, int CoutPad, int ks, int stride, int pad, int act, int npix_valid,
                  long w_image_stride, void* y_preact, int dilation, const void* gate_out, const void* gate_y,
                  const float* gate_mean, const float* gate_invstd, int gate_groups, const float* in_scale,
                  const float* in_shift, int in_groups, int phase_mode, const void* w_frag, void* stream);
/* w_frag (optional; bf16 3x3 only): the SAME weights in fragment order [CoutPad / 16][Cin / 32][9][64][8] (what
 * dh_pack_weights_multi writes for jobs with dtype | 0x200): lane l of a wavefront finds the 8 reduction channels
 * 32 c + 8 (l >> 4) .. of output channel 16 r + (l & 15) at [r][c][tap][l], i.e. one MFMA weight fragment is 1 KiB
 * contiguous.  Read only where the register-resident-weights kernel runs (dh_conv_wreg_mode); NULL = it loads w_packed. */
/* phase_mode (0 = off; ks = 2, pad = 1 only): conv3x3(nearest-upsample-x2(x)) -- models/networks.py:251-256, upsamplex2 +
 * conv_pred -- as four 2x2 convolutions on x, one per output parity, weights from dh_pack_phase_weights (2.25x fewer
 * FLOPs, the upsampled tensor is never written).  1: forward, logical Cout = 4 * 32, y is the [N][2 OH][2 OW][32] output
 * (depth-to-space store), bias tiled 4 x.  2: data gradient, x is the [N][2 H][2 W][32] gradient (space-to-depth gather),
 * logical Cin = 4 * 32, y = [N][H][W][Cout]. */
/* in_scale / in_shift ([in_groups][Cin] fp32, NULL = off; 3x3 stride-1 convolutions): BatchNorm-apply + ReLU on LOAD.  x is
 * then the PRE-normalisation output of the previous convolution and the kernel consumes relu(x * in_scale[g][ci] +
 * in_shift[g][ci]), g = n / (N / in_groups); zero padding applies to that post-activation tensor.  Replaces the separate
 * F.batch_norm + ReLU pass between two convolutions (models/resnet.py:60-65, models/help_funcs.py:11-13). */
/* gate_* (all NULL / 0 = off): BatchNorm-backward gating of a data-gradient launch.  gate_y / gate_out are the
 * pre-normalisation input and the post-ReLU output (NULL: no ReLU) of the BN layer whose output gradient this launch
 * produces ([N][OH][OW][Cout] like y), gate_mean / gate_invstd its batch statistics [gate_groups][Cout].  The
 * epilogue then stores g = dout * (gate_out > 0) and writes stats_partial[tile] = (sum g, sum g * xhat), which
 * dh_bn_bwd_from_partials consumes (replaces the reduction pass of dh_bn_bwd; torch autograd's native_batch_norm
 * backward + threshold_backward, called from models/resnet.py:58-73 via loss.backward()). */
int dh_conv2d_fwd_num_tiles(int dtype, int N, int OH, int OW, int Cin, int ks, int stride);   /* dtype: of the launch that fills the buffer */
/* bf16 3x3 / stride-1 / pad-1 convolutions with 64 / 128 / 256 input channels and Cout % 64 == 0 on whole 8x16 tiles run,
 * inside dh_conv2d_fwd, on the register-resident-weights kernel (csrc/conv_wreg.hip: persistent workgroups, the weights of
 * a wavefront's output channels stay in its registers, only the input halo is staged) -- same products in the same order as
 * the tap-oriented kernel, i.e. bit-identical outputs.  mode -1 (default): on the shapes where it is the faster kernel; 0: never
 * (everything through the tap-oriented kernel); 1: wherever it can run.  Returns the previous mode.  Same reference call sites as dh_conv2d_fwd
 * (models/resnet.py:24-73: BasicBlock conv1 / conv2 and their input gradients). */
int dh_conv_wreg_mode(int mode);
/* How the matrix products of DH_F32 launches are computed (dh_conv2d_fwd, dh_conv3x3_head_fwd, dh_conv2d_wgrad* and the
 * kernels that call them), per host thread, read when a launch is issued:
 *   0 (default)  exact fp32: v_mfma_f32_16x16x4_f32;
 *   1            split-bf16, two planes: while an operand tile is staged every fp32 element x becomes hi = bf16(x),
 *                lo = bf16(x - hi) and a product is ah*bh + al*bh + ah*bl on v_mfma_f32_16x16x32_bf16 with fp32 accumulation
 *                (unit roundoff 2^-17 against fp32's 2^-24; 3/16 of the exact form's matrix-core cycles);
 *   2            split-bf16, three planes (x = p0 + p1 + p2, 8 mantissa bits each): the six products a_i*b_j, i + j < 3
 *                (unit roundoff 2^-23; 6/16 of the cycles).  Convolution / linear launches only: a weight-gradient launch
 *                issued in mode 2 or 3 runs form 1.
 *   3            split-fp16, two planes (hi = fp16(x), lo = fp16(x - hi), 11 mantissa bits each): three products on
 *                v_mfma_f32_16x16x32_f16, ~2^-21 at the cycles of form 1 -- for operands inside fp16's RANGE: |x| < 65504, and
 *                |w| < 255 for the weights, which are staged times 2^8 (exact; undone on the accumulators) so that weights of
 *                size 1e-2 keep a normal lo plane.  The forward's form (activations are O(1)); not for gradients.
 * Tensors, accumulators, statistics, everything that is not a matrix product, launches with Cin % 32 != 0 and launches
 * whose staging planes would not fit the LDS are unchanged, so DH_F32 buffers, packs and workspaces are interchangeable
 * between the modes.  The reference computes these products in fp32 (models/networks.py:358-392,
 * models/help_funcs.py:66-114, loss.backward()): compute_dtype="bf16x3" (form 3 in the forward, form 1 in the backward)
 * is the parity mode's fast form. */
int dh_set_f32_mma_mode(int mode);
int dh_get_f32_mma_mode(void);
/* The backward of |a - b| -> nn.Upsample(4, 'bilinear') -> conv3x3 (models/networks.py:383-389; autograd's
 * upsample_bilinear2d_backward + convolution_backward input gradient) WITHOUT the fine-grid gradient tensor: the data-gradient
 * launch of the 3x3 convolution (dy [N][H][W][K] bf16, w_packed = its data-gradient pack [9][32][K]) reduces each 8x16 tile
 * to the 4 x 6 coarse pixels it interpolates from and writes fp32 partials (dh_conv3x3_dgrad_up4_partial_floats of them);
 * dh_absdiff_up4_combine sums the <= 4 tiles of every coarse pixel of the [N][H/4][W/4][32] maps a, b and applies
 * sign(a - b): da, db.  bf16 mode; H % 8 == 0, W % 16 == 0. */
long dh_conv3x3_dgrad_up4_partial_floats(int N, int H, int W);
int dh_conv3x3_dgrad_up4(int dtype, const void* dy, const void* w_packed, int N, int H, int W, int K, float* partial, void* stream);
int dh_absdiff_up4_combine(const float* partial, const void* a, const void* b, void* da, void* db, int N, int H, int W, void* stream);
/* The class head -- classifier[-1]: nn.Conv2d(32, n_class, 3, padding 1) (models/networks.py:201-204, 1121-1129) -- with the
 * fp32 NCHW logits [N][Cout][H][W] (the reference's output layout) written by the convolution itself: no NHWC logits tensor,
 * no layout pass.  w_packed: dh_pack_weight with OPad = 16; in_scale / in_shift (optional): BatchNorm-apply + ReLU on load. */
int dh_conv3x3_head_fwd(int dtype, const void* x, const void* w_packed, const float* bias, int N, int H, int W, int Cin,
                        int Cout, const float* in_scale, const float* in_shift, int in_groups, float* logits_nchw, void* stream);

/* weight gradient (autograd convolution_backward / mm for nn.Linear): groups == 1 writes the
 * torch OIHW layout [Cout_real][Cin][ks][ks]; groups == N (ks == 1) one [Cout][Cin] per image. */
int dh_conv2d_wgrad(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N, int H,
                    int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad, int groups,
                    int npix_valid, int use_tr, int Cout_real, int cin_pitch, int dilation, void* workspace, void* stream);
/* > 0: a plain bf16 1x1 / stride-1 weight gradient of this shape runs in the block form (wgrad1x1_kernel: a 256 x 128 / 128 x 256 /
 * 256 x 64 / 64 x 256 block of dW per workgroup over flat pixels, all blocks of a pixel split on one XCD); the value = blocks per
 * split.  0: the 64 x 64-slab kernel (too few pixels to fill the chip with fat blocks, or a small layer). */
int dh_conv2d_wgrad_1x1_blocks(int N, int H, int W, int Cin, int Cout);
long dh_conv2d_wgrad_workspace_size(int N, int OH, int OW, int Cin, int Cout, int ks, int groups);
/* The weight gradient with its split-K reduce deferred: only the partial slabs are written into `workspace` (which
 * must stay alive until the batched reduce) and *splitk_out receives their count (0: the single-slab 1x1 case wrote
 * dW directly).  dh_wgrad_reduce_multi then sums every deferred layer of a backward pass in one launch.  jobs_dev:
 * njobs records {const float* part; float* dw; int splitk, taps, Oslab, O, I, accumulate, first_block, nblocks;}
 * (dh_wgrad_reduce_job_size() bytes each) sorted by first_block, nblocks = ceil(O*I*taps / dh_wgrad_reduce_outputs_per_block(I)), Oslab = the Cout the
 * kernel was launched with, O = Cout_real.  (autograd convolution_backward's weight term, as dh_conv2d_wgrad.) */
int dh_conv2d_wgrad_partial(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N, int H,
                            int W, int Cin, int OH, int OW, int Cout, int ks, int stride, int pad, int groups,
                            int npix_valid, int use_tr, int Cout_real, int cin_pitch, int dilation, void* workspace,
                            int* splitk_out, void* stream);
/* dh_conv2d_wgrad (splitk_out == NULL) / dh_conv2d_wgrad_partial (splitk_out != NULL) against
 * relu(x * in_scale[g][ci] + in_shift[g][ci]) computed on load (see dh_conv2d_fwd's in_scale) */
int dh_conv2d_wgrad_bn_in(int dtype, const void* x, const void* dy, float* dw_oihw, int accumulate, int N, int H, int W,
                          int Cin, int OH, int OW, int Cout, int ks, int stride, int pad, int use_tr, int Cout_real,
                          int dilation, const float* in_scale, const float* in_shift, int in_groups, void* workspace,
                          int* splitk_out, void* stream);
/* Channel concatenation without the concatenated tensor (bf16, 3x3 / stride 1 / pad 1): models/networks.py:1344,
 * conv_layer2_0(torch.cat([a_128, b_128], 1)) -- here the two temporal streams are the two halves of ONE [2N]-image tensor.
 *   dh_conv3x3_split_fwd   x_split_bytes != 0: the input is cat([A, B], channel) of two [N][H][W][Cin / 2] tensors, A at x, B
 *                          at x + x_split_bytes; y_split_bytes != 0: the output's channel halves go to two
 *                          [N][H][W][Cout / 2] tensors, y and y + y_split_bytes (the data gradient of such a layer, with
 *                          w_packed / w_frag its data-gradient packs).  No bias / residual / activation; stats_partial as
 *                          dh_conv2d_fwd (rows = dh_conv2d_fwd_num_tiles).  w_frag (fragment-order pack) is required.
 *   dh_conv2d_wgrad_split  dh_conv2d_wgrad_partial against such an input (joins an open weight-gradient batch).
 *   dh_conv3x3_split_supported  1 when all three run for a layer Cin -> Cout on N x H x W pixels; callers otherwise
 *                          materialise the concatenation with dh_copy_channels.
 * (replaces torch.cat + F.conv2d and their autograd terms; the copies were 4 x 134 MB per step at batch 32) */
int dh_conv3x3_split_supported(int N, int H, int W, int Cin, int Cout);
int dh_conv3x3_split_fwd(const void* x, long x_split_bytes, const void* w_packed, const void* w_frag, void* y, long y_split_bytes,
                         float* stats_partial, int N, int H, int W, int Cin, int Cout, void* stream);
int dh_conv2d_wgrad_split(const void* x, long x_split_bytes, const void* dy, float* dw_oihw, int accumulate, int N, int H, int W,
                          int Cin, int Cout, void* workspace, int* splitk_out, void* stream);
/* classifier.0 on nn.Upsample(4, 'bilinear')(abs(x1 - x2)) WITHOUT that map (models/networks.py:383-389, models/help_funcs.py:9;
 * bf16): a, b [N][H / 4][W / 4][32] are the two streams' decoder outputs, H x W the fine size (multiples of 4).
 *   dh_conv3x3_up4_fwd   y [N][H][W][32] = act(conv3x3(upsample4(|a - b|)) + bias), act 0 / ReLU; w_packed [9][32][32]
 *                        (dh_pack_weight forward form); stats_partial as dh_conv2d_fwd, dh_conv2d_fwd_num_tiles(N, H, W, 32, 3, 1)
 *                        rows.  The 8 x 16-pixel tile's haloed input is interpolated from its 4 x 6 coarse footprint with the
 *                        terms and order of dh_absdiff_upsample4_fwd: equal to dh_conv2d_fwd on that kernel's output, bit for bit.
 * (the data gradient through the upsample is dh_conv3x3_dgrad_up4; the weight gradient still reads the materialised map:
 * F.interpolate + abs + F.conv2d.  Measured slower than dh_absdiff_upsample4_fwd + the register-resident-weights kernel -- 121
 * against 32 + 76 us at batch 32 -- and therefore not the default path) */
int dh_conv3x3_up4_fwd(const void* a, const void* b, const void* w_packed, const float* bias, int act, void* y, float* stats_partial,
                       int N, int H, int W, void* stream);
/* Batched weight gradients: the 3x3 stride-1 bf16 layers of one backward pass as ONE launch per kernel family (the
 * wave-specialised 64co x 64ci form: Cin and Cout multiples of 64; the 32-wide output tile: 16 < Cout <= 32).  Between dh_wgrad_batch_begin() and dh_wgrad_batch_end(), dh_conv2d_wgrad_partial /
 * dh_conv2d_wgrad_bn_in (with splitk_out) only RECORD an eligible layer -- *splitk_out is its in-batch slice count, smaller
 * than a launch of its own would take -- and every other layer launches as before; dh_wgrad_batch_launch(stream) issues what
 * has been recorded (x / dy / scale / shift / workspace must stay alive and unchanged until then; a 17th layer issues the
 * first 16), dh_wgrad_batch_pending() counts the recorded layers.  The reduce (dh_wgrad_reduce_multi) follows as before.
 * The state is per host thread.  Replaces nothing in the reference (autograd computes each layer's gradient on its own):
 * it exists because a launch per layer needs 256 workgroups each to fill the chip. */
int dh_wgrad_batch_begin(void);
int dh_wgrad_batch_pending(void);
int dh_wgrad_batch_launch(void* stream);
int dh_wgrad_batch_end(void* stream);
int dh_wgrad_batch_abort(void);        /* closes the batch without launching (an aborted pass) */
int dh_wgrad_reduce_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream);
/* ---- the 2x2 phase form of conv3x3(nearest-upsample-x2(x)) with 32 output channels (dh_conv2d_fwd's phase_mode) ----
 * dh_pack_phase_weights: OIHW fp32 [32][Cin][3][3] (+ bias [32]) -> fwd [4 taps][4 * 32][Cin] T, data-gradient form
 * [4 taps][Cin][4 * 32] T, bias4 [128] fp32; W_ab[t][u] = sum of the 3x3 taps that read the same source pixel.
 * dh_conv2d_wgrad_phase: per-phase weight gradients, x [N][H][W][Cin], dy [N][2H][2W][32] -> partial slabs in `workspace`
 * ([4 phases][splitk][4 taps][32][Cin] fp32, *splitk_out slabs per phase; reduce each phase with dh_wgrad_reduce_multi /
 * the returned layout into dwab [4][32][Cin][2][2]).  dh_phase_wgrad_combine: dw_oihw [32][Cin][3][3] (+)= the sums of the
 * phase gradients that share a 3x3 tap.  (autograd of F.interpolate(nearest) + conv2d's weight.) */
int dh_pack_phase_weights(int dtype, const float* w_oihw, const float* bias, int Cin, void* fwd, void* dgrad, float* bias4,
                          void* stream);
long dh_conv2d_wgrad_phase_workspace_size(int N, int H, int W, int Cin);
int dh_conv2d_wgrad_phase(int dtype, const void* x, const void* dy, int N, int H, int W, int Cin, int use_tr, void* workspace,
                          int* splitk_out, void* stream);
int dh_phase_wgrad_combine(const float* dwab, float* dw_oihw, int Cin, int accumulate, void* stream);
/* Data gradient of a 3x3 / stride-2 / pad-1 convolution (models/resnet.py:24-27 with stride 2; autograd convolution_backward)
 * WITHOUT the zero-inserted gradient: dh_conv2d_fwd(phase_mode = 1, ks = 2, x = dY [N][OH][OW][Co], logical Cout = 4 * Ci,
 * y = dX [N][2 OH][2 OW][Ci], residual = coarse [N][OH][OW][Ci] gradient of the 1x1 stride-2 shortcut or NULL) over the
 * weights packed here: [4 taps][4 * Ci][Co] T from OIHW fp32 [Co][Ci][3][3]; Ci = 32 or 64. */
int dh_pack_s2_dgrad_phase_weights(int dtype, const float* w_oihw, int Co, int Ci, void* out, void* stream);
int dh_wgrad_reduce_job_size(void);
int dh_wgrad_reduce_outputs_per_block(int Cin);
int dh_conv2d_wgrad_splitk(int N, int OH, int OW, int Cin, int Cout, int ks, int groups);

/* OIHW fp32 master weight -> kernel layouts: fwd [ks*ks][OPad][I] T and (optional) the data-gradient
 * form [ks*ks flipped][IPad][max(O, dgrad_inner)] T.  out_scale (optional, [O]) multiplies the forward form per
 * output channel: eval-mode BatchNorm folded into the convolution (its shift then goes in as `bias`). */
int dh_pack_weight(int dtype, const float* w_oihw, const float* out_scale, int O, int I, int ks, int OPad, void* fwd, int IPad,
                   int dgrad_inner, void* dgrad, void* stream);
/* every weight of a net in one launch.  jobs_dev: njobs records {const float* w; void* fwd; void* dgrad; int O, I, KS,
 * OPad, IPad, OK, dtype (| 0x200: fragment-order destinations, see dh_conv2d_fwd), first_block, nblocks;} (dh_pack_job_size() bytes each) in device memory, sorted by
 * first_block; record k is served by workgroups [first_block, first_block + nblocks). */
int dh_pack_weights_multi(const void* jobs_dev, int njobs, int total_blocks, void* stream);
int dh_pack_job_size(void);
/* z[n,2y,2x,c] = dy[n,y,x,c] (zero elsewhere): stride-2 data gradients as stride-1 convolutions */
int dh_zero_insert2(int dtype, const void* dy, void* z, int N, int OH, int OW, int H, int W, int C, void* stream);
/* x [N][H][W][C] (+)= coarse [N][(H+1)/2][(W+1)/2][C] at the even-even positions: the data gradient of a 1x1 stride-2 convolution
 * (the shortcut of a stride-2 Bottleneck, models/resnet.py:106-118) computed on its own coarse grid and added where it lives,
 * instead of zero insertion + a 1x1 convolution on the fine grid. */
int dh_add_coarse(int dtype, void* x, const void* coarse, int N, int OH, int OW, int H, int W, int C, void* stream);

/* stem nn.Conv2d(3,64,7,2,3) (models/resnet.py:150) as a 4x4/stride-1 conv on a space-to-depth image */
int dh_stem_space_to_depth(int dtype, const float* x_nchw, void* y, int N, int H, int W, int CP, void* stream);
int dh_stem_pack_weight(int dtype, const float* w_oihw, const float* out_scale, void* packed, int O, int CP, void* stream);
int dh_stem_unpack_grad(const float* dw2, float* dw_oihw, int O, int CP, int accumulate, void* stream);
/* The same stem (models/resnet.py:150, applied to both images by forward_single, models/networks.py:215-224) as one bf16
 * kernel on the NCHW fp32 images themselves: images [0, B) from xa, [B, N) from xb (xb may be NULL when B == N).
 * w_oihw fp32 [64][3][7][7]; out_scale / bias (optional, per cout): eval-mode BatchNorm folded in; relu: 0 / 1.
 * y: bf16 NHWC [N][H/2][W/2][64].  stats (optional, raw convolution only): [2][64][dh_stem7_fwd_num_slots] sum / sum of
 * squares per workgroup for dh_bn_finalize (ntiles = slots); a workgroup stays inside one of the `groups` equal image
 * ranges, slots of group 0 first.  xs16 (optional): the space-to-depth image [N][H/2][W/2][16] bf16 (channel (ry*2+rx)*3 + c, 4 zero
 * channels) the weight gradient reads (dh_conv2d_wgrad with ks = 4, Cin = 16, pitch 16 + dh_stem_unpack_grad). */
int dh_stem7_fwd(const float* xa, const float* xb, int B, int N, int H, int W, const float* w_oihw, const float* out_scale,
                 const float* bias, int relu, void* y, float* stats, int groups, void* xs16, void* stream);
int dh_stem7_fwd_num_slots(int N, int H, int W, int groups);
/* Backward of the stem's tail maxpool(relu(bn1(conv1(x)))) (models/resnet.py:150-153,198-201) without a BatchNorm pass of its
 * own (bf16).  dh_stem_pool_bn_bwd: max-pool backward from the saved arg-max, the ReLU mask recomputed from y
 * (y * mask_scale + mask_shift > 0) and the BatchNorm-backward sums in ONE pass over the 128x128 map: d [N][H][W][C] = masked
 * gradient of the BatchNorm output, dgamma / dbeta (+)=, coef [groups][3][C] = per-channel (A, B, C) of
 * dconv = A * d + B * y + C.  dh_stem_wgrad_bn: the weight gradient against the space-to-depth image xs16 with that
 * expression applied while d and y are loaded; dw2 [64][16][4][4] is assigned (then dh_stem_unpack_grad). */
long dh_stem_pool_bn_bwd_workspace_size(int C, int groups);
int dh_stem_pool_bn_bwd(const unsigned char* argmax, const void* dpool, const void* y, const float* mask_scale,
                        const float* mask_shift, const float* mean, const float* invstd, const float* gamma, int N, int H, int W,
                        int C, int groups, void* d, float* coef, float* dgamma, float* dbeta, int accumulate, void* workspace,
                        void* stream);
/* dh_stem_pool_bn_bwd with a second gradient of the same pre-pool activation, extra [N][H][W][C] (or NULL), added before the
 * mask: the hierarchical model reads the stem's output twice (models/networks.py:1118-1128 max-pool -> layer1, :1344
 * cat([a_128, b_128]) -> conv_layer2_0), so the add, the max-pool backward and both BatchNorm-backward passes are this one. */
int dh_stem_pool_bn_bwd_plus(const unsigned char* argmax, const void* dpool, const void* extra, const void* y,
                             const float* mask_scale, const float* mask_shift, const float* mean, const float* invstd,
                             const float* gamma, int N, int H, int W, int C, int groups, void* d, float* coef, float* dgamma,
                             float* dbeta, int accumulate, void* workspace, void* stream);
int dh_stem_wgrad_bn(const void* xs16, const void* d, const void* y, const float* coef, int groups, int N, int OH, int OW,
                     float* dw2, int use_tr, void* workspace, void* stream);

/* ---- BatchNorm2d (models/resnet.py:152,40-44; help_funcs.py:11) and LayerNorm(32) (help_funcs.py:34-49) */
int dh_bn_finalize(const float* partial, int ntiles, int CP, int C, int groups, double count, const float* gamma,
                   const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                   float* mean, float* invstd, float* scale, float* shift, long long* num_batches_tracked,
                   void* stream);   /* partial: [2][CP][ntiles]; num_batches_tracked (optional) += groups */
int dh_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* scale, float* shift, void* stream);
int dh_bn_apply(int dtype, const void* x, const void* residual, void* y, const float* scale, const float* shift,
                long npix, int C, int groups, int act, void* stream);
int dh_bn_bwd(int dtype, const void* dout, const void* out_relu, const void* x, const float* mean,
              const float* invstd, const float* gamma, long npix, int C, int groups, void* dx, void* dres,
              float* dgamma, float* dbeta, int accumulate, const float* mask_scale, const float* mask_shift,
              void* workspace, void* stream);
/* ReLU mask of the layer: out_relu (its post-activation output) OR mask_scale / mask_shift [groups][C] (the forward's
 * scale / shift: mask = x * scale + shift > 0, for layers without a residual input) OR neither (no ReLU) */
long dh_bn_bwd_workspace_size(long npix, int C, int groups);
/* dh_bn_bwd as ONE persistent launch (csrc/bn_bwd_persist.hip): 256 workgroups hold dout (registers) and x (LDS) on chip
 * across a device-wide barrier, so each tensor is read once.  bf16 only; dh_bn_bwd_persist_supported() tells whether the
 * layer fits (otherwise call dh_bn_bwd).  sync: 8192 zero-initialised uint32 words owned by the caller and reused by every
 * call on ONE stream.  The barrier needs all 256 workgroups resident: never launch it where a kernel of another stream
 * (a collective, a side-stream launch) may hold CUs.  Error word sync[2]: bit 0 = the bounded barrier spin timed out, bit 1 =
 * a non-finite / out-of-range partial sum; when it is set dgamma / dbeta come out NaN.  dh_bn_bwd_persist_status() returns
 * and clears the word (synchronises the stream; < 0: the copy failed).  workspace: as dh_bn_bwd.  Same autograd call site. */
int dh_bn_bwd_persist_supported(int dtype, long npix, int C, int groups);
int dh_bn_bwd_persist_preferred(int dtype, long npix, int C, int groups);   /* supported AND large enough to be faster */
int dh_bn_bwd_persist(const void* dout, const void* out_relu, const void* x, const float* mean, const float* invstd,
                      const float* gamma, long npix, int C, int groups, void* dx, void* dres, float* dgamma, float* dbeta,
                      int accumulate, const float* mask_scale, const float* mask_shift, void* workspace, unsigned* sync,
                      void* stream);
/* The ReLU mask of a BatchNorm + residual + ReLU layer (models/resnet.py:56-71, 104-121: out = relu(bn(y) + identity)) as BYTES
 * instead of the post-activation tensor: dh_bn_apply_bits = dh_bn_apply that also writes relu_bits [npix * C / V] (V = 8 bf16 /
 * 4 fp32 elements per 16-byte piece) -- byte i holds the mask of piece i of y, bit j = (y[V i + j] > 0) -- and dh_bn_bwd_bits /
 * dh_bn_bwd_persist_bits (bf16) =
 * dh_bn_bwd / dh_bn_bwd_persist reading those bytes where the latter read out_relu: one sixteenth of a tensor pass instead of
 * one (persistent form) or two (two-pass form).  Same results bit for bit.  Replaces the same autograd nodes as dh_bn_bwd
 * (native_batch_norm_backward + threshold_backward). */
int dh_bn_apply_bits(int dtype, const void* x, const void* residual, void* y, const float* scale, const float* shift, long npix,
                     int C, int groups, int act, unsigned char* relu_bits, void* stream);
int dh_bn_bwd_bits(int dtype, const void* dout, const unsigned char* relu_bits, const void* x, const float* mean, const float* invstd,
                   const float* gamma, long npix, int C, int groups, void* dx, void* dres, float* dgamma, float* dbeta, int accumulate,
                   void* workspace, void* stream);
int dh_bn_bwd_persist_bits(const void* dout, const unsigned char* relu_bits, const void* x, const float* mean, const float* invstd,
                           const float* gamma, long npix, int C, int groups, void* dx, void* dres, float* dgamma, float* dbeta,
                           int accumulate, void* workspace, unsigned* sync, void* stream);
int dh_bn_bwd_persist_status(unsigned* sync, void* stream);
int dh_bn_bwd_persist_test_spin_limit(unsigned limit);   /* tests: limit > 0 forces the timeout path (0 restores 2^22 spins) */
int dh_bn_bwd_from_partials(int dtype, const void* g, const void* x, const float* partial, int ntiles, const float* mean,
                            const float* invstd, const float* gamma, long npix, int C, int groups, void* dx,
                            float* dgamma, float* dbeta, int accumulate, void* workspace, void* stream);
int dh_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* y, float* stats,
                     long rows, int C, float eps, void* stream);
int dh_layernorm_bwd(int dtype, const void* dy, const void* x, const float* stats, const float* gamma, void* dx,
                     const void* dx_add, float* dgamma, float* dbeta, int accumulate, long rows, int C,
                     void* workspace, void* stream);
long dh_layernorm_bwd_workspace_size(long rows);
int dh_reduce_partials(const float* partial, long nt, long n, float scale, float* out, int accumulate, void* stream);

/* ---- pointwise / resampling / layout (models/resnet.py:154; networks.py:199-200,384,1312,1348) ---- */
/* input pipeline on the device (replaces the per-sample PIL work of datasets/data_utils.py:55-111 -- crop window, h / v
 * flip, ToTensor + Normalize(0.5, 0.5) -- for pre-decoded pairs): a, b [S][H][W][3] uint8, l [S][H][W] uint8 (or NULL);
 * output sample n takes source pair idx[n] with params[n] = {x0, y0, hflip, vflip}; out_a / out_b fp32 [N][3][h][w],
 * out_l uint8 [N][1][h][w]. */
int dh_augment_pairs_u8(const unsigned char* a, const unsigned char* b, const unsigned char* l, const int* idx,
                        const int* params, int N, int H, int W, int h, int w, float* out_a, float* out_b,
                        unsigned char* out_l, void* stream);
int dh_nchw_to_nhwc(int dtype, const float* src, void* dst, int N, int C, long HW, int CP, void* stream);
/* data gradient of the class head (3x3 / s1 / p1, 32 -> n_class <= 8 channels; help_funcs.py:13-14, networks.py:1247):
 * dy [N][H][W][CP] (CP = 8 bf16 / 4 or 8 fp32 channels per pixel, the first NC real), w_oihw [NC][32][3][3] fp32,
 * dx [N][H][W][32] */
int dh_head_dgrad3x3(int dtype, const void* dy, int CP, const float* w_oihw, int NC, void* dx, int N, int H, int W,
                     void* stream);
/* The same data gradient for a head whose 32 input channels are the output of a ReLU (classifier(conv_layer2(...)),
 * models/networks.py:1351-1355; bf16, n_class <= 2, CP = 8): relu_out [N][H][W][32] is that output and
 * dx = gradient * (relu_out > 0) -- the activation's backward pass (3 x 134 MB at batch 32) folded into this kernel. */
int dh_head_dgrad3x3_relu(const void* dy, const float* w_oihw, int NC, const void* relu_out, void* dx, int N, int H, int W,
                          void* stream);
/* The same data gradient GATED for the BatchNorm + ReLU behind the 32 channels (classifier[0..2], models/help_funcs.py:7-15; bf16,
 * n_class <= 2): g = gradient * (y * mask_scale + mask_shift > 0) and the per-workgroup partials [2][32][blocks] (sum g,
 * sum g * xhat) for dh_bn_bwd_from_partials -- that BatchNorm's reduction pass disappears. */
int dh_head_dgrad3x3_bn_blocks(int N, int H, int W, int groups);
int dh_head_dgrad3x3_bn(const void* dy, const float* w_oihw, int NC, const void* y, const float* mask_scale, const float* mask_shift,
                        const float* mean, const float* invstd, int groups, void* g, float* partial, int N, int H, int W,
                        void* stream);
/* The head's data gradient AND the backward of the BatchNorm + ReLU behind its 32 channels in two passes that never write the
 * gradient in between (n_class <= 8; the autograd of Conv2d(32, n_class, 3) <- ReLU <- BatchNorm2d(32),
 * models/help_funcs.py:7-15): each pass forms g = (y * mask_scale + mask_shift > 0) * (W^T (*) dlogits) on the matrix cores from
 * dlp, the dlogits as [N][H + 2][W + 2] class pieces inside a border of zeros (dh_head_dlogits_pack from the loss kernel's
 * [N][n_class][H][W] fp32: a piece = 2 classes in one word for n_class <= 2, 8 classes in 16 bytes for 3 .. 8 -- the five-class
 * heads of the xBD nets; a tap is one load, no bounds test); pass 1 reduces (sum g, sum g y), pass 2 writes dx = gamma invstd (g - (s1 + xhat s2) / M).  y
 * [N][H][W][32] pre-BatchNorm, statistics [groups][32] as dh_bn_finalize left them; dgamma / dbeta [32] (+)= when accumulate. */
int dh_head_bn_bwd_blocks(int N, int H, int W, int groups);
long dh_head_bn_bwd_workspace_size(int N, int H, int W, int groups);
/* dw [n_class][32][3][3] / db [n_class] (both or NULL): the head convolution's own weight and bias gradient (input =
 * relu(BatchNorm(y)) as bf16), taken by pass 1 from the loads it makes anyway -- pixels are the K dimension of an MFMA through
 * wave-private LDS tiles read back transposed; (+)= when accumulate. */
/* dtype DH_DTYPE_BF16: dlp one word per pixel; DH_DTYPE_F32 (the fp32 pipeline under dh_set_f32_mma_mode != 0): y / dx fp32, dlp two
 * words per pixel (bf16 heads, bf16 remainders) and every product as the three split bf16 products of that mode. */
int dh_head_dlogits_pack(int dtype, const float* dlogits_nchw, int N, int NC, int H, int W, void* dlp, void* stream);
int dh_head_bn_bwd(int dtype, const void* dlp, const float* w_oihw, int NC, const void* y, const float* mask_scale, const float* mask_shift,
                   const float* mean, const float* invstd, const float* gamma, int groups, void* dx, float* dgamma, float* dbeta,
                   float* dw, float* db, int accumulate, int N, int H, int W, void* workspace, void* stream);
/* The class head behind a ReLU (classifier(conv_layer2(...)), models/networks.py:1351-1355; n_class <= 8): the data gradient
 * dx = (relu_out > 0) * (W^T (*) dlogits) of dh_head_dgrad3x3_relu, from the pair map dlp (dh_head_dlogits_pack), AND the head's
 * own weight / bias gradient dw [n_class][32][3][3] / db [n_class] ((+)= when accumulate) from the same loads of relu_out --
 * the head's input.  workspace: dh_head_bn_bwd_workspace_size(N, H, W, 1) bytes. */
int dh_head_relu_bwd(int dtype, const void* dlp, const float* w_oihw, int NC, const void* relu_out, void* dx, float* dw, float* db,
                     int accumulate, int N, int H, int W, void* workspace, void* stream);
/* The class head's forward (nn.Conv2d(32, n_class, 3, padding=1): models/help_funcs.py:13-14, networks.py:1247 / 1355) for
 * n_class <= 2 and bf16 activations: x [N][H][W][32] (pre-BatchNorm when in_scale / in_shift [in_groups][32] are given: the
 * head's input is relu(x * scale + shift)), w_oihw [n_class][32][3][3] fp32, bias [n_class] or NULL -> logits [N][n_class][H][W]
 * fp32.  The contraction over the input channels is done once per input pixel for all nine taps (M = (tap, class) = 18), the
 * convolution is a nine-term gather from a ring of rows in LDS: no halo, no padded output channels.  dh_head_fwd_supported:
 * whether (n_class, W) is this kernel's shape (else dh_conv3x3_head_fwd). */
int dh_head_fwd_supported(int NC, int W);
/* dtype DH_DTYPE_F32: x fp32 and every product as three fp16-plane products -- the arithmetic of dh_set_f32_mma_mode(3), for
 * callers in that mode only. */
int dh_head_fwd(int dtype, const void* x, const float* w_oihw, const float* bias, int NC, const float* in_scale, const float* in_shift,
                int in_groups, float* logits_nchw, int N, int H, int W, void* stream);
/* A job table of small element-wise kernels in ONE launch (no reference counterpart: autograd runs one op at a time): the three
 * levels of _forward_trans_module (models/networks.py:1297-1318) each issue a positional add, the channel concatenation of the two
 * streams, |token2 - token1| and their gradients between the launches they share; the independent ones of a round go out together.
 * n <= 12 jobs that must not depend on each other; per job an op and its operands (see csrc/pointwise.hip dh_ew_multi):
 * ADD_POS = dh_add_pos (bf16, C % 8 == 0), CAT_HALVES / SPLIT_HALVES = dh_cat_halves (bf16), ABSDIFF_HALVES(_BWD) =
 * dh_absdiff_halves(_bwd) (fp32), ADD_POS_BWD = dh_add_pos_bwd (bf16, C = 32). */
enum { DH_EW_ADD_POS = 1, DH_EW_CAT_HALVES = 2, DH_EW_SPLIT_HALVES = 3, DH_EW_ABSDIFF_HALVES = 4, DH_EW_ABSDIFF_HALVES_BWD = 5,
       DH_EW_ADD_POS_BWD = 6 };
int dh_ew_multi(int n, const int* op, const void* const* a, const void* const* b, void* const* c, const int* i0, const int* i1,
                const long* l0, void* stream);
int dh_nhwc_to_nchw(int dtype, const void* src, float* dst, int N, int C, long HW, void* stream);
int dh_copy_channels(int dtype, const void* src, int Cs, int sc0, void* dst, int Cd, int dc0, int Cn, long P, void* stream);
/* torch.cat([x1, x2], 1) of the two temporal streams (models/networks.py:1309, 1344), which are the two batch halves of
 * t [2 P][C] here: cat [P][2 C] <- t (inverse = 0), or t <- cat (inverse = 1: the concatenation's gradient back to the
 * streams); both halves in one launch.  C a multiple of the 16-byte piece (8 bf16 / 4 fp32). */
int dh_cat_halves(int dtype, void* t, void* cat, int C, long P, int inverse, void* stream);
int dh_add(int dtype, const void* a, const void* b, void* y, long n, void* stream);
int dh_add_pos(int dtype, const void* x, const float* pos, void* y, int N, long HW, int C, void* stream);
int dh_add_pos_bwd(int dtype, const void* dy, float* dpos, int N, long HW, int C, int accumulate, void* stream);
int dh_act_bwd(int dtype, const void* dy, const void* ref, void* dx, long n, int act, void* stream);
/* argmax (optional, [N][OH][OW][C] bytes): window position 0..8 of the first maximum, consumed by the backward */
int dh_maxpool3x3s2_fwd(int dtype, const void* x, void* y, unsigned char* argmax, int N, int H, int W, int C,
                        const float* bn_scale, const float* bn_shift, int groups, void* stream);
/* bn_scale / bn_shift [groups][C] (optional): x is the PRE-normalisation input of a train-mode BatchNorm + ReLU whose
 * only consumer is this pool (the ResNet stem, models/resnet.py:207-210): relu(x * scale + shift) is applied on load */
int dh_maxpool3x3s2_bwd(int dtype, const unsigned char* argmax, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int dh_upsample2_nearest_fwd(int dtype, const void* x, void* y, int N, int H, int W, int C, void* stream);
int dh_upsample2_nearest_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int C, void* stream);
int dh_absdiff_upsample4_fwd(int dtype, const void* a, const void* b, void* y, int N, int H, int W, int C, void* stream);
int dh_absdiff_upsample4_bwd(int dtype, const void* a, const void* b, const void* dy, void* da, void* db, int N, int H, int W, int C, void* stream);
int dh_absdiff(int dtype, const void* a, const void* b, void* y, long n, void* stream);
int dh_absdiff_bwd(int dtype, const void* a, const void* b, const void* dy, void* da, void* db, long n, int accumulate, void* stream);
int dh_colsum(int dtype, const void* x, long P, int C, float* out, int accumulate, void* workspace, void* stream);  /* workspace: 1024 * C floats */
int dh_cast_from_f32(int dtype, const float* src, void* dst, long n, void* stream);
int dh_cast_to_f32(int dtype, const void* src, float* dst, long n, int accumulate, void* stream);

/* ---- token side (models/networks.py:312-336,457-488; help_funcs.py:66-114) --------------------
 * xattn_prep: wq / wk / wv / wo are the fp32 masters ([inner][32] resp. [32][inner]); wkT, wvT, wqT ([32][inner])
 * and woT ([inner][32]) are their transposes in T as produced by dh_pack_weight's data-gradient form. */
int dh_tokenizer_fwd(int dtype, const void* x, const float* wa, const float* pos, int S, int B, int HW, int L,
                     float* logits, float* stats, float* pooled, float* tok_cat, void* workspace, void* stream);
long dh_tokenizer_fwd_workspace_size(int S, int HW, int L);
int dh_tokenizer_bwd(int dtype, const void* x, const float* wa, int S, int B, int HW, int L, const float* logits,
                     const float* stats, const float* pooled, const float* dtok_cat, void* dx_accum, float* dwa,
                     float* dpos, int accumulate, void* workspace, void* stream);
long dh_tokenizer_bwd_workspace_size(int S, int HW, int L);
int dh_xattn_prep_fwd(int dtype, const void* tok, long tok_bstride, long tok_sstride, int B, int S, int L, int heads,
                      int dim_head, int HLP, float scale, float eps, const float* ln_g, const float* ln_b,
                      const float* wq, const void* wkT, const void* wvT, const void* woT, float* mn, float* mstats,
                      float* k, float* v, void* kq, void* kqT, void* vo, void* voT, void* stream);
/* stacked forms: ONE launch prepares all `layers` of a decoder stack (they read the same tokens).  Parameter /
 * gradient pointers are the first layer's, consecutive layers lie param_stride floats apart (the net's flat arena);
 * wkT / wvT / woT / wqT and every saved / output tensor are stacked [layers][single-layer shape]. */
int dh_xattn_prep_fwd_stack(int dtype, const void* tok, long tok_bstride, long tok_sstride, int B, int S, int L, int heads,
                            int dim_head, int HLP, float scale, float eps, int layers, long param_stride,
                            const float* ln_g, const float* ln_b, const float* wq, const void* wkT, const void* wvT,
                            const void* woT, float* mn, float* mstats, float* k, float* v, void* kq, void* kqT, void* vo,
                            void* voT, void* stream);
/* The same preparation on the matrix cores (csrc/tokens.hip: four images = the 16 columns of one MFMA per workgroup, heads over
 * the waves) for bf16 nets with L = 4, S a multiple of 4, dim_head 32 / 64, HLP = 32 (dh_xattn_prep_mfma_supported).  wk / wv
 * ([inner][32]) and wo ([32][inner]) are the first layer's fp32 masters, wqT the stacked bf16 transposes [layers][32][inner]. */
int dh_xattn_prep_mfma_supported(int dtype, int S, int L, int heads, int dim_head, int HLP);
int dh_xattn_prep_fwd_stack_mfma(const void* tok, long tok_bstride, long tok_sstride, int B, int S, int heads, int dim_head,
                                 int HLP, float scale, float eps, int layers, long param_stride, const float* ln_g,
                                 const float* ln_b, const float* wk, const float* wv, const float* wo, const void* wqT,
                                 float* mn, float* mstats, float* k, float* v, void* kq, void* kqT, void* vo, void* voT,
                                 void* stream);
int dh_xattn_prep_bwd_stack_mfma(const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride, int B, int S, int heads,
                                 int dim_head, int HLP, float scale, int layers, long param_stride, const float* ln_g,
                                 const float* wq, const void* woT, const void* wkT, const void* wvT, const float* mn,
                                 const float* mstats, const float* k, const float* v, const float* dkq, const float* dvoT,
                                 float* dk, float* dv, float* dln_g, float* dln_b, float* dwq, float* dwk, float* dwv,
                                 float* dwo, int accumulate, void* workspace, void* stream);
int dh_xattn_prep_bwd_stack(int dtype, const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride, int B, int S,
                            int L, int heads, int dim_head, int HLP, float scale, int layers, long param_stride,
                            const float* ln_g, const void* wqT, const float* wk, const float* wv, const float* wo,
                            const float* mn, const float* mstats, const float* k, const float* v, const float* dkq,
                            const float* dvoT, float* dk, float* dv, float* dln_g, float* dln_b, float* dwq, float* dwk,
                            float* dwv, float* dwo, int accumulate, void* workspace, void* stream);
long dh_xattn_prep_bwd_stack_workspace_size(int S, int L, int layers);
int dh_xattn_prep_bwd(int dtype, const void* tok, void* dtok_accum, long tok_bstride, long tok_sstride, int B, int S,
                      int L, int heads, int dim_head, int HLP, float scale, const float* ln_g, const void* wqT,
                      const float* wk, const float* wv, const float* wo, const float* mn, const float* mstats,
                      const float* k, const float* v, const float* dkq, const float* dvoT, float* dk, float* dv,
                      float* dln_g, float* dln_b, float* dwq, float* dwk, float* dwv, float* dwo, int accumulate,
                      void* workspace, void* stream);
long dh_xattn_prep_bwd_workspace_size(int S);
int dh_softmax_groups_fwd(int dtype, const void* x, void* y, long rows, int heads, int L, int HLP, void* stream);
int dh_softmax_groups_bwd(int dtype, const void* y, const void* dy, void* dx, long rows, int heads, int L, int HLP, void* stream);
int dh_self_attn_fwd(int dtype, const void* qkv, void* o, float* attn, int B, int n, int heads, int dim_head, float scale, void* stream);
int dh_self_attn_bwd(int dtype, const void* qkv, const float* attn, const void* dout, void* dqkv, int B, int n, int heads, int dim_head, float scale, void* stream);

/* Fused token encoder stack (models/networks.py:457-512, help_funcs.py:117-167) on the 2*token_len <= 8 tokens of an
 * image pair, fp32: one workgroup per image and direction + one launch for all parameter gradients.  x, y, dy, dx:
 * [B][n][32]; parameters are the first layer's pointers in torch layouts (to_qkv [3*inner][32], to_out [32][inner],
 * net.0 [mlp][32], net.3 [32][mlp]), consecutive layers param_stride floats apart; gradients are accumulated.
 * dh_encoder_supported: 1 when the shape fits the kernels (n <= 8, mlp <= 64, working set within the 160 KB LDS). */
int dh_encoder_supported(int n, int heads, int dim_head, int mlp);
int dh_encoder_fwd(const float* x, float* y, float* saved_inputs, int B, int n, int depth, int heads, int dim_head,
                   int mlp, float scale, float eps, long param_stride, const float* ln1_g, const float* ln1_b,
                   const float* wqkv, const float* wo, const float* bo, const float* ln2_g, const float* ln2_b,
                   const float* w1, const float* b1, const float* w2, const float* b2, void* stream);
int dh_encoder_bwd(const float* dy, float* dx, const float* saved_inputs, int B, int n, int depth, int heads,
                   int dim_head, int mlp, float scale, float eps, long param_stride, const float* ln1_g,
                   const float* ln1_b, const float* wqkv, const float* wo, const float* bo, const float* ln2_g,
                   const float* ln2_b, const float* w1, const float* b1, const float* w2, const float* b2,
                   float* dln1_g, float* dln1_b, float* dwqkv, float* dwo, float* dbo, float* dln2_g, float* dln2_b,
                   float* dw1, float* db1, float* dw2, float* db2, void* workspace, void* stream);
/* Batched encoder stacks (csrc/encoder_fused.hip): a stack runs one workgroup per image, i.e. a launch of its own keeps B of
 * the 256 CUs busy; the three levels of the hierarchical model are independent, so their stacks can share a launch.  Between
 * dh_encoder_batch_begin() and _end(), dh_encoder_fwd / dh_encoder_bwd only RECORD (up to four of each direction; a fifth
 * issues the first four); dh_encoder_batch_launch(stream) issues the recorded forward stacks as one launch and / or the
 * recorded backward stacks as one data-gradient + one parameter-gradient launch.  Every buffer of a recorded call (its
 * workspace included: one per call) must stay alive and unchanged until then.  State per host thread; _abort drops the
 * recorded calls without launching. */
int dh_encoder_batch_begin(void);
int dh_encoder_batch_pending(void);
int dh_encoder_batch_launch(void* stream);
int dh_encoder_batch_end(void* stream);
int dh_encoder_batch_abort(void);
/* The same for the fused decoder layers (csrc/decoder_fused.hip): between _begin and _end, dh_decoder_layer_fwd and the
 * data-gradient-only form of dh_decoder_layer_bwd (dw1 == NULL) only RECORD (up to four per direction and MLP width);
 * dh_decoder_batch_launch issues the recorded layers -- of INDEPENDENT stacks: never two layers of one stack -- as one launch
 * per direction and width. */
int dh_decoder_batch_begin(void);
int dh_decoder_batch_pending(void);
int dh_decoder_batch_launch(void* stream);
int dh_decoder_batch_end(void* stream);
int dh_decoder_batch_abort(void);
/* And for the cross-attention operand preparation of a decoder stack (csrc/tokens.hip; models/help_funcs.py:118-160, the
 * token side of Cross_Attention): between _begin and _end, dh_xattn_prep_fwd_stack_mfma and dh_xattn_prep_bwd_stack_mfma with
 * dim_head 64 only RECORD (up to four each).  _launch_fwd issues the recorded preparations as ONE launch; _launch_bwd the
 * recorded gradients as one launch per kernel of the family (operand gradients, token-gradient reduction, projection weight
 * gradients, LayerNorm gradients).  A recorded backward needs a workspace and dk / dv of its OWN until then.
 * dh_decoder_stack_bwd_finalize is recorded by the decoder batch in the same way and issued after its backward launches, so
 * the order `dh_xprep_batch_launch_fwd, dh_decoder_batch_launch, dh_xprep_batch_launch_bwd` is right for calls recorded
 * in program order.  _pause(1): calls launch at once although a batch is open; _pause(0) records again. */
int dh_xprep_batch_begin(void);
int dh_xprep_batch_pending(void);
int dh_xprep_batch_pause(int paused);
int dh_xprep_batch_launch_fwd(void* stream);
int dh_xprep_batch_launch_bwd(void* stream);
int dh_xprep_batch_end(void* stream);
int dh_xprep_batch_abort(void);
long dh_encoder_bwd_workspace_size(int B, int n, int depth, int heads, int dim_head, int mlp);
/* floats of `saved_inputs` (non-null in training): per (layer, image) the forward's intermediates -- layer input, LayerNorm
 * outputs and statistics, qkv, attention probabilities and output, MLP activations -- which dh_encoder_bwd reads back
 * instead of recomputing the layer */
long dh_encoder_saved_floats(int B, int n, int depth, int heads, int dim_head, int mlp);

/* Fused cross-attention decoder layer (help_funcs.py:170-186: Residual2(PreNorm2(Cross_Attention)) + Residual(PreNorm(
 * FeedForward))) in one kernel per direction; bf16, token_len 4, heads*4 <= 32, rows per image % 128 == 0.
 * kq / voT / vo / kqT are the per-image operands of dh_xattn_prep_fwd; w1 [mlp][32], w2 [32][mlp] (+ transposes). */
int dh_decoder_layer_fwd(const void* x, void* y, const void* kq, const void* voT, const float* ln1_g,
                         const float* ln1_b, const float* bo, const float* ln2_g, const float* ln2_b, const void* w1,
                         const float* b1, const void* w2, const float* b2, long rows, int rows_per_image, int mlp,
                         float eps, void* stream);
/* the same layer with OCP fp8 (e4m3) operands in its two ATTENTION products (BASELINE configs[4] "fp8 MFMA attention"): Kq /
 * VoT quantised per output row (absmax / 448) while they are staged, LN(x) / softmax probabilities converted at scale 1,
 * fp32 accumulation; output within ~3e-2 (relative L2) of dh_decoder_layer_fwd.  Forward only: the backward runs
 * dh_decoder_layer_bwd. */
int dh_decoder_layer_fwd_fp8(const void* x, void* y, const void* kq, const void* voT, const float* ln1_g,
                         const float* ln1_b, const float* bo, const float* ln2_g, const float* ln2_b, const void* w1,
                         const float* b1, const void* w2, const float* b2, long rows, int rows_per_image, int mlp,
                         float eps, void* stream);
int dh_decoder_layer_bwd(const void* x, const void* dy, void* dx, const void* kq, const void* voT, const void* vo,
                         const void* kqT, const float* ln1_g, const float* ln1_b, const float* bo, const float* ln2_g,
                         const float* ln2_b, const void* w1, const void* w1T, const float* b1, const void* w2,
                         const void* w2T, const float* b2, float* dw1, float* dw2, float* db1, float* db2, float* dbo,
                         float* dln1_g, float* dln1_b, float* dln2_g, float* dln2_b, float* dkq, float* dvoT, long rows,
                         int rows_per_image, int mlp, float eps, void* workspace, void* stream);
/* A whole fused decoder stack -- `depth` layers of the same shapes attending to ONE set of tokens (TransformerDecoder,
 * models/help_funcs.py:170-186: x = attn(x, m); x = ff(x) per layer with m fixed) -- in ONE launch per direction: a pixel row
 * only ever meets the tokens of its image, so a workgroup takes its rows through all layers, re-staging the layer's weights,
 * with no synchronisation between workgroups.  x [rows][32] bf16; ys [depth][rows][32] receives every layer's output
 * (ys[depth - 1] = the stack's; the backward recomputes each layer from its input).  Layer l's operands: kq / voT / vo / kqT
 * + l * kq_lstride elements (the dh_xattn_prep_fwd_stack outputs), packed MLP weights + l * w_lstride elements, fp32
 * parameter vectors + l * par_lstride floats.  dh_decoder_stack_bwd: dy = gradient of ys[depth - 1], dx = gradient of x, dwork
 * [rows][32] bf16 scratch; partials of layer l at workspace + l * dh_decoder_layer_bwd_workspace_size bytes, for
 * dh_decoder_stack_bwd_finalize.  Bit-identical to `depth` dh_decoder_layer_fwd / _bwd calls; both join an open decoder batch.
 * depth <= 8 (a workgroup keeps every layer's parameter vectors in LDS).  A launch of SEVERAL recorded jobs sizes the pixel
 * blocks of each for the launch as a whole (csrc/decoder_fused.hip dec_balance; DAHITRA_DEC_BALANCE=0: the per-job rule): y / dx
 * do not depend on it, the partial sums -- hence the parameter gradients -- to fp32 summation order.  The workspace of
 * dh_decoder_layer_bwd_workspace_size holds the smallest blocks such a launch may choose (only the blocks written are read);
 * the finalize looks the layout up by the workspace's address, so it must follow the backward that filled that workspace. */
int dh_decoder_stack_fwd(const void* x, void* ys, const void* kq, const void* voT, const float* ln1_g, const float* ln1_b,
                         const float* bo, const float* ln2_g, const float* ln2_b, const void* w1, const float* b1, const void* w2,
                         const float* b2, int depth, long kq_lstride, long w_lstride, long par_lstride, long rows,
                         int rows_per_image, int mlp, float eps, void* stream);
int dh_decoder_stack_bwd(const void* x, const void* ys, const void* dy, void* dx, void* dwork, const void* kq, const void* voT,
                         const void* vo, const void* kqT, const float* ln1_g, const float* ln1_b, const float* bo,
                         const float* ln2_g, const float* ln2_b, const void* w1, const void* w1T, const float* b1, const void* w2,
                         const void* w2T, const float* b2, int depth, long kq_lstride, long w_lstride, long par_lstride, long rows,
                         int rows_per_image, int mlp, float eps, void* workspace, void* stream);
/* dh_decoder_layer_bwd with dw1 == NULL leaves its per-workgroup partials in `workspace`; this sums the partials of the `depth`
 * layers of one decoder stack (same shapes; layer l's workspace at + l * dh_decoder_layer_bwd_workspace_size bytes, its gradients
 * at + l * grad_stride floats from layer 0's, its dkq / dvoT at + l * images * 1024 floats) in ONE launch. */
int dh_decoder_stack_bwd_finalize(const void* workspace, int depth, long rows, int rows_per_image, int mlp, float* dw1, float* dw2,
                                  float* db1, float* db2, float* dbo, float* dln1_g, float* dln1_b, float* dln2_g, float* dln2_b,
                                  long grad_stride, float* dkq, float* dvoT, void* stream);
long dh_decoder_layer_bwd_workspace_size(long rows, int rows_per_image, int mlp);

/* ---- loss, mask, optimizer (models/losses.py:106-196; trainer.py:39-40,170) ------------------- */
int dh_focal_loss(const float* logits_nchw, const long long* target, int B, int C, long HW, float alpha,
                  float grad_scale, float* loss_out, float* dlogits_nchw, void* workspace, void* stream);
int dh_argmax_nchw(const float* logits_nchw, long long* mask, int B, int C, long HW, void* stream);
/* batch-size-1 branch of the trainer (models/trainer.py:260-261 -> losses.py:9-26): F.cross_entropy with class
 * weights [1, 1], ignore_index, mean over the contributing pixels.  out_dev[0] = loss, out_dev[1] = pixel count.
 * workspace: 16 KiB. */
int dh_cross_entropy_fwd(const float* logits_nchw, const long long* target, int B, int C, long HW, int ignore_index,
                         float* out_dev, void* workspace, void* stream);
int dh_cross_entropy_bwd(const float* logits_nchw, const long long* target, int B, int C, long HW, int ignore_index,
                         const float* fwd_out_dev, const float* upstream_dev, float* dlogits_nchw, void* stream);
/* the gradient-free dice term of models/trainer.py:256-259 (losses.py:333-339): binary DiceLoss of
 * segmentation_models_pytorch applied to the arg-max mask (third party, not vendored: see oracle dice_constant).
 * workspace: 24 KiB. */
int dh_dice_argmax_constant(const float* logits_nchw, const long long* target, int B, int C, long HW, float eps,
                            float* loss_out, void* workspace, void* stream);
/* HIP-graph-capturable form: hyper_dev = [lr, beta1, beta2, eps, weight_decay, grad_scale, bc1, bc2_sqrt] and the
 * step counter live on the device; every call (or graph replay) advances the counter and the bias correction */
int dh_adamw_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                        float* hyper_dev, int* step_dev, void* stream);
/* arg-max over classes (first maximum wins) + confusion counts[gt*C+pred] (int64, accumulated); mask optional */
int dh_confusion_matrix(const float* logits_nchw, const long long* target, int B, int C, long HW, long long* mask,
                        long long* counts, void* stream);
int dh_scale_by_scalar(const float* src, const float* scalar_dev, float* dst, long n, void* stream);
/* |tok[b][1] - tok[b][0]| over [B][2][n] token sets (models/networks.py:1311) and its gradient */
int dh_absdiff_halves(int dtype, const void* tok, void* out, int B, long n, void* stream);
int dh_absdiff_halves_bwd(int dtype, const void* tok, const void* dout, void* dtok_accum, int B, long n, void* stream);
int dh_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);


/* ---- xBD train step (csrc/xbd_step.hip) -------------------------------------------
 * ComboLoss{dice, focal} per output channel on sigmoid(logits) (xBD_code/losses.py:24-34, 95-126, 273-288) with the
 * per-channel weights of xBD_code/train.py:348-353.  logits / masks are [B][C][HW] fp32.  fwd writes the channel
 * totals sums_out [C][4] = {sum s*t, sum s, sum t, sum focal} (needed by bwd), channel_loss_out [C] and loss_out [1]. */
long dh_combo_loss_workspace_size(int C);
int dh_combo_loss_fwd(const float* logits, const float* masks, int B, int C, long HW, const float* weights_dev,
                      float dice_weight, float focal_weight, float* sums_out, float* channel_loss_out, float* loss_out,
                      void* workspace, void* stream);
int dh_combo_loss_bwd(const float* logits, const float* masks, const float* sums, const float* weights_dev,
                      const float* upstream_dev, float dice_weight, float focal_weight, int B, int C, long HW,
                      float* dlogits, void* stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) over the flat gradient arena (xBD_code/train.py:373):
 * out_dev[0] = total L2 norm, out_dev[1] = min(1, max_norm / (norm + 1e-6)); the optimizer reads out_dev + 1 */
long dh_grad_norm_workspace_size(void);
int dh_grad_norm_clip_coef(const float* grad, long n, float max_norm, float* out_dev, void* workspace, void* stream);
/* xBD_code/adamw.py:37-86: like dh_adamw_step but denom = sqrt(v) + eps (eps before the bias correction) and the
 * gradient scale is read from device memory (NULL = 1) */
int dh_adamw_xbd_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int step, const float* grad_scale_dev, void* stream);
/* HIP-graph form: hyper_dev = [lr, beta1, beta2, eps, weight_decay, (unused), step_size (out)]; every call / replay
 * advances the device step counter */
int dh_adamw_xbd_step_graph(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float* hyper_dev,
                            int* step_dev, const float* grad_scale_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
