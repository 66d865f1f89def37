"""Per-kernel parity: every C-ABI entry point against a plain torch fp32 computation of the same op
on the CPU (same seeded inputs).  fp32 mode must agree to fp32 round-off (the parity mode of the
product); bf16 mode to bf16 round-off of inputs/outputs (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
# the matrix-product kernels additionally in the split forms of the fp32 mode (dh_set_f32_mma_mode 1 / 2 / 3: fp32 tensors; three
# products on two bf16 planes, six on three bf16 planes, three on two fp16 planes): same inputs, same fp32 tolerance (measured
# 5e-6 / 1e-6 / 8e-7 of the output scale)
DTYPES_MMA = [pytest.param(torch.float32, 0, id="float32"), pytest.param(torch.bfloat16, 0, id="bfloat16"),
              pytest.param(torch.float32, 1, id="bf16x3"), pytest.param(torch.float32, 2, id="bf16x6"),
              pytest.param(torch.float32, 3, id="f16x3")]


from _bounds import close, tol      # (shared with tests/test_bounds_cpu.py, which checks the checker)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g) * scale
    return x.to(dtype).float()        # value representable in `dtype`, kept as fp32 on the CPU


def dev(x, dtype):
    return x.to(dtype).cuda().contiguous()


def nhwc(x):      # NCHW cpu -> NHWC
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.fixture(scope="module")
def ops():
    from dahitra_amd import ops as o
    return o


@pytest.fixture(autouse=True)
def _f32_mma_mode(request, ops):
    """tests parametrized with `mma` (DTYPES_MMA) run with that dh_set_f32_mma_mode; every other test with the exact form"""
    cs = getattr(request.node, "callspec", None)
    with ops.f32_mma_mode(cs.params.get("mma", 0) if cs is not None else 0):
        yield


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(ks=3, stride=1, pad=1, cin=64, cout=64, h=24, w=40),
    dict(ks=3, stride=2, pad=1, cin=64, cout=128, h=32, w=32),
    dict(ks=1, stride=1, pad=0, cin=128, cout=32, h=16, w=16),
    dict(ks=1, stride=2, pad=0, cin=64, cout=128, h=32, w=32),
    dict(ks=3, stride=1, pad=1, cin=32, cout=2, h=20, w=20),
    dict(ks=3, stride=1, pad=1, cin=32, cout=5, h=16, w=16),
    dict(ks=3, stride=1, pad=1, cin=256, cout=256, h=16, w=16),
    dict(ks=3, stride=1, pad=2, dil=2, cin=64, cout=64, h=20, w=24),      # ResNet-50 layer3 (resnet.py:31-33)
    dict(ks=3, stride=1, pad=2, dil=2, cin=256, cout=256, h=16, w=32),    # 8-row tile at this N (16-row: bench-scale test below)
    dict(ks=1, stride=1, pad=0, cin=1024, cout=256, h=8, w=8),
])
def test_conv2d_fwd(ops, dtype, mma, cfg):
    N = 3
    x = rnd((N, cfg["cin"], cfg["h"], cfg["w"]), dtype, 1)
    w = rnd((cfg["cout"], cfg["cin"], cfg["ks"], cfg["ks"]), dtype, 2, scale=(cfg["cin"] * cfg["ks"] ** 2) ** -0.5)
    b = rnd((cfg["cout"],), torch.float32, 3, 0.1)
    dil = cfg.get("dil", 1)
    want_pre = F.conv2d(x, w, b, cfg["stride"], cfg["pad"], dil)
    r = rnd(tuple(want_pre.shape), dtype, 4)
    want = F.relu(want_pre + r)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    y, stats, pre = ops.conv2d(dev(nhwc(x), dtype), wp, cfg["cout"], cfg["ks"], cfg["stride"], cfg["pad"],
                               bias=b.cuda(), residual=dev(nhwc(r), dtype), act=ops.ACT_RELU, want_stats=True,
                               want_preact=True, dilation=dil)
    close(nchw(y), want, dtype, "conv2d out")
    close(nchw(pre), want_pre + r, dtype, "conv2d preact")
    tot = stats.sum(2).cpu()           # [2][CoutPad]
    close(tot[0, :cfg["cout"]], want.sum((0, 2, 3)), dtype, "stats sum", scale=float(want.abs().sum((0, 2, 3)).max()))
    close(tot[1, :cfg["cout"]], (want * want).sum((0, 2, 3)), dtype, "stats sumsq")


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_bf16_conv_statistics_buffer_under_a_split_fp32_mode(ops, mode):
    """dh_conv2d_fwd_num_tiles takes the dtype of the launch that fills the buffer: a bf16 convolution with BatchNorm partial sums
    issued while the thread's fp32 MMA mode is a split form (the library default under DAHITRA_F32_MMA, or after a bf16x3 engine
    ran in the thread) tiles by the bf16 rule -- 8-row tiles for 32 <= Cin < 128 where the split fp32 forms take 16 rows.  Sized
    by the fp32 rule the buffer had half the tile columns the kernel writes."""
    dtype, N, C, H = torch.bfloat16, 8, 64, 64                      # 8 x 4 x 8 = 256 tiles of 8 rows
    x = rnd((N, C, H, H), dtype, 21)
    w = rnd((C, C, 3, 3), dtype, 22, scale=(C * 9) ** -0.5)
    want = F.conv2d(x, w, None, 1, 1)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    with ops.f32_mma_mode(0):
        y0, st0 = ops.conv2d(dev(nhwc(x), dtype), wp, C, 3, 1, 1, want_stats=True)
    with ops.f32_mma_mode(mode):
        y1, st1 = ops.conv2d(dev(nhwc(x), dtype), wp, C, 3, 1, 1, want_stats=True)
    torch.cuda.synchronize()
    assert st1.shape == st0.shape
    assert torch.equal(y1, y0) and torch.equal(st1, st0)
    close(st1.sum(2).cpu()[0, :C], want.sum((0, 2, 3)), dtype, "stats sum", scale=float(want.abs().sum((0, 2, 3)).max()))


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("want_stats", [False, True])
@pytest.mark.parametrize("cfg", [
    dict(ks=3, stride=1, pad=1, cin=64, cout=64, h=32, w=32),       # tiles fully inside the image (unmasked statistics)
    dict(ks=3, stride=1, pad=1, cin=64, cout=64, h=24, w=40),       # ragged: masked statistics, partial stores
    dict(ks=3, stride=1, pad=1, cin=256, cout=256, h=32, w=32),     # Cin >= 128 (8-row tile at N = 2; 16-row: bench-scale test)
    dict(ks=3, stride=1, pad=1, cin=128, cout=128, h=20, w=24),     # Cin >= 128, ragged (8-row tile at N = 2)
    dict(ks=3, stride=2, pad=1, cin=64, cout=128, h=32, w=32),
    dict(ks=1, stride=1, pad=0, cin=128, cout=32, h=16, w=16),      # 32-channel tile
    dict(ks=3, stride=1, pad=1, cin=32, cout=16, h=16, w=20),       # 16-channel tile
])
def test_conv2d_fwd_compact_epilogue(ops, dtype, mma, cfg, want_stats, relu):
    """the host-selected compact-epilogue instantiation of conv_mfma_kernel (no pre-activation copy, no gating, no
    GELU, Cout a multiple of the 16-byte piece): bias, residual, optional ReLU, optional statistics, LDS-transposed
    16-byte stores -- every straight-line variant of it"""
    N = 2
    x = rnd((N, cfg["cin"], cfg["h"], cfg["w"]), dtype, 11)
    w = rnd((cfg["cout"], cfg["cin"], cfg["ks"], cfg["ks"]), dtype, 12, scale=(cfg["cin"] * cfg["ks"] ** 2) ** -0.5)
    b = rnd((cfg["cout"],), torch.float32, 13, 0.1)
    pre = F.conv2d(x, w, b, cfg["stride"], cfg["pad"])
    r = rnd(tuple(pre.shape), dtype, 14)
    want = pre + r
    if relu:
        want = F.relu(want)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    out = ops.conv2d(dev(nhwc(x), dtype), wp, cfg["cout"], cfg["ks"], cfg["stride"], cfg["pad"], bias=b.cuda(),
                     residual=dev(nhwc(r), dtype), act=ops.ACT_RELU if relu else ops.ACT_NONE, want_stats=want_stats)
    y = out[0] if want_stats else out
    close(nchw(y), want, dtype, "conv2d out (compact epilogue)")
    if want_stats:
        tot = out[1].sum(2).cpu()
        close(tot[0, :cfg["cout"]], want.sum((0, 2, 3)), dtype, "stats sum", scale=float(want.abs().sum((0, 2, 3)).max()))
        close(tot[1, :cfg["cout"]], (want * want).sum((0, 2, 3)), dtype, "stats sumsq")
    # and without bias / residual
    y2 = ops.conv2d(dev(nhwc(x), dtype), wp, cfg["cout"], cfg["ks"], cfg["stride"], cfg["pad"])
    close(nchw(y2), F.conv2d(x, w, None, cfg["stride"], cfg["pad"]), dtype, "conv2d out (plain)")


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(ks=3, stride=1, pad=1, cin=64, cout=64, h=24, w=24),
    dict(ks=3, stride=2, pad=1, cin=64, cout=128, h=32, w=32),
    dict(ks=1, stride=2, pad=0, cin=64, cout=128, h=32, w=32),
    dict(ks=1, stride=1, pad=0, cin=256, cout=32, h=16, w=16),
    dict(ks=3, stride=1, pad=1, cin=32, cout=2, h=16, w=16),
    dict(ks=3, stride=1, pad=2, dil=2, cin=64, cout=64, h=20, w=24),
    dict(ks=3, stride=1, pad=2, dil=2, cin=128, cout=128, h=8, w=8),      # dilation reaches past a whole 8x8 map edge
])
def test_conv2d_dgrad_and_wgrad(ops, dtype, mma, cfg):
    N = 2
    x = rnd((N, cfg["cin"], cfg["h"], cfg["w"]), dtype, 5).requires_grad_(True)
    w = rnd((cfg["cout"], cfg["cin"], cfg["ks"], cfg["ks"]), dtype, 6, scale=(cfg["cin"] * cfg["ks"] ** 2) ** -0.5)
    w.requires_grad_(True)
    dil = cfg.get("dil", 1)
    y = F.conv2d(x, w, None, cfg["stride"], cfg["pad"], dil)
    dy = rnd(tuple(y.shape), dtype, 7)
    y.backward(dy)
    ck = ops.chunk_channels(dtype)
    cout_k = -(-cfg["cout"] // ck) * ck            # reduction dim of the data gradient, padded
    _, wd = ops.pack_weight(w.detach().cuda(), dtype, want_dgrad=True, dgrad_inner=cout_k)
    dyp = torch.zeros(N, y.shape[2], y.shape[3], cout_k)
    dyp[..., :cfg["cout"]] = nhwc(dy)
    dyd = dev(dyp, dtype)
    if cfg["stride"] == 2:
        dyd = ops.zero_insert2(dyd, cfg["h"], cfg["w"])
    dx = ops.conv2d(dyd, wd, cfg["cin"], cfg["ks"], 1, dil * (cfg["ks"] - 1) - cfg["pad"], out_hw=(cfg["h"], cfg["w"]),
                    dilation=dil)
    close(nchw(dx), x.grad, dtype, "dgrad", factor=2.0)
    for tr in ([True, False] if dtype == torch.bfloat16 else [False]):
        dw = torch.full(tuple(w.shape), 0.5, device="cuda")
        ops.conv2d_wgrad(dev(nhwc(x.detach()), dtype), dev(nhwc(dy), dtype), dw, cfg["ks"], cfg["stride"], cfg["pad"],
                         accumulate=True, use_tr=tr, dilation=dil)
        close(dw - 0.5, w.grad, dtype, "wgrad tr=%s" % tr, factor=4.0)


@pytest.mark.parametrize("mma", [0, 1])
def test_weight_gradient_of_the_class_head_from_one_piece_per_pixel_dlogits(ops, mma):
    """fp32: the class head's dlogits are ONE 16-byte piece per pixel (4 floats, n_class = 2 real channels); its weight gradient
    under both fp32 matrix-product forms -- the split-bf16 form stages 8-channel pieces and zero-fills the half past the 4th
    channel (csrc/conv_wgrad.hip wg_body<f32x3>)"""
    N, H, W = 3, 40, 48
    x = rnd((N, 32, H, W), torch.float32, 31).requires_grad_(False)
    w = rnd((2, 32, 3, 3), torch.float32, 32, scale=288 ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = rnd(tuple(y.shape), torch.float32, 33)
    y.backward(dy)
    dy4 = torch.zeros(N, H, W, 4)
    dy4[..., :2] = nhwc(dy)
    dw = torch.zeros(2, 32, 3, 3, device="cuda")
    ops.conv2d_wgrad(nhwc(x).cuda(), dy4.cuda(), dw, 3, 1, 1, cout_real=2, defer=False)
    close(dw, w.grad, torch.float32, "class-head weight gradient (mma mode %d)" % mma, factor=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_stem_space_to_depth_conv(ops, dtype):
    N = 2
    x = rnd((N, 3, 64, 64), torch.float32, 8)
    w = rnd((64, 3, 7, 7), dtype, 9, scale=147 ** -0.5).requires_grad_(True)
    xq = x.to(dtype).float()
    y = F.conv2d(xq, w, None, 2, 3)
    dy = rnd(tuple(y.shape), dtype, 10)
    y.backward(dy)
    xs = ops.stem_space_to_depth(x.cuda(), dtype)
    wp = ops.stem_pack_weight(w.detach().cuda(), dtype)
    out = ops.conv2d(xs, wp, 64, ks=4, stride=1, pad=2, out_hw=(32, 32))
    close(nchw(out), y.detach(), dtype, "stem fwd")
    dw = torch.zeros(64, 3, 7, 7, device="cuda")
    ops.stem_wgrad(xs, dev(nhwc(dy), dtype), dw)
    close(dw, w.grad, dtype, "stem wgrad", factor=4.0)


@pytest.mark.parametrize("hw,B", [((64, 64), 2), ((40, 72), 1), ((256, 256), 3)])
def test_stem_direct_kernel_on_nchw_images(ops, hw, B):
    """dh_stem7_fwd (csrc/stem.hip): conv + per-tile BN statistics + the space-to-depth by-product the weight gradient reads,
    both streams in one launch; ragged tiles (40x72 -> 20x36 outputs); eval form with folded scale / bias / ReLU"""
    dtype = torch.bfloat16
    H, W = hw
    x1, x2 = rnd((B, 3, H, W), torch.float32, 8), rnd((B, 3, H, W), torch.float32, 18)
    w = rnd((64, 3, 7, 7), dtype, 9, scale=147 ** -0.5).requires_grad_(True)
    xq = torch.cat([x1, x2]).to(dtype).float()
    y = F.conv2d(xq, w, None, 2, 3)
    got, st, xs = ops.stem7_fwd(x1.cuda(), x2.cuda(), w.detach().cuda(), want_stats=True, want_xs=True, groups=2)
    assert got.shape == (2 * B, H // 2, W // 2, 64) and xs.shape == (2 * B, H // 2, W // 2, 16)
    close(nchw(got), y.detach(), dtype, "stem7 fwd")
    # statistics are those of the fp32 accumulators, per workgroup; the first half of the slots is stream 1 (group 0)
    yd = y.detach()
    half = st.shape[2] // 2
    for grp, sl in ((0, slice(0, half)), (1, slice(half, None))):
        tot, part = st[:, :, sl].sum(dim=2).cpu(), yd[grp * B:(grp + 1) * B]
        assert torch.allclose(tot[0], part.sum(dim=(0, 2, 3)), rtol=1e-2, atol=4e-4 * float(part.abs().sum(dim=(0, 2, 3)).max()))
        assert torch.allclose(tot[1], (part * part).sum(dim=(0, 2, 3)), rtol=2e-2)
    # the by-product equals the stand-alone space-to-depth kernel's first 16 channels, bit for bit
    ref = ops.stem_space_to_depth(torch.cat([x1, x2]).cuda(), dtype)
    assert torch.equal(xs, ref[..., :16].contiguous())
    dy = rnd(tuple(y.shape), dtype, 10)
    y.backward(dy)
    dw = torch.zeros(64, 3, 7, 7, device="cuda")
    ops.stem_wgrad(xs, dev(nhwc(dy), dtype), dw)
    close(dw, w.grad, dtype, "stem wgrad on the 16-channel by-product", factor=4.0)
    # one stream only, eval form
    sc, sh = rnd((64,), torch.float32, 3).abs() + 0.5, rnd((64,), torch.float32, 4)
    want = F.relu(F.conv2d(x1.to(dtype).float(), (w.detach() * sc.view(-1, 1, 1, 1)).to(dtype).float(), sh, 2, 3))
    got1 = ops.stem7_fwd(x1.cuda(), None, w.detach().cuda(), out_scale=sc.cuda(), bias=sh.cuda(), relu=True)[0]
    close(nchw(got1), want, dtype, "stem7 eval form")


@pytest.mark.parametrize("hw,B", [((32, 64), 1), ((128, 128), 2)])
def test_stem_tail_backward_without_a_batchnorm_pass(ops, hw, B):
    """dh_stem_pool_bn_bwd + dh_stem_wgrad_bn against autograd through conv7x7/2 -> BatchNorm (batch statistics, two groups)
    -> ReLU -> maxpool3x3/2: weight, gamma, beta gradients"""
    dtype = torch.bfloat16
    H, W = hw
    x1, x2 = rnd((B, 3, H, W), torch.float32, 8), rnd((B, 3, H, W), torch.float32, 18)
    w = rnd((64, 3, 7, 7), dtype, 9, scale=147 ** -0.5).requires_grad_(True)
    gamma = (rnd((64,), torch.float32, 3) * 0.1 + 1.0).requires_grad_(True)
    beta = (rnd((64,), torch.float32, 4) * 0.1).requires_grad_(True)
    y_dev, st, xs = ops.stem7_fwd(x1.cuda(), x2.cuda(), w.detach().cuda(), want_stats=True, want_xs=True, groups=2)
    oh, ow = H // 2, W // 2
    rm, rv = torch.zeros(64, device="cuda"), torch.ones(64, device="cuda")
    mean, invstd, scale, shift = ops.bn_finalize(st, 64, 2, B * oh * ow, gamma.detach().cuda(), beta.detach().cuda(), rm, rv)
    pooled, parg = ops.maxpool(y_dev, want_arg=True, bn=(scale, shift, 2))
    # reference: the same graph in fp32 on the bf16-rounded conv output the device holds (so masks / arg-max agree)
    yq = nchw(y_dev.float().cpu()).requires_grad_(True)
    outs = []
    for grp in range(2):
        part = yq[grp * B:(grp + 1) * B]
        outs.append(F.max_pool2d(F.relu(F.batch_norm(part, None, None, gamma, beta, True, 0.1, 1e-5)), 3, 2, 1))
    ref_pooled = torch.cat(outs)
    close(nchw(pooled), ref_pooled.detach(), dtype, "pooled")
    dpool = rnd(tuple(ref_pooled.shape), dtype, 10)
    ref_pooled.backward(dpool)
    dgamma, dbeta = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    d, coef = ops.stem_pool_bn_bwd(parg, dev(nhwc(dpool), dtype), y_dev, scale, shift, mean, invstd, gamma.detach().cuda(),
                                   dgamma, dbeta, 2)
    close(dgamma, gamma.grad, dtype, "dgamma", factor=2.0)
    close(dbeta, beta.grad, dtype, "dbeta", factor=2.0)
    # dL/dy = A * d + B * y + C per channel and group.  The arg-max of a near-tie may differ between the device's pool (bf16
    # BatchNorm statistics path) and torch's fp32 graph -- a handful of windows route their gradient to the neighbouring
    # pixel -- so the element-wise check allows 0.1 % outliers
    c = coef.view(2, 3, 1, 1, 1, 64).cpu()
    shp = (2, B, oh, ow, 64)
    dx = (c[:, 0] * d.float().cpu().view(shp) + c[:, 1] * y_dev.float().cpu().view(shp) + c[:, 2]).view(2 * B, oh, ow, 64)
    want = nhwc(yq.grad)
    off = ((dx - want).abs() > tol(dtype) * float(want.abs().max())).float().mean()
    assert float(off) < 1e-3, float(off)
    # the weight gradient with that expression applied on load == the two-pass form (pool backward, BatchNorm backward, wgrad)
    dw = torch.zeros(64, 3, 7, 7, device="cuda")
    ops.stem_wgrad(xs, d, dw, bn=(y_dev, coef, 2))
    dg2, db2 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dx2 = ops.bn_bwd(ops.maxpool_bwd(parg, dev(nhwc(dpool), dtype), tuple(y_dev.shape)), None, y_dev, mean, invstd,
                     gamma.detach().cuda(), dg2, db2, 2, mask_scale=scale, mask_shift=shift)
    close(dgamma, dg2.cpu(), dtype, "dgamma fused vs two-pass", factor=0.1)
    dw2 = torch.zeros(64, 3, 7, 7, device="cuda")
    ops.stem_wgrad(xs, dx2, dw2)
    close(dw, dw2.cpu(), dtype, "fused vs two-pass", factor=0.2)
    # and against autograd through the convolution on the exact dL/dy of the device (no arg-max ambiguity left)
    xq = torch.cat([x1, x2]).to(dtype).float()
    F.conv2d(xq, w, None, 2, 3).backward(nchw(dx2.float().cpu()))
    close(dw, w.grad, dtype, "stem wgrad with BatchNorm backward on load", factor=1.0)


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("ncls,hw,lazy", [(2, (64, 64), True), (5, (40, 56), False), (2, (256, 256), True)])
def test_class_head_writes_nchw_logits_itself(ops, dtype, mma, ncls, hw, lazy):
    """dh_conv3x3_head_fwd == dh_conv2d_fwd + dh_nhwc_to_nchw (fp32 NCHW logits), plain and with BatchNorm + ReLU on load"""
    H, W = hw
    N = 2
    ck = ops.chunk_channels(dtype)
    x = rnd((N, 32, H, W), dtype, 21)
    w = rnd((ncls, 32, 3, 3), dtype, 22, scale=288 ** -0.5)
    b = rnd((ncls,), torch.float32, 23)
    xd = dev(nhwc(x), dtype)
    if ck > 32:
        xd = torch.cat([xd, torch.zeros(N, H, W, ck - 32, dtype=xd.dtype, device="cuda")], dim=-1).contiguous()
    wp, _ = ops.pack_weight(torch.cat([w, torch.zeros(ncls, xd.shape[-1] - 32, 3, 3)], 1).cuda() if xd.shape[-1] > 32 else w.cuda(),
                            dtype, want_dgrad=False)
    src, ref_in = xd, x
    if lazy and dtype == torch.bfloat16:
        sc, sh = (rnd((2, xd.shape[-1]), torch.float32, 24).abs() + 0.5).cuda(), rnd((2, xd.shape[-1]), torch.float32, 25).cuda()
        src = ops.BnInput(xd, sc, sh, 2)
        g = torch.arange(N) // (N // 2)
        ref_in = F.relu(x * sc.cpu()[g][:, :32, None, None] + sh.cpu()[g][:, :32, None, None]).to(dtype).float()
    got = ops.conv3x3_head(src, wp, ncls, b.cuda())
    assert got.dtype == torch.float32 and tuple(got.shape) == (N, ncls, H, W)
    want = F.conv2d(ref_in, w, b, 1, 1)
    close(got, want, dtype, "class head")
    two = ops.nhwc_to_nchw(ops.conv2d(src, wp, ncls, 3, 1, 1, bias=b.cuda()))
    close(got, two.cpu(), dtype, "head vs conv + layout pass", factor=0.5)


@pytest.mark.parametrize("lazy", [False, True])
@pytest.mark.parametrize("cfg", [dict(n=2, h=20, w=36, ncls=2), dict(n=4, h=7, w=23, ncls=1), dict(n=2, h=1, w=5, ncls=2),
                                 dict(n=6, h=256, w=256, ncls=2), dict(n=2, h=33, w=270, ncls=2)])
@pytest.mark.parametrize("dtype,mma", [pytest.param(torch.bfloat16, 0, id="bfloat16"), pytest.param(torch.float32, 3, id="f16x3")])
def test_class_head_forward_as_one_product_per_input_pixel(ops, cfg, lazy, dtype, mma):
    """dh_head_fwd (P[pixel][tap][class] by one MFMA per 16 input pixels, the convolution as a nine-term gather from LDS rows)
    against F.conv2d in fp32 (models/help_funcs.py:13-14) and against dh_conv3x3_head_fwd, plain and with BatchNorm + ReLU on
    load (two statistics groups); ragged 16-pixel groups, strips of several rows per workgroup, fewer rows than a strip.
    fp32 tensors: the fp16-plane form of f32_mma_mode 3 at the fp32 tolerance"""
    N, H, W, ncls = cfg["n"], cfg["h"], cfg["w"], cfg["ncls"]
    assert ops._lib.lib().dh_head_fwd_supported(ncls, W) and not ops._lib.lib().dh_head_fwd_supported(3, W) \
        and not ops._lib.lib().dh_head_fwd_supported(2, 2048)
    x = rnd((N, 32, H, W), dtype, 2201)
    w = rnd((ncls, 32, 3, 3), torch.float32, 2202, scale=288 ** -0.5)
    b = rnd((ncls,), torch.float32, 2203)
    xd = dev(nhwc(x), dtype)
    ck = ops.chunk_channels(dtype)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    src, ref_in = xd, x
    if lazy:
        sc, sh = (rnd((2, 32), torch.float32, 2204).abs() + 0.5).cuda(), rnd((2, 32), torch.float32, 2205).cuda()
        src = ops.BnInput(xd, sc, sh, 2)
        g = torch.arange(N) // (N // 2)
        ref_in = F.relu(x * sc.cpu()[g][:, :, None, None] + sh.cpu()[g][:, :, None, None]).to(dtype).float()
    got = ops.conv3x3_head(src, wp, ncls, b.cuda(), w_oihw=w.cuda())
    assert got.dtype == torch.float32 and tuple(got.shape) == (N, ncls, H, W)
    close(got, F.conv2d(ref_in, w.to(dtype).float(), b, 1, 1), dtype, "class head vs conv2d", factor=0.25)
    if ck <= 32:
        old = ops.conv3x3_head(src, wp, ncls, b.cuda())
        close(got, old.cpu(), dtype, "class head vs the tile convolution", factor=0.01 if dtype == torch.bfloat16 else 1.0)      # bf16: same operands, fp32 sums re-ordered


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_weight_repack_of_a_step_equals_the_permutation_it_states(ops, dtype):
    """ops.PackPlan (dh_pack_weights_multi, one launch per step for every layer): forward form [tap][OPad][I] = w[o][ci][tap],
    data-gradient form [tap][IPad][OK] = w[o][ci][taps - 1 - tap], zero padding; shapes with and without the 8-element
    vector path (I or OK not a multiple of 8), padded rows, a 1x1 and a linear layer, several jobs in one plan"""
    shapes = [(64, 64, 3), (2, 32, 3), (40, 24, 3), (32, 12, 3), (128, 64, 1), (20, 36, 1), (64, 32, 0)]      # O, I, kernel (0: linear)
    plan = ops.PackPlan(torch.device("cuda"))
    jobs = []
    for k, (O, I, ks) in enumerate(shapes):
        w = rnd((O, I, ks, ks) if ks else (O, I), dtype, 4100 + k).cuda().float()
        inner = 48 if (O, I) == (40, 24) else 0
        jobs.append((w, inner) + plan.add(w, dtype, dgrad_inner=inner))
    plan.run()
    torch.cuda.synchronize()
    for w, inner, fwd, dg in jobs:
        O, I = w.shape[:2]
        taps = w[0, 0].numel() if w.dim() == 4 else 1
        w3 = w.reshape(O, I, taps)
        OPad, IPad, OK = fwd.shape[1], dg.shape[1], dg.shape[2]
        assert OK == max(O, inner) and OPad % 16 == 0 and IPad % 16 == 0
        want_f = torch.zeros(taps, OPad, I, device="cuda")
        want_f[:, :O] = w3.permute(2, 0, 1)
        want_d = torch.zeros(taps, IPad, OK, device="cuda")
        want_d[:, :I, :O] = w3.flip(2).permute(2, 1, 0)
        assert torch.equal(fwd.float(), want_f.to(dtype).float()), (O, I, taps)
        assert torch.equal(dg.float(), want_d.to(dtype).float()), (O, I, taps)


@pytest.mark.parametrize("cfg", [(64, 64, 64, 64, 64), (64, 32, 32, 256, 256), (64, 32, 32, 128, 256), (64, 32, 32, 256, 128)])
def test_fragment_order_packed_weights(ops, cfg):
    """dh_pack_weights_multi with dtype | 0x200: the fragment-order copies hold the same numbers as the row-major packs
    ([r][c][tap][lane][e] = row 16 r + (lane & 15), reduction channel 32 c + 8 (lane >> 4) + e), and dh_conv2d_fwd(w_frag) on
    the register-resident-weights kernel gives the bit-identical forward / data gradient"""
    from dahitra_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    N, H, W, Cin, Cout = cfg
    x = rnd((N, Cin, H, W), dtype, 31)
    w = rnd((Cout, Cin, 3, 3), dtype, 32, scale=(9 * Cin) ** -0.5)
    plan = ops.PackPlan(torch.device("cuda"))
    wd = w.cuda()
    ff, df = plan.add(wd, dtype, dgrad_inner=Cout, frag=True)
    plan.run()
    frm, drm = ops.pack_weight(wd, dtype, dgrad_inner=Cout)
    for frag, row in ((ff, frm), (df, drm)):
        R, K = row.shape[1], row.shape[2]
        assert tuple(frag.shape) == (R // 16, K // 32, 9, 64, 8)
        # [r, c, tap, g, pl, e] -> [tap, r, pl, c, g, e]
        back = frag.view(R // 16, K // 32, 9, 4, 16, 8).permute(2, 0, 4, 1, 3, 5).reshape(9, R, K)
        assert torch.equal(back, row)
    xd = dev(nhwc(x), dtype)
    prev = L.dh_conv_wreg_mode(1)
    try:
        y_f = ops.conv2d(xd, frm, Cout, 3, 1, 1, w_frag=ff)
        y_r = ops.conv2d(xd, frm, Cout, 3, 1, 1)
        dy = rnd(tuple(nchw(y_f).shape), dtype, 33)
        dyd = dev(nhwc(dy), dtype)
        g_f = ops.conv2d(dyd, drm, Cin, 3, 1, 1, w_frag=df)
        g_r = ops.conv2d(dyd, drm, Cin, 3, 1, 1)
    finally:
        L.dh_conv_wreg_mode(prev)
    assert torch.equal(y_f, y_r) and torch.equal(g_f, g_r)
    close(nchw(y_f), F.conv2d(x, w, None, 1, 1), dtype, "conv, fragment-order weights")


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
def test_linear_rows_gelu_and_per_image_weights(ops, dtype, mma):
    rows, cin, cout = 40, 32, 64        # rows not a multiple of 16 -> masked tail
    x = rnd((rows, cin), dtype, 11)
    w = rnd((cout, cin), dtype, 12, cin ** -0.5).requires_grad_(True)
    b = rnd((cout,), torch.float32, 13, 0.1)
    xr = x.clone().requires_grad_(True)
    z = F.linear(xr, w, b)
    h = F.gelu(z)
    dh = rnd((rows, cout), dtype, 14)
    h.backward(dh)
    wp, wd = ops.pack_weight(w.detach().cuda(), dtype)
    y, pre = ops.linear(dev(x, dtype), wp, cout, bias=b.cuda(), act=ops.ACT_GELU, want_preact=True)
    close(y, h.detach(), dtype, "linear+gelu")
    close(pre, z.detach(), dtype, "linear preact")
    dz = ops.act_bwd(dev(dh, dtype), pre, ops.ACT_GELU)
    dx = ops.linear(dz, wd, cin)
    close(dx, xr.grad, dtype, "linear dgrad", factor=3.0)
    dw = torch.zeros(cout, cin, device="cuda")
    ops.linear_wgrad(dev(x, dtype), dz, dw)
    close(dw, w.grad, dtype, "linear wgrad", factor=4.0)
    # per-image weights: 3 images x 32 rows, each with its own [cout=32][cin=32] matrix
    xi = rnd((96, 32), dtype, 15)
    wi = rnd((3, 32, 32), dtype, 16, 32 ** -0.5)
    want = torch.cat([xi[i * 32:(i + 1) * 32] @ wi[i].t() for i in range(3)])
    got = ops.linear(dev(xi, dtype), dev(wi, dtype), 32, images=3, w_image_stride=32 * 32)
    close(got, want, dtype, "per-image linear")
    dyi = rnd((96, 32), dtype, 17)
    dwi = torch.zeros(3, 32, 32, device="cuda")
    ops.linear_wgrad(dev(xi, dtype), dev(dyi, dtype), dwi, images=3, per_image=True)
    want_dw = torch.stack([dyi[i * 32:(i + 1) * 32].t() @ xi[i * 32:(i + 1) * 32] for i in range(3)])
    close(dwi, want_dw, dtype, "per-image wgrad", factor=4.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm_train_two_streams_and_backward(ops, dtype):
    B, C, H, W = 3, 64, 24, 24
    xs = [(rnd((B, C, H, W), torch.float32, 20 + i, 1.5) + 0.3).to(dtype).float() for i in range(2)]
    res = [rnd((B, C, H, W), dtype, 30 + i) for i in range(2)]
    gamma = (1 + 0.1 * rnd((C,), torch.float32, 22)).requires_grad_(True)
    beta = (0.1 * rnd((C,), torch.float32, 23)).requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    outs, xr = [], []
    for i in range(2):          # the reference runs the two streams one after the other
        xi = xs[i].clone().requires_grad_(True)
        xr.append(xi)
        outs.append(F.relu(F.batch_norm(xi, rm, rv, gamma, beta, True, 0.1, 1e-5) + res[i]))
    dout = [rnd((B, C, H, W), dtype, 40 + i) for i in range(2)]
    (outs[0] * dout[0]).sum().backward()
    (outs[1] * dout[1]).sum().backward()
    # device: conv (identity 1x1) is not needed -- feed statistics from a stats-producing conv call
    xcat = dev(nhwc(torch.cat(xs)), dtype)
    eye = torch.eye(C).reshape(C, C, 1, 1)
    wp, _ = ops.pack_weight(eye.cuda(), dtype, want_dgrad=False)
    xconv, stats = ops.conv2d(xcat, wp, C, 1, 1, 0, want_stats=True)
    g, b = gamma.detach().cuda(), beta.detach().cuda()
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    mean, invstd, scale, shift = ops.bn_finalize(stats, C, 2, B * H * W, g, b, rmd, rvd)
    rcat = dev(nhwc(torch.cat(res)), dtype)
    y = ops.bn_apply(xconv, scale, shift, groups=2, act=ops.ACT_RELU, residual=rcat)
    close(nchw(y), torch.cat(outs).detach(), dtype, "bn fwd")
    # (in bf16 mode the device input is the bf16 rounding of xs, so the statistics move by ~2^-9)
    close(rmd, rm, dtype, "running_mean", factor=5 if dtype == torch.float32 else 0.5)
    close(rvd, rv, dtype, "running_var", factor=5 if dtype == torch.float32 else 0.5)
    dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dx, dres = ops.bn_bwd(dev(nhwc(torch.cat(dout)), dtype), y, xconv, mean, invstd, g, dg, db, groups=2,
                          want_dres=True)
    close(nchw(dx), torch.cat([xr[0].grad, xr[1].grad]), dtype, "bn dx", factor=4)
    close(dg, gamma.grad, dtype, "bn dgamma", factor=8)
    close(db, beta.grad, dtype, "bn dbeta", factor=8)
    mask = (torch.cat(outs) > 0).float()
    close(nchw(dres), torch.cat(dout) * mask, dtype, "bn dres")


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm(ops, dtype):
    rows = 1000
    x = (rnd((rows, 32), dtype, 50, 2.0) + 0.5).requires_grad_(True)
    g = (1 + 0.1 * rnd((32,), torch.float32, 51)).requires_grad_(True)
    b = (0.1 * rnd((32,), torch.float32, 52)).requires_grad_(True)
    y = F.layer_norm(x, (32,), g, b, 1e-5)
    dy = rnd((rows, 32), dtype, 53)
    extra = rnd((rows, 32), dtype, 54)
    y.backward(dy)
    yd, st = ops.layernorm(dev(x.detach(), dtype), g.detach().cuda(), b.detach().cuda())
    close(yd, y.detach(), dtype, "ln fwd")
    dg, dbb = torch.ones(32, device="cuda"), torch.ones(32, device="cuda")
    dx = ops.layernorm_bwd(dev(dy, dtype), dev(x.detach(), dtype), st, g.detach().cuda(), dg, dbb,
                           dx_add=dev(extra, dtype), accumulate=True)
    close(dx, x.grad + extra, dtype, "ln dx", factor=3)
    close(dg - 1, g.grad, dtype, "ln dgamma", factor=8)
    close(dbb - 1, b.grad, dtype, "ln dbeta", factor=8)


@pytest.mark.parametrize("dtype", DTYPES)
def test_pool_and_resample(ops, dtype):
    N, C, H, W = 2, 32, 20, 28
    x = F.relu(rnd((N, C, H, W), dtype, 60)).requires_grad_(True)     # many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = rnd(tuple(y.shape), dtype, 61)
    y.backward(dy)
    xd = dev(nhwc(x.detach()), dtype)
    yd, arg = ops.maxpool(xd, want_arg=True)
    close(nchw(yd), y.detach(), dtype, "maxpool")
    close(nchw(ops.maxpool_bwd(arg, dev(nhwc(dy), dtype), xd.shape)), x.grad, dtype, "maxpool bwd", factor=2)
    x2 = rnd((N, C, 10, 12), dtype, 62).requires_grad_(True)
    u = F.interpolate(x2, scale_factor=2, mode="nearest")
    du = rnd(tuple(u.shape), dtype, 63)
    u.backward(du)
    close(nchw(ops.upsample2(dev(nhwc(x2.detach()), dtype))), u.detach(), dtype, "up2")
    close(nchw(ops.upsample2_bwd(dev(nhwc(du), dtype))), x2.grad, dtype, "up2 bwd", factor=2)
    a = rnd((N, C, 8, 12), dtype, 64).requires_grad_(True)
    b = rnd((N, C, 8, 12), dtype, 65).requires_grad_(True)
    o = F.interpolate(torch.abs(a - b), scale_factor=4, mode="bilinear", align_corners=False)
    do = rnd(tuple(o.shape), dtype, 66)
    o.backward(do)
    ad, bd = dev(nhwc(a.detach()), dtype), dev(nhwc(b.detach()), dtype)
    close(nchw(ops.absdiff_upsample4(ad, bd)), o.detach(), dtype, "absdiff+bilinear")
    da, db = ops.absdiff_upsample4_bwd(ad, bd, dev(nhwc(do), dtype))
    close(nchw(da), a.grad, dtype, "bilinear bwd a", factor=4)
    close(nchw(db), b.grad, dtype, "bilinear bwd b", factor=4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L", [4, 8])
def test_tokenizer(ops, dtype, L):
    B, H, W = 2, 16, 16
    Sn = 2 * B
    x = rnd((Sn, 32, H, W), dtype, 70).requires_grad_(True)
    wa = rnd((L, 32, 1, 1), torch.float32, 71, 0.5).requires_grad_(True)
    pos = rnd((1, 2 * L, 32), torch.float32, 72, 0.5).requires_grad_(True)
    att = torch.softmax(F.conv2d(x, wa).reshape(Sn, L, -1), -1)
    tk = att @ x.reshape(Sn, 32, -1).transpose(1, 2)
    cat = torch.cat([tk[:B], tk[B:]], 1) + pos
    dt = rnd((B, 2 * L, 32), dtype, 73)
    cat.backward(dt)
    xd = dev(nhwc(x.detach()), dtype)
    tok, saved = ops.tokenizer_fwd(xd, wa.detach().reshape(L, 32).cuda(), pos.detach().reshape(2 * L, 32).cuda(), B, L)
    close(tok, cat.detach(), dtype, "tokens")
    dx = torch.zeros_like(xd)
    dwa = torch.zeros(L, 32, device="cuda")
    dpos = torch.zeros(2 * L, 32, device="cuda")
    ops.tokenizer_bwd(xd, wa.detach().reshape(L, 32).cuda(), saved, dt.float().cuda(), dx, dwa, dpos, B, L)   # tokens: fp32
    close(nchw(dx), x.grad, dtype, "tokenizer dx", factor=4)
    close(dwa, wa.grad.reshape(L, 32), dtype, "tokenizer dwa", factor=8)
    close(dpos, pos.grad.reshape(2 * L, 32), dtype, "dpos", factor=4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_self_attention_core_and_grouped_softmax(ops, dtype):
    B, n, heads, dh = 3, 8, 8, 64
    qkv = rnd((B * n, 3 * heads * dh), dtype, 80).requires_grad_(True)
    q, k, v = qkv.reshape(B, n, 3, heads, dh).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B * n, heads * dh)
    do = rnd((B * n, heads * dh), dtype, 81)
    o.backward(do)
    qd = dev(qkv.detach(), dtype)
    od, attn = ops.self_attn(qd, B, n, heads, dh)
    close(od, o.detach(), dtype, "self-attn out")
    close(ops.self_attn_bwd(qd, attn, dev(do, dtype), B, n, heads, dh), qkv.grad, dtype, "self-attn dqkv", factor=4)
    rows, H, L, HLP = 300, 4, 4, 32
    z = rnd((rows, HLP), dtype, 82).requires_grad_(True)
    p = torch.softmax(z[:, :H * L].reshape(rows, H, L), -1).reshape(rows, H * L)
    dp = rnd((rows, HLP), dtype, 83)
    p.backward(dp[:, :H * L])
    pd = ops.softmax_groups(dev(z.detach(), dtype), H, L)
    close(pd[:, :H * L], p.detach(), dtype, "grouped softmax")
    assert float(pd[:, H * L:].float().abs().max()) == 0.0
    dz = ops.softmax_groups_bwd(pd, dev(dp, dtype), H, L)
    close(dz[:, :H * L], z.grad[:, :H * L], dtype, "grouped softmax bwd", factor=4)


def test_focal_argmax_adamw(ops):
    import cdnet_ref as O
    B, C, H, W = 3, 2, 32, 32
    for C in (2, 5):
        logits = rnd((B, C, H, W), torch.float32, 90, 2.0).requires_grad_(True)
        tgt = torch.randint(0, C, (B, 1, H, W), generator=torch.Generator().manual_seed(91))
        loss = O.focal_loss(logits, tgt)
        loss.backward()
        l, dl = ops.focal_loss(logits.detach().cuda(), tgt[:, 0].contiguous().cuda())
        assert abs(float(l) - float(loss)) < 1e-6
        close(dl, logits.grad, torch.float32, "focal grad", factor=2)
        m = ops.argmax_nchw(logits.detach().cuda())
        assert torch.equal(m.cpu(), torch.argmax(logits.detach(), 1))
    n = 10007
    p = rnd((n,), torch.float32, 92)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=0.01, betas=(0.9, 0.999), weight_decay=0.01)
    pd, m, v = p.cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        g = rnd((n,), torch.float32, 92 + step)
        ref.grad = g.clone()
        opt.step()
        ops.adamw_step(pd, g.cuda(), m, v, 0.01, 0.9, 0.999, 1e-8, 0.01, step)
        close(pd, ref.detach(), torch.float32, "adamw step %d" % step, factor=1)


def test_cross_entropy_branch_and_dice_constant(ops):
    """batch-size-1 branch (models/losses.py:9-26) and the gradient-free dice term (losses.py:333-339)"""
    import cdnet_ref as O
    from dahitra_amd.models import losses
    g = torch.Generator().manual_seed(17)
    for C in (2, 5):
        logits = (torch.randn(1, C, 40, 56, generator=g) * 2).requires_grad_(True)
        tgt = torch.randint(0, C, (1, 1, 40, 56), generator=g)
        tgt[0, 0, :3] = 255                                  # ignore_index rows
        want = O.cross_entropy(logits, tgt)
        (want * 0.7).backward()
        lg = logits.detach().cuda().requires_grad_(True)
        got = losses.cross_entropy(lg, tgt.cuda())
        (got * 0.7).backward()
        assert abs(float(got) - float(want)) <= 2e-6 * abs(float(want))
        assert float((lg.grad.cpu() - logits.grad).abs().max()) <= 1e-6 * float(logits.grad.abs().max()) + 1e-12
        assert float(lg.grad[0, :, :3].abs().max()) == 0.0
    for C, bs in ((2, 3), (5, 2)):
        logits = torch.randn(bs, C, 32, 32, generator=g)
        tgt = (torch.rand(bs, 1, 32, 32, generator=g) > 0.7).long()
        want = O.dice_constant(logits, tgt)
        got = losses.diceloss(logits.cuda(), tgt.cuda())
        assert abs(float(got) - float(want)) <= 1e-6, (float(got), float(want))
        assert not got.requires_grad
    empty = torch.zeros(2, 1, 32, 32, dtype=torch.long)
    assert float(losses.diceloss(torch.randn(2, 2, 32, 32, generator=g).cuda(), empty.cuda())) == 0.0   # empty-target mask


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("relu", [True, False])
def test_bn_backward_gated_in_the_dgrad_epilogue_equals_two_pass(ops, dtype, mma, relu):
    """conv2d(..., gate=...) + bn_bwd_from_partials against the separate (conv2d -> bn_bwd) passes, two BN groups"""
    N, H, W, Cin, Cout = 4, 24, 40, 64, 64           # the data-gradient conv maps Cout -> Cin; BN layer has Cin channels
    dy = dev(rnd((N, H, W, Cout), dtype, 201), dtype)
    w = rnd((Cin, Cout, 3, 3), dtype, 202, (Cout * 9) ** -0.5)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    res = dev(rnd((N, H, W, Cin), dtype, 203), dtype)
    y_pre = dev(rnd((N, H, W, Cin), dtype, 204, 1.5), dtype)
    groups = 2
    mean = rnd((groups, Cin), torch.float32, 205, 0.2).cuda()
    invstd = (rnd((groups, Cin), torch.float32, 206, 0.1) + 0.8).cuda()
    gamma = (rnd((Cin,), torch.float32, 207, 0.1) + 1.0).cuda()
    out = torch.relu(dev(rnd((N, H, W, Cin), dtype, 208), dtype)) if relu else None
    # two-pass reference path
    dout = ops.conv2d(dy, wp, Cin, 3, 1, 1, residual=res)
    dg0, db0 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    dx0, dres0 = ops.bn_bwd(dout, out, y_pre, mean, invstd, gamma, dg0, db0, groups, accumulate=False, want_dres=True)
    # gated path
    g, part = ops.conv2d(dy, wp, Cin, 3, 1, 1, residual=res, gate=(out, y_pre, mean, invstd, groups))
    dg1, db1 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    dx1 = ops.bn_bwd_from_partials(g, y_pre, part, mean, invstd, gamma, dg1, db1, groups, accumulate=False)
    close(g, dres0.float().cpu(), dtype, "gated g vs dres")
    close(dg1, dg0.cpu(), dtype, "dgamma", factor=4.0)
    close(db1, db0.cpu(), dtype, "dbeta", factor=4.0)
    close(dx1, dx0.float().cpu(), dtype, "dx", factor=4.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ncls", [2, 5])
def test_head_data_gradient_kernel(ops, dtype, ncls):
    """dh_head_dgrad3x3 (compact dlogits, one 16-byte piece per pixel) against autograd of F.conv2d(32 -> n_class)"""
    N, H, W = 2, 20, 36
    x = rnd((N, 32, H, W), dtype, 301).requires_grad_(True)
    w = rnd((ncls, 32, 3, 3), torch.float32, 302, 0.1)
    dy = rnd((N, ncls, H, W), dtype, 303)
    F.conv2d(x, w, None, 1, 1).backward(dy)
    cp = 8 if (dtype == torch.bfloat16 or ncls > 4) else 4
    dl = ops.nchw_to_nhwc(dy.float().cuda().contiguous(), dtype, cpad=cp)
    assert dl.shape[-1] == cp
    dx = ops.head_dgrad3x3(dl, w.cuda(), ncls)
    close(nchw(dx), x.grad, dtype, "head dgrad", factor=2.0)


@pytest.mark.parametrize("cfg", [
    dict(n=2, h=20, w=36, groups=1, ncls=2),
    dict(n=4, h=17, w=23, groups=2, ncls=2),         # ragged 16-pixel groups, two statistics groups
    dict(n=1, h=3, w=5, groups=1, ncls=1),           # fewer pixels than one wave round
    dict(n=8, h=128, w=128, groups=1, ncls=2),       # more 16-pixel groups than waves: the two-group rounds wrap
    dict(n=2, h=20, w=36, groups=1, ncls=5),         # 3 .. 8 classes: the 8-class piece form (k = tap * 8 + class, three K steps)
    dict(n=4, h=17, w=23, groups=2, ncls=8),
    dict(n=2, h=64, w=64, groups=1, ncls=3),
])
@pytest.mark.parametrize("dtype,mma", [pytest.param(torch.bfloat16, 0, id="bfloat16"), pytest.param(torch.float32, 1, id="bf16x3")])
def test_head_gradient_recomputed_inside_the_batchnorm_backward(ops, cfg, dtype, mma):
    """dh_head_bn_bwd (the class head's data gradient formed by both passes of the BatchNorm + ReLU backward behind it, never
    written) against (a) torch autograd of conv2(relu(batch_norm(y))) in fp32 (models/help_funcs.py:7-15) and (b) the
    three-kernel path dh_head_dgrad3x3 -> dh_bn_bwd, whose intermediate is rounded to bf16; fp32 tensors: the split-product
    form (three bf16 products per matrix product, the bf16x3 mode's backward arithmetic) at the fp32 tolerance"""
    N, H, W, G, ncls = cfg["n"], cfg["h"], cfg["w"], cfg["groups"], cfg["ncls"]
    y = rnd((N, 32, H, W), dtype, 2101, 1.5)
    w = rnd((ncls, 32, 3, 3), torch.float32, 2102, 0.1).requires_grad_(True)
    hb = torch.zeros(ncls, requires_grad=True)
    dlog = rnd((N, ncls, H, W), dtype, 2103)
    gamma = (rnd((32,), torch.float32, 2104, 0.2) + 1.0)
    beta = rnd((32,), torch.float32, 2105, 0.3)
    # torch: per statistics group (the reference's two forward_single calls), train-mode BatchNorm
    yt = y.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True)
    bt = beta.clone().requires_grad_(True)
    outs = []
    for k in range(G):
        sl = slice(k * N // G, (k + 1) * N // G)
        outs.append(F.conv2d(torch.relu(F.batch_norm(yt[sl], None, None, gm, bt, True, 0.1, 1e-5)), w, hb, 1, 1))
    torch.cat(outs).backward(dlog.float())
    w, wg = w.detach(), w.grad
    # device: statistics as dh_bn_finalize leaves them
    yd = dev(nhwc(y), dtype)
    yg = yd.float().view(G, -1, 32)
    mean = yg.mean(1)
    var = yg.var(1, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale = gamma.cuda()[None] * invstd
    shift = beta.cuda()[None] - mean * scale
    mean, invstd, scale, shift = (t.contiguous() for t in (mean, invstd, scale, shift))
    dl = ops.nchw_to_nhwc(dlog.float().cuda().contiguous(), dtype, cpad=8)
    dlp = ops.head_dlogits_pack(dlog.float().cuda().contiguous(), dtype)
    assert int(dlp[:, 0].abs().max()) == 0 and int(dlp[:, :, -1].abs().max()) == 0          # the border of zeros
    pc = 2 if ncls <= 2 else 8          # classes per piece
    if dtype == torch.bfloat16:
        assert torch.equal(dlp[:, 1:-1, 1:-1].contiguous().view(torch.bfloat16).view(N, H, W, pc)[..., :ncls], dl[..., :ncls])
    else:       # heads + remainders = the fp32 values to 2^-17
        pl = dlp[:, 1:-1, 1:-1].contiguous().view(torch.bfloat16).view(N, H, W, 2, pc).float().sum(3)
        assert float((pl[..., :ncls] - dl[..., :ncls]).abs().max()) <= 2.0 ** -16 * float(dl.abs().max())
    dg1, db1 = torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    dx1 = ops.head_bn_bwd(dlp, w.cuda(), ncls, yd, scale, shift, mean, invstd, gamma.cuda(), dg1, db1, G, accumulate=False)
    close(nchw(dx1), yt.grad, dtype, "dx vs autograd", factor=2.0)
    close(dg1, gm.grad, dtype, "dgamma vs autograd", factor=4.0)
    close(db1, bt.grad, dtype, "dbeta vs autograd", factor=4.0)
    # the three-kernel path
    g0 = ops.head_dgrad3x3(dl, w.cuda(), ncls)
    dg0, db0 = torch.zeros(32, device="cuda"), torch.zeros(32, device="cuda")
    dx0 = ops.bn_bwd(g0, None, yd, mean, invstd, gamma.cuda(), dg0, db0, G, accumulate=False, mask_scale=scale, mask_shift=shift)
    close(dx1, dx0.float().cpu(), dtype, "dx vs three kernels", factor=2.0)
    close(dg1, dg0.cpu(), dtype, "dgamma vs three kernels", factor=2.0)
    close(db1, db0.cpu(), dtype, "dbeta vs three kernels", factor=2.0)
    # accumulate: (+)= into the parameter gradients; with the head convolution's own weight / bias gradient from the same pass
    dw = torch.full((ncls, 32, 3, 3), 0.5, device="cuda")
    dbias = torch.full((ncls,), -2.0, device="cuda")
    dx2 = ops.head_bn_bwd(dlp, w.cuda(), ncls, yd, scale, shift, mean, invstd, gamma.cuda(), dg1, db1, G, accumulate=True,
                          dw=dw, db=dbias)
    assert torch.equal(dx2, dx1)
    close(dg1, 2 * gm.grad, dtype, "dgamma accumulated", factor=4.0)
    close(dw - 0.5, wg, dtype, "head weight gradient vs autograd", factor=2.0)
    close(dbias + 2.0, hb.grad, dtype, "head bias gradient vs autograd", factor=2.0)
    dw2, dbias2 = torch.zeros_like(dw), torch.zeros_like(dbias)
    ops.head_bn_bwd(dlp, w.cuda(), ncls, yd, scale, shift, mean, invstd, gamma.cuda(), dg1, db1, G, accumulate=False, dw=dw2, db=dbias2)
    # ... against the weight-gradient kernel it replaces (input = the BatchNorm + ReLU applied on load)
    dw0 = torch.zeros_like(dw)
    ops.conv2d_wgrad(ops.BnInput(yd, scale, shift, G), dl, dw0, 3, 1, 1, accumulate=False, cout_real=ncls)
    close(dw2, dw0.cpu(), dtype, "head weight gradient vs conv2d_wgrad", factor=1.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_bn_backward_mask_recomputed_from_input(ops, dtype):
    """ReLU mask as x * scale + shift > 0 (layers without a residual) == mask from the stored post-ReLU output"""
    N, H, W, C, groups = 4, 12, 20, 64, 2
    x = dev(rnd((N, H, W, C), dtype, 401, 1.5), dtype)
    scale = (rnd((groups, C), torch.float32, 402, 0.2) + 1.0).cuda()
    shift = rnd((groups, C), torch.float32, 403, 0.5).cuda()
    out = ops.bn_apply(x, scale, shift, groups, ops.ACT_RELU)
    dout = dev(rnd((N, H, W, C), dtype, 404), dtype)
    mean = rnd((groups, C), torch.float32, 405, 0.2).cuda()
    invstd = (rnd((groups, C), torch.float32, 406, 0.1) + 0.9).cuda()
    gamma = (rnd((C,), torch.float32, 407, 0.1) + 1.0).cuda()
    r = []
    for kw in (dict(out=out), dict(out=None, mask_scale=scale, mask_shift=shift)):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        o = kw.pop("out")
        dx = ops.bn_bwd(dout, o, x, mean, invstd, gamma, dg, db, groups, accumulate=False, **kw)
        r.append((dx.float(), dg, db))
    assert float((out > 0).float().mean()) > 0.2 and float((out == 0).float().mean()) > 0.2
    assert torch.equal(r[0][0], r[1][0]) and torch.equal(r[0][1], r[1][1]) and torch.equal(r[0][2], r[1][2])


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool_with_batchnorm_relu_on_load(ops, dtype):
    """maxpool(relu(x * scale + shift)) fused == bn_apply followed by maxpool (values and arg-max, two BN groups)"""
    N, H, W, C, groups = 4, 18, 22, 64, 2
    x = dev(rnd((N, H, W, C), dtype, 501, 1.5), dtype)
    scale = (rnd((groups, C), torch.float32, 502, 0.5) + 0.6).cuda()      # some negative scales too
    shift = rnd((groups, C), torch.float32, 503, 0.5).cuda()
    h = ops.bn_apply(x, scale, shift, groups, ops.ACT_RELU)
    y0, a0 = ops.maxpool(h, want_arg=True)
    y1, a1 = ops.maxpool(x, want_arg=True, bn=(scale, shift, groups))
    assert torch.equal(y0, y1) and torch.equal(a0, a1)


# ---- the instantiations bench.py's configs[1] step actually launches (2B = 64 images per launch) -----------------
# pick_rw (csrc/conv_mfma_impl.h) selects the 16x16-pixel tile (RW = 4) only when N * ceil(OH/16) * ceil(OW/16) >= 256
# and Cin >= 128: the small-N cases above never reach it.  These cases do, and assert that they do.
BENCH_N = 64


def _is_rw4(ops, N, OH, OW, cin, ks=3, stride=1, dtype=torch.bfloat16):
    from dahitra_amd import _lib
    nt = _lib.lib().dh_conv2d_fwd_num_tiles(0 if dtype == torch.float32 else 1, N, OH, OW, cin, ks, stride)
    return nt == N * ops.cdiv(OH, 16) * ops.cdiv(OW, 16) and nt != N * ops.cdiv(OH, 8) * ops.cdiv(OW, 16)


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(cin=128, cout=128, h=32, w=32, stats=True, relu=False, res=False),     # layer2 conv (+BN statistics)
    dict(cin=128, cout=256, h=32, w=32, stats=True, relu=False, res=False),     # layer3.0.conv1
    dict(cin=256, cout=256, h=32, w=32, stats=True, relu=False, res=False),     # layer3 convs: <T,3,1,64,4,1,true,true>
    dict(cin=256, cout=256, h=32, w=32, stats=False, relu=True, res=True),      # eval-mode form: bias + residual + ReLU
    dict(cin=256, cout=128, h=32, w=32, stats=False, relu=False, res=True),     # layer3.0.conv1 data gradient (+ residual)
    dict(cin=256, cout=32, h=64, w=64, stats=False, relu=False, res=False),     # conv_pred: <T,3,1,32,4,1,true,true>
    dict(cin=256, cout=256, h=32, w=32, dil=2, stats=True, relu=False, res=False),   # ResNet-50 layer3 3x3, 16-row tile
    dict(cin=128, cout=128, h=24, w=40, n=96, stats=True, relu=True, res=True),      # 16-row tiles, ragged map
])
def test_conv2d_fwd_bench_scale_16row_tiles(ops, dtype, mma, cfg):
    N = cfg.get("n", BENCH_N)
    dil = cfg.get("dil", 1)
    assert _is_rw4(ops, N, cfg["h"], cfg["w"], cfg["cin"], dtype=dtype), "case does not select the 16-row tile"
    x = rnd((N, cfg["cin"], cfg["h"], cfg["w"]), dtype, 601)
    w = rnd((cfg["cout"], cfg["cin"], 3, 3), dtype, 602, scale=(cfg["cin"] * 9) ** -0.5)
    b = rnd((cfg["cout"],), torch.float32, 603, 0.1)
    want = F.conv2d(x, w, b, 1, dil, dil)
    r = rnd(tuple(want.shape), dtype, 604) if cfg["res"] else None
    if r is not None:
        want = want + r
    if cfg["relu"]:
        want = F.relu(want)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    out = ops.conv2d(dev(nhwc(x), dtype), wp, cfg["cout"], 3, 1, dil, bias=b.cuda(),
                     residual=dev(nhwc(r), dtype) if r is not None else None,
                     act=ops.ACT_RELU if cfg["relu"] else ops.ACT_NONE, want_stats=cfg["stats"], dilation=dil)
    y = out[0] if cfg["stats"] else out
    close(nchw(y), want, dtype, "conv2d out (16-row tile)")
    if cfg["stats"]:
        tot = out[1].double().sum(2).float().cpu()
        close(tot[0, :cfg["cout"]], want.sum((0, 2, 3)), dtype, "stats sum", scale=float(want.abs().sum((0, 2, 3)).max()))
        close(tot[1, :cfg["cout"]], (want * want).sum((0, 2, 3)), dtype, "stats sumsq")


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(cin=256, cout=256, h=32, w=32),       # layer3: dgrad on the 16-row tile, wgrad at the one-resident-round split
    dict(cin=128, cout=256, h=32, w=32),
    dict(cin=128, cout=128, h=32, w=32),
    dict(cin=64, cout=64, h=64, w=64, n=32),   # layer1 shape (8-row tile; the wgrad split of the 64-channel layers)
    dict(cin=256, cout=32, h=64, w=64, n=32),  # conv_pred: 32-wide output-channel tile of the weight gradient
])
def test_conv2d_dgrad_wgrad_bench_scale(ops, dtype, mma, cfg):
    """data gradient through the same RW = 4 instantiation, weight gradient at the 512-workgroup split, both the direct
    (dh_conv2d_wgrad) and the deferred (dh_conv2d_wgrad_partial + dh_wgrad_reduce_multi) reductions"""
    N = cfg.get("n", BENCH_N)
    x = rnd((N, cfg["cin"], cfg["h"], cfg["w"]), dtype, 611).requires_grad_(True)
    w = rnd((cfg["cout"], cfg["cin"], 3, 3), dtype, 612, scale=(cfg["cin"] * 9) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = rnd(tuple(y.shape), dtype, 613)
    y.backward(dy)
    ck = ops.chunk_channels(dtype)
    cout_k = -(-cfg["cout"] // ck) * ck
    _, wd = ops.pack_weight(w.detach().cuda(), dtype, want_dgrad=True, dgrad_inner=cout_k)
    dyp = torch.zeros(N, cfg["h"], cfg["w"], cout_k)
    dyp[..., :cfg["cout"]] = nhwc(dy)
    if cout_k >= 128:
        assert _is_rw4(ops, N, cfg["h"], cfg["w"], cout_k, dtype=dtype)
    dx = ops.conv2d(dev(dyp, dtype), wd, cfg["cin"], 3, 1, 1)
    close(nchw(dx), x.grad, dtype, "dgrad (bench scale)", factor=2.0)
    xd, dyd = dev(nhwc(x.detach()), dtype), dev(nhwc(dy), dtype)
    gscale = float(w.grad.abs().max())
    # K = N*H*W pixels (65 536 .. 131 072): bf16 products, fp32 accumulation in split-K slabs
    fac = 4.0 if dtype == torch.float32 else 1.0
    dw = torch.full(tuple(w.shape), 0.5, device="cuda")
    ops.conv2d_wgrad(xd, dyd, dw, 3, 1, 1, accumulate=True)
    close(dw - 0.5, w.grad, dtype, "wgrad direct", scale=gscale, factor=fac)
    plan = ops.WgradPlan(xd.device)
    dw2 = torch.full(tuple(w.shape), 0.25, device="cuda")
    dw3 = torch.zeros(tuple(w.shape), device="cuda")
    for _ in range(2):                  # second pass reuses the persistent slabs and the device job table
        dw2.fill_(0.25)
        dw3.zero_()
        with plan:
            ops.conv2d_wgrad(xd, dyd, dw2, 3, 1, 1, accumulate=True)
            ops.conv2d_wgrad(xd, dyd, dw3, 3, 1, 1, accumulate=False)
            plan.run()
        close(dw2 - 0.25, w.grad, dtype, "wgrad deferred (+=)", scale=gscale, factor=fac)
        close(dw3, w.grad, dtype, "wgrad deferred (=)", scale=gscale, factor=fac)
    assert torch.equal(dw3, dw - 0.5) or float((dw3 - (dw - 0.5)).abs().max()) <= 1e-6 * gscale


@pytest.mark.parametrize("cfg", [
    dict(cin=64, cout=256, n=16, h=128, w=129, blocks=1),      # 256 x 64 block, one per pixel split
    dict(cin=256, cout=64, n=16, h=128, w=128, blocks=1),      # 64 x 256
    dict(cin=128, cout=512, n=8, h=127, w=129, blocks=2),      # 256 x 128, two co tiles; a ragged last stage
    dict(cin=512, cout=128, n=8, h=128, w=128, blocks=2),      # 128 x 256
    dict(cin=256, cout=1024, n=8, h=64, w=64, blocks=8),       # 4 x 2 blocks (the layer3 shape of a ResNet-50 trunk)
    dict(cin=1024, cout=256, n=8, h=64, w=64, blocks=8),
    dict(cin=264, cout=520, n=8, h=64, w=66, blocks=9),        # channel counts that are multiples of 8 only: ragged blocks both ways
    dict(cin=256, cout=1024, n=8, h=32, w=32, blocks=0),       # too few pixels to fill the chip with fat blocks: the slab kernel
    dict(cin=256, cout=512, n=8, h=128, w=128, blocks=4, stride=2),      # the stride-2 shortcut of a Bottleneck: x read at (2 oy, 2 ox)
    dict(cin=128, cout=512, n=8, h=127, w=129, blocks=2, stride=2),      # odd input sizes would not matter: output 127 x 129 of 253 x 257
])
def test_weight_gradient_of_wide_1x1_layers_in_256_by_128_blocks(ops, cfg):
    """dh_conv2d_wgrad for the bf16 1x1 / stride-1 layers of a Bottleneck (models/resnet.py:76-122) -- wgrad1x1_kernel, a CT x IT
    block of dW per workgroup over flat pixels -- against autograd of F.conv2d in fp32: direct (+)= / (=) and deferred"""
    dtype = torch.bfloat16
    N, H, W, Cin, Cout = cfg["n"], cfg["h"], cfg["w"], cfg["cin"], cfg["cout"]
    st = cfg.get("stride", 1)
    assert ops._lib.lib().dh_conv2d_wgrad_1x1_blocks(N, H, W, Cin, Cout) == cfg["blocks"]       # (H, W: the OUTPUT grid)
    x = rnd((N, Cin, st * H - (st - 1), st * W - (st - 1)), dtype, 2301)
    w = rnd((Cout, Cin, 1, 1), dtype, 2302, scale=Cin ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, st)
    assert tuple(y.shape[2:]) == (H, W)
    dy = rnd(tuple(y.shape), dtype, 2303)
    y.backward(dy)
    xd, dyd = dev(nhwc(x), dtype), dev(nhwc(dy), dtype)
    gscale = float(w.grad.abs().max())
    dw = torch.full(tuple(w.shape), 0.5, device="cuda")
    ops.conv2d_wgrad(xd, dyd, dw, 1, st, 0, accumulate=True)
    close(dw - 0.5, w.grad, dtype, "wgrad 1x1 (+=)", scale=gscale)
    dw2 = torch.full(tuple(w.shape), 7.0, device="cuda")
    ops.conv2d_wgrad(xd, dyd, dw2, 1, st, 0, accumulate=False)
    assert torch.equal(dw2, dw - 0.5) or float((dw2 - (dw - 0.5)).abs().max()) <= 1e-6 * gscale
    plan = ops.WgradPlan(xd.device)
    dw3 = torch.zeros(tuple(w.shape), device="cuda")
    with plan:
        ops.conv2d_wgrad(xd, dyd, dw3, 1, st, 0, accumulate=False)
        plan.run()
    close(dw3, w.grad, dtype, "wgrad 1x1 deferred", scale=gscale)


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(n=4, cin=64, cout=64, h=24, w=40, groups=2),          # layer1 conv2 shape class (8-row tile, no prefetch), ragged
    dict(n=64, cin=128, cout=128, h=32, w=32, groups=2),       # 16-row tile
    dict(n=64, cin=256, cout=256, h=32, w=32, groups=2),       # 16-row tile, 8 chunks (prefetch pipeline)
    dict(n=2, cin=32, cout=2, h=40, w=56, groups=1),           # class head: one chunk, generic epilogue
    dict(n=3, cin=32, cout=32, h=20, w=20, groups=1),
])
def test_batchnorm_relu_on_load_equals_materialised_activation(ops, dtype, mma, cfg):
    """dh_conv2d_fwd(in_scale, in_shift) and dh_conv2d_wgrad_bn_in on the PRE-normalisation tensor against the same
    kernels fed the activation that dh_bn_apply materialises: same values enter the MFMAs (the transform is the same
    fp32 expression, rounded once), so outputs agree to summation-order level; zero padding stays zero."""
    N, C, H, W, G = cfg["n"], cfg["cin"], cfg["h"], cfg["w"], cfg["groups"]
    y = dev(rnd((N, H, W, C), dtype, 701, 1.5), dtype)
    scale = (rnd((G, C), torch.float32, 702, 0.3) + 1.0).cuda()
    shift = rnd((G, C), torch.float32, 703, 0.5).cuda()
    w = rnd((cfg["cout"], C, 3, 3), dtype, 704, (C * 9) ** -0.5)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    b = rnd((cfg["cout"],), torch.float32, 705, 0.1).cuda()
    act = ops.bn_apply(y, scale, shift, G, ops.ACT_RELU)
    lazy = ops.BnInput(y, scale, shift, G)
    want_stats = cfg["cout"] % 16 == 0
    o0 = ops.conv2d(act, wp, cfg["cout"], 3, 1, 1, bias=b, want_stats=want_stats)
    o1 = ops.conv2d(lazy, wp, cfg["cout"], 3, 1, 1, bias=b, want_stats=want_stats)
    if want_stats:
        assert torch.equal(o0[1], o1[1])
        o0, o1 = o0[0], o1[0]
    assert torch.equal(o0, o1), float((o0.float() - o1.float()).abs().max())
    # against torch: relu(bn) with zero padding AFTER the activation
    gi = torch.arange(N) // (N // G)
    ref_act = torch.relu(y.float().cpu() * scale.cpu()[gi][:, None, None, :] + shift.cpu()[gi][:, None, None, :])
    ref_act = ref_act.to(dtype).float()
    want = F.conv2d(nchw(ref_act), w, b.cpu(), 1, 1)
    close(nchw(o1), want, dtype, "conv on BN+ReLU-on-load")
    # weight gradient
    dy = dev(rnd((N, H, W, cfg["cout"]), dtype, 706), dtype)
    dw0 = torch.zeros(cfg["cout"], C, 3, 3, device="cuda")
    dw1 = torch.zeros_like(dw0)
    ops.conv2d_wgrad(act, dy, dw0, 3, 1, 1)
    ops.conv2d_wgrad(lazy, dy, dw1, 3, 1, 1)
    assert torch.equal(dw0, dw1), float((dw0 - dw1).abs().max())
    plan = ops.WgradPlan(y.device)
    dw2 = torch.zeros_like(dw0)
    with plan:
        ops.conv2d_wgrad(lazy, dy, dw2, 3, 1, 1, accumulate=True)
        plan.run()
    assert torch.equal(dw0, dw2)


def test_encoder_stacks_of_independent_levels_in_one_launch(ops):
    """ops.EncoderBatch (dh_encoder_batch_*): the token-encoder stacks of DAHiTra's three levels (different heads, batch and
    token counts) recorded and issued as ONE forward launch and one backward + one parameter-gradient launch -- bit-identical
    to the three separate launches (same kernel bodies, one workgroup per image)"""
    from dahitra_amd import _lib
    cases = [(6, 8, 1, 8, 64, 64), (5, 8, 2, 4, 64, 64), (3, 6, 1, 4, 64, 32)]      # B, n, depth, heads, dim_head, mlp
    data = []
    for k, (B, n, depth, heads, dh, mlp) in enumerate(cases):
        inner = heads * dh
        shapes = [(32,), (32,), (3 * inner, 32), (32, inner), (32,), (32,), (32,), (mlp, 32), (mlp,), (32, mlp), (32,)]
        stride = sum(int(np.prod(sh)) for sh in shapes)
        flat = (rnd((depth * stride,), torch.float32, 1300 + k, 0.2)).cuda()
        params, off = [], 0
        for i, sh in enumerate(shapes):
            nsh = int(np.prod(sh))
            for d in range(depth):
                if i in (0, 5):
                    flat[d * stride + off:d * stride + off + nsh] += 1.0
            params.append(flat[off:off + nsh].view(sh))
            off += nsh
        x = rnd((B * n, 32), torch.float32, 1310 + k).cuda()
        dy = rnd((B * n, 32), torch.float32, 1320 + k).cuda()
        data.append((x, dy, params, stride, flat))

    def run(batched):
        outs, gflat = [], [torch.zeros_like(d[4]) for d in data]
        grads = []
        for (x, dy, params, stride, flat), gf, (B, n, depth, heads, dh, mlp) in zip(data, gflat, cases):
            off, gl = 0, []
            for p in params:
                gl.append(gf[off:off + p.numel()].view(p.shape))
                off += p.numel()
            grads.append(gl)
        with ops.EncoderBatch() as eb:
            assert eb.on == batched
            fw = [ops.encoder_fwd(x, B, n, depth, heads, dh, mlp, stride if depth > 1 else 0, params, True)
                  for (x, dy, params, stride, flat), (B, n, depth, heads, dh, mlp) in zip(data, cases)]
            assert _lib.lib().dh_encoder_batch_pending() == (3 if batched else 0)
            eb.launch()
            dxs = [ops.encoder_bwd(dy, xs, B, n, depth, heads, dh, mlp, stride if depth > 1 else 0, params, gl)
                   for (x, dy, params, stride, flat), (y, xs), gl, (B, n, depth, heads, dh, mlp) in zip(data, fw, grads, cases)]
            assert _lib.lib().dh_encoder_batch_pending() == (3 if batched else 0)
            eb.launch()
        torch.cuda.synchronize()
        return [y for y, _ in fw], dxs, gflat

    import os
    os.environ["DAHITRA_ENC_BATCH"] = "0"
    try:
        ref = run(False)
    finally:
        del os.environ["DAHITRA_ENC_BATCH"]
    got = run(True)
    for group_r, group_g in zip(ref, got):
        for a, b in zip(group_r, group_g):
            assert torch.equal(a, b)
    assert all(float(g.abs().max()) > 0 for g in got[2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cat_of_the_two_streams_and_its_gradient_in_one_launch(ops, dtype):
    """ops.cat_halves / split_halves (dh_cat_halves): torch.cat([t[:B], t[B:]], channel) of the two temporal streams
    (models/networks.py:1309, 1344) and the way back, exact copies"""
    t = dev(rnd((6, 5, 7, 32), dtype, 2201), dtype)
    cat = ops.cat_halves(t)
    assert torch.equal(cat, torch.cat([t[:3], t[3:]], dim=3))
    assert torch.equal(ops.split_halves(cat), t)


def test_decoder_layers_of_independent_stacks_in_one_launch(ops):
    """ops.EncoderBatch(decoder=True) (dh_decoder_batch_*): fused decoder layers of independent stacks -- different image
    counts and map sizes, both MLP widths -- recorded and issued as one launch per direction and width: bit-identical to the
    separate launches (forward output, data gradient); the per-workgroup parameter-gradient partials add up to the same sums (a
    launch of several jobs sizes its blocks for the launch as a whole, csrc/decoder_fused.hip dec_balance)"""
    from dahitra_amd import _lib
    dtype, D = torch.bfloat16, 32
    cases = [(6, 1024, 32), (4, 256, 32), (2, 4096, 32), (3, 512, 64)]          # images, rows per image, mlp

    class Prep:
        pass
    data = []
    for k, (images, rpi, mlp) in enumerate(cases):
        rows = images * rpi
        prep = Prep()
        kq, voT = rnd((images, 32, D), dtype, 1402 + 10 * k, 0.3), rnd((images, D, 32), dtype, 1403 + 10 * k, 0.3)
        prep.kq, prep.voT = dev(kq, dtype), dev(voT, dtype)
        prep.vo, prep.kqT = dev(voT.transpose(1, 2).contiguous(), dtype), dev(kq.transpose(1, 2).contiguous(), dtype)
        vec = lambda n, seed, base=0.0: (base + 0.1 * rnd((n,), torch.float32, seed)).cuda()
        g1, b1, g2, b2, bo = vec(D, 1404 + 10 * k, 1.0), vec(D, 1405 + 10 * k), vec(D, 1406 + 10 * k, 1.0), vec(D, 1407 + 10 * k), vec(D, 1408 + 10 * k)
        fb1, fb2 = vec(mlp, 1409 + 10 * k), vec(D, 1400 + 10 * k)
        w1p, w1T = ops.pack_weight(rnd((mlp, D), dtype, 1401 + 10 * k, D ** -0.5).cuda(), dtype, want_dgrad=True)
        w2p, w2T = ops.pack_weight(rnd((D, mlp), dtype, 1391 + 10 * k, mlp ** -0.5).cuda(), dtype, want_dgrad=True)
        x = dev(rnd((rows, D), dtype, 1392 + 10 * k, 1.0), dtype)
        dy = dev(rnd((rows, D), dtype, 1393 + 10 * k, 1.0), dtype)
        data.append((x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, mlp))

    def run(batched):
        parts = [torch.zeros(ops.decoder_layer_bwd_partial_floats(d[0].shape[0], d[3], d[15]), dtype=torch.float32, device="cuda")
                 for d in data]
        with ops.EncoderBatch(decoder=batched) as eb:
            ys = [ops.decoder_layer_fwd(x, prep, rpi, g1, b1, bo, g2, b2, w1p, fb1, w2p, fb2, mlp)
                  for (x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, mlp) in data]
            assert _lib.lib().dh_decoder_batch_pending() == (4 if batched else 0)
            eb.launch()
            dxs = [ops.decoder_layer_bwd(x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, None, mlp, partial=pt)[0]
                   for (x, dy, prep, rpi, g1, b1, bo, g2, b2, w1p, w1T, fb1, w2p, w2T, fb2, mlp), pt in zip(data, parts)]
            assert _lib.lib().dh_decoder_batch_pending() == (4 if batched else 0)
            eb.launch()
        torch.cuda.synchronize()
        return ys, dxs, parts
    ref, got = run(False), run(True)
    for gr, gg in zip(ref[:2], got[:2]):
        for a, b in zip(gr, gg):
            assert torch.equal(a, b)
    assert all(float(p.abs().max()) > 0 for p in got[2])
    for d, pr, pg in zip(data, ref[2], got[2]):
        size = 6400 if d[15] == 64 else 4320                  # PL<MLP>::SIZE floats per workgroup
        sr, sg = pr.view(-1, size).double().sum(0), pg.view(-1, size).double().sum(0)
        assert float((sr - sg).abs().max()) <= 1e-5 * float(sr.abs().max())


def test_weight_gradients_of_a_pass_in_one_launch(ops):
    """dh_wgrad_batch_*: the wave-specialised 3x3 weight gradients of a backward pass recorded and issued as ONE launch (fewer,
    longer K slices per layer) against the same layers launched one by one: same products, another split of the pixel sum.
    Covers both batched families (64-wide wave-specialised blocks, 32-channel layers), a BatchNorm-on-load input, a ragged
    image size, accumulate into an existing gradient, a layer the batch does not take (16 output channels), and the pass
    repeated on the same plan (the recorded graph does exactly that)."""
    from dahitra_amd import _lib
    dtype = torch.bfloat16
    layers = [dict(n=64, c=64, o=64, h=64, w=64, bn=False), dict(n=64, c=128, o=128, h=32, w=32, bn=True),
              dict(n=16, c=256, o=256, h=32, w=32, bn=False), dict(n=6, c=64, o=128, h=24, w=40, bn=True),
              dict(n=8, c=32, o=32, h=32, w=32, bn=False), dict(n=32, c=64, o=32, h=64, w=64, bn=True),
              dict(n=4, c=32, o=16, h=32, w=32, bn=False)]
    xs, dys = [], []
    for i, L in enumerate(layers):
        x = dev(rnd((L["n"], L["h"], L["w"], L["c"]), dtype, 1200 + i, 1.0), dtype)
        if L["bn"]:
            sc = (rnd((2, L["c"]), torch.float32, 1220 + i, 0.3) + 1.0).cuda()
            sh = rnd((2, L["c"]), torch.float32, 1240 + i, 0.5).cuda()
            x = ops.BnInput(x, sc, sh, 2)
        xs.append(x)
        dys.append(dev(rnd((L["n"], L["h"], L["w"], L["o"]), dtype, 1260 + i, 1.0), dtype))
    out = {}
    for batch in (False, True):
        plan = ops.WgradPlan("cuda", batch=batch)
        for rep in range(2):
            dws = [torch.full((L["o"], L["c"], 3, 3), 0.25, device="cuda") for L in layers]
            with plan:
                for x, dy, dw in zip(xs, dys, dws):
                    ops.conv2d_wgrad(x, dy, dw, 3, 1, 1, accumulate=True)
                if rep == 0:
                    assert _lib.lib().dh_wgrad_batch_pending() == (6 if batch else 0)
                plan.run()
                assert _lib.lib().dh_wgrad_batch_pending() == 0
            torch.cuda.synchronize()
            if rep == 1:
                assert all(torch.equal(a, b) for a, b in zip(dws, out[batch])), "the repeated pass differs"
            out[batch] = dws
    for i, (a, b) in enumerate(zip(out[False], out[True])):
        s = float((a - 0.25).abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * s + 1e-6, (i, float((a - b).abs().max()), s)
    assert torch.equal(out[False][6], out[True][6])          # the 16-channel layer never entered a batch
    # a batch left open by a pass that never finished must not swallow the launches of the next (plain) plan
    _lib.lib().dh_wgrad_batch_begin()
    dw = torch.full((64, 64, 3, 3), 0.25, device="cuda")
    with ops.WgradPlan("cuda") as plan:
        ops.conv2d_wgrad(xs[0], dys[0], dw, 3, 1, 1, accumulate=True)
        assert _lib.lib().dh_wgrad_batch_pending() == 0
        plan.run()
    assert torch.equal(dw, out[False][0])


@pytest.mark.parametrize("cfg", [
    dict(n=64, h=64, w=64, c=64, groups=2, mode="out"),        # layer1 bn2 (residual: mask from the stored output), full chip
    dict(n=64, h=64, w=64, c=64, groups=2, mode="recompute"),  # layer1 bn1 (mask recomputed from x)
    dict(n=64, h=32, w=32, c=256, groups=2, mode="out"),       # layer3
    dict(n=64, h=32, w=32, c=128, groups=2, mode="none"),      # downsample BN (no ReLU)
    dict(n=6, h=24, w=40, c=64, groups=2, mode="out"),         # small, ragged: partly filled workgroups, idle workgroups
    dict(n=3, h=20, w=20, c=128, groups=1, mode="recompute"),
])
def test_batchnorm_backward_persistent_launch_equals_two_pass(ops, cfg, monkeypatch):
    """csrc/bn_bwd_persist.hip (tensors held on chip across a device-wide barrier) against the reduce / finalize / apply
    kernels: same formula, different summation order; repeated launches re-arm the barrier words"""
    from dahitra_amd import _lib
    dtype = torch.bfloat16
    N, H, W, C, G = cfg["n"], cfg["h"], cfg["w"], cfg["c"], cfg["groups"]
    assert _lib.lib().dh_bn_bwd_persist_supported(1, N * H * W, C, G)
    x = dev(rnd((N, H, W, C), dtype, 801, 1.5), dtype)
    dout = dev(rnd((N, H, W, C), dtype, 802), dtype)
    mean = rnd((G, C), torch.float32, 803, 0.2).cuda()
    invstd = (rnd((G, C), torch.float32, 804, 0.1) + 0.9).cuda()
    gamma = (rnd((C,), torch.float32, 805, 0.1) + 1.0).cuda()
    scale = (rnd((G, C), torch.float32, 806, 0.2) + 1.0).cuda()
    shift = rnd((G, C), torch.float32, 807, 0.5).cuda()
    out = torch.relu(dev(rnd((N, H, W, C), dtype, 808), dtype)) if cfg["mode"] == "out" else None
    kw = dict(mask_scale=scale, mask_shift=shift) if cfg["mode"] == "recompute" else {}
    res = {}
    for persist in (False, True, True, True):
        monkeypatch.setattr(ops, "BN_BWD_PERSIST", "force" if persist else False)
        dg, db = torch.full((C,), 0.5, device="cuda"), torch.full((C,), -0.25, device="cuda")
        r = ops.bn_bwd(dout, out, x, mean, invstd, gamma, dg, db, G, accumulate=True, want_dres=(out is not None), **kw)
        dx, dres = r if out is not None else (r, None)
        cur = (dx.float(), dg - 0.5, db + 0.25, None if dres is None else dres.float())
        if persist and True in res:
            for a, b in zip(cur, res[True]):          # repeated persistent launches: bitwise identical
                assert a is None or torch.equal(a, b)
        res[persist] = cur
    ops.bn_persist_check(x.device)                # raises if the device-wide barrier timed out
    assert ops.bn_sync_words(x.device)[:2].abs().sum().item() == 0, "barrier words not re-armed"
    two, per = res[False], res[True]
    s = float(two[0].abs().max())
    assert float((two[0] - per[0]).abs().max()) <= 2 ** -7 * s, "dx"                 # one bf16 ulp of the output scale
    assert float((two[0] - per[0]).abs().mean()) <= 2e-4 * s
    assert float((two[1] - per[1]).abs().max()) <= 2e-4 * float(two[1].abs().max()) + 1e-4, "dgamma"
    assert float((two[2] - per[2]).abs().max()) <= 2e-4 * float(two[2].abs().max()) + 1e-4, "dbeta"
    if two[3] is not None:
        assert torch.equal(two[3], per[3]), "dres"


def test_batchnorm_backward_persistent_keeps_small_gradients(ops, monkeypatch):
    """the cross-workgroup sums of dh_bn_bwd_persist are integer (order-independent) but must not round small partial sums
    away: a gradient scaled by 2^-30 gives dgamma / dbeta scaled by exactly 2^-30 (every step of the fp32 chain scales
    exactly, and the two-word fixed point carries fp32 partials down to 2^-74), and both agree with an fp64 sum"""
    dtype = torch.bfloat16
    N, H, W, C, G = 64, 32, 32, 128, 2
    x = dev(rnd((N, H, W, C), dtype, 821, 1.5), dtype)
    dout = dev(rnd((N, H, W, C), dtype, 822), dtype)
    mean = rnd((G, C), torch.float32, 823, 0.2).cuda()
    invstd = (rnd((G, C), torch.float32, 824, 0.1) + 0.9).cuda()
    gamma = (rnd((C,), torch.float32, 825, 0.1) + 1.0).cuda()
    monkeypatch.setattr(ops, "BN_BWD_PERSIST", "force")
    got = {}
    for e in (0, 30):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ops.bn_bwd((dout.float() * 2.0 ** -e).to(dtype), None, x, mean, invstd, gamma, dg, db, G)
        got[e] = (dg.double() * 2.0 ** e, db.double() * 2.0 ** e)
    ops.bn_persist_check(x.device)
    g64 = dout.double().view(G, -1, C)
    xh = (x.double().view(G, -1, C) - mean.double()[:, None, :]) * invstd.double()[:, None, :]
    want = ((g64 * xh).sum((0, 1)), g64.sum((0, 1)))
    for name, a, b, w in zip(("dgamma", "dbeta"), got[0], got[30], want):
        s = float(w.abs().max())
        assert float((a - w).abs().max()) <= 1e-5 * s, name
        assert float((b - w).abs().max()) <= 1e-5 * s, name + " of the 2^-30 gradient"
        assert float((a - b).abs().max()) <= 3e-7 * s, name + ": scaling the gradient must scale the sums"


def test_batchnorm_backward_persistent_failures_are_loud(ops, monkeypatch):
    """the device-wide barrier of dh_bn_bwd_persist: a timeout (forced: the barrier waits for one arrival that never comes,
    with a short spin limit) and a non-finite gradient must both (a) poison dgamma / dbeta with NaN in the same launch and
    (b) make ops.bn_persist_check() raise; afterwards the words are clear again and a normal launch is clean"""
    from dahitra_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    N, H, W, C, G = 64, 32, 32, 128, 2
    x = dev(rnd((N, H, W, C), dtype, 811, 1.5), dtype)
    dout = dev(rnd((N, H, W, C), dtype, 812), dtype)
    mean = rnd((G, C), torch.float32, 813, 0.2).cuda()
    invstd = (rnd((G, C), torch.float32, 814, 0.1) + 0.9).cuda()
    gamma = (rnd((C,), torch.float32, 815, 0.1) + 1.0).cuda()
    monkeypatch.setattr(ops, "BN_BWD_PERSIST", "force")

    def run(d):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dx = ops.bn_bwd(d, None, x, mean, invstd, gamma, dg, db, G)
        torch.cuda.synchronize()
        return dx, dg, db
    ops.bn_persist_check()                       # clean slate
    _, dg, db = run(dout)
    assert torch.isfinite(dg).all() and torch.isfinite(db).all()
    ops.bn_persist_check()                       # no complaint
    # (a) forced barrier timeout
    L.dh_bn_bwd_persist_test_spin_limit(256)
    try:
        _, dg, db = run(dout)
    finally:
        L.dh_bn_bwd_persist_test_spin_limit(0)
    assert torch.isnan(dg).all() and torch.isnan(db).all(), "a timed-out barrier must poison the parameter gradients"
    with pytest.raises(_lib.HipLibraryError, match="barrier"):
        ops.bn_persist_check()
    ops.bn_persist_check()                       # the word was cleared by the failed check
    # (b) a non-finite gradient: the fixed-point accumulators cannot carry it -> flagged, NaN out (as the two-pass path gives)
    bad = dout.clone()
    bad.view(-1)[12345] = float("inf")
    _, dg, db = run(bad)
    assert torch.isnan(dg).all() and torch.isnan(db).all()
    with pytest.raises(_lib.HipLibraryError):
        ops.bn_persist_check()
    _, dg, db = run(dout)                         # and the barrier words re-armed themselves
    assert torch.isfinite(dg).all() and torch.isfinite(db).all()
    ops.bn_persist_check()
    # no_persist_bn(): the two-pass kernels, whatever the switch says
    with ops.no_persist_bn():
        before = {k: v.clone() for k, v in ops._BN_SYNC.items()}
        run(dout)
        assert all(torch.equal(before[k], ops._BN_SYNC[k]) for k in before)


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(n=2, cin=256, h=16, w=16),          # 8-row tiles
    dict(n=3, cin=64, h=12, w=20),           # ragged coarse map
    dict(n=64, cin=256, h=32, w=32),         # the bench's conv_pred: 16-row tiles, 4 phases x 256 tiles
    dict(n=2, cin=32, h=16, w=16, relu=True),         # the UNet up path (conv_layer4 / 3 / 2: 32 -> 32 channels + ReLU)
    dict(n=3, cin=32, h=12, w=20, relu=True),         # ... on a ragged coarse map
    dict(n=8, cin=32, h=128, w=128, relu=True),       # ... conv_layer2's map at batch 8
])
def test_upsample2_conv3x3_as_four_phase_convs(ops, dtype, mma, cfg):
    """conv3x3(nearest-upsample-x2(x)) (models/networks.py:251-256: upsamplex2 + conv_pred; :1341-1351 with a ReLU: upsamplex2 +
    conv_layer<l>) as four 2x2 phase convolutions: forward, data gradient and weight / bias gradients against torch autograd
    of F.interpolate + F.conv2d (+ relu: the gradient then goes through the activation's backward first, as the engine does)"""
    N, Cin, H, W = cfg["n"], cfg["cin"], cfg["h"], cfg["w"]
    relu = cfg.get("relu", False)
    x = rnd((N, Cin, H, W), dtype, 901).requires_grad_(True)
    w = rnd((32, Cin, 3, 3), torch.float32, 902, (Cin * 9) ** -0.5).requires_grad_(True)
    b = rnd((32,), torch.float32, 903, 0.1)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, 1, 1)
    if relu:
        y = F.relu(y)
    dy = rnd(tuple(y.shape), dtype, 904)
    y.backward(dy)
    wf, wd, b4 = ops.pack_phase_weights(w.detach().cuda(), b.cuda(), dtype)
    xd = dev(nhwc(x.detach()), dtype)
    out = ops.conv_up2_fwd(xd, wf, b4, act=ops.ACT_RELU if relu else ops.ACT_NONE)
    assert tuple(out.shape) == (N, 2 * H, 2 * W, 32)
    # (bf16: the phase weights are sums of up to four taps rounded once, the reference rounds nothing: same tolerance class)
    close(nchw(out), y.detach(), dtype, "phase conv forward", factor=2.0)
    dyd = dev(nhwc(dy), dtype)
    if relu:        # the mask of the REFERENCE's output (an element within rounding of zero may differ between the two)
        dyd = ops.act_bwd(dyd, dev(nhwc(y.detach()), dtype), ops.ACT_RELU)
    dx = ops.conv_up2_dgrad(dyd, wd, Cin)
    close(nchw(dx), x.grad, dtype, "phase conv data gradient", factor=3.0)
    dw = torch.full((32, Cin, 3, 3), 0.5, device="cuda")
    ops.conv_up2_wgrad(xd, dyd, dw, accumulate=True)
    fac = 4.0 if dtype == torch.float32 or N < 64 else 1.0
    close(dw - 0.5, w.grad, dtype, "phase conv weight gradient", factor=fac)


@pytest.mark.parametrize("dtype,mma", DTYPES_MMA)
@pytest.mark.parametrize("cfg", [
    dict(n=2, cin=64, cout=128, h=32, w=32),          # layer2.0.conv1 at small batch (8-row tiles)
    dict(n=3, cin=32, cout=64, h=20, w=24),           # 32-channel phases, ragged coarse map (10 x 12)
    dict(n=64, cin=64, cout=128, h=64, w=64),         # the bench's shape: 16-row tiles
])
def test_stride2_conv_data_gradient_as_phase_convs(ops, dtype, mma, cfg):
    """data gradient of conv3x3 / stride 2 / pad 1 without the zero-inserted dY (four output-parity phases), plus the coarse
    gradient of the block's 1x1 stride-2 shortcut added at the even-even positions, against torch autograd"""
    N, Cin, Cout, H, W = cfg["n"], cfg["cin"], cfg["cout"], cfg["h"], cfg["w"]
    x = rnd((N, Cin, H, W), dtype, 911).requires_grad_(True)
    w = rnd((Cout, Cin, 3, 3), dtype, 912, (Cin * 9) ** -0.5)
    wds = rnd((Cout, Cin, 1, 1), dtype, 913, Cin ** -0.5)
    dy = rnd((N, Cout, H // 2, W // 2), dtype, 914)
    dyds = rnd((N, Cout, H // 2, W // 2), dtype, 915)
    (F.conv2d(x, w, None, 2, 1) * dy).sum().backward()
    g_main = x.grad.clone()
    x.grad = None
    (F.conv2d(x, wds, None, 2, 0) * dyds).sum().backward()
    g_both = g_main + x.grad
    wph = ops.pack_s2_dgrad_phase_weights(w.cuda(), dtype)
    dyd, dydsd = dev(nhwc(dy), dtype), dev(nhwc(dyds), dtype)
    dx = ops.conv3x3s2_dgrad(dyd, wph, Cin)
    close(nchw(dx), g_main, dtype, "stride-2 data gradient (phases)", factor=2.0)
    _, wd1 = ops.pack_weight(wds.cuda(), dtype, want_dgrad=True, dgrad_inner=Cout)
    coarse = ops.conv2d(dydsd, wd1, Cin, 1, 1, 0)
    dx2 = ops.conv3x3s2_dgrad(dyd, wph, Cin, coarse)
    # (bf16: the coarse shortcut gradient is rounded to bf16 before it is added, as the separate path does too)
    close(nchw(dx2), g_both, dtype, "stride-2 data gradient + coarse shortcut gradient", factor=3.0)


@pytest.mark.parametrize("mlp", [64, 32])
def test_decoder_layer_fp8_forward(ops, mlp):
    """csrc/decoder_fp8.hip (OCP e4m3 MFMA operands, per-row weight scales, fp32 accumulation) against the bf16 fused
    layer and against a plain torch fp32 evaluation of the same layer (help_funcs.py:66-114, 52-63 in re-associated form).
    Accuracy contract: relative L2 distance to the bf16 kernel <= 3e-2."""
    torch.manual_seed(0)
    images, rpi, D = 3, 256, 32
    rows = images * rpi
    dtype = torch.bfloat16
    x = dev(rnd((rows, D), dtype, 1001, 1.0), dtype)

    class Prep:
        pass
    prep = Prep()
    kq32 = rnd((images, 32, D), dtype, 1002, 0.3)
    voT32 = rnd((images, D, 32), dtype, 1003, 0.3)
    prep.kq, prep.voT = dev(kq32, dtype), dev(voT32, dtype)
    g1, b1 = (1 + 0.1 * rnd((D,), torch.float32, 1004)).cuda(), (0.1 * rnd((D,), torch.float32, 1005)).cuda()
    g2, b2 = (1 + 0.1 * rnd((D,), torch.float32, 1006)).cuda(), (0.1 * rnd((D,), torch.float32, 1007)).cuda()
    bo = (0.1 * rnd((D,), torch.float32, 1008)).cuda()
    w1 = rnd((mlp, D), dtype, 1009, D ** -0.5)
    w2 = rnd((D, mlp), dtype, 1010, mlp ** -0.5)
    fb1, fb2 = (0.1 * rnd((mlp,), torch.float32, 1011)).cuda(), (0.1 * rnd((D,), torch.float32, 1012)).cuda()
    w1p, _ = ops.pack_weight(w1.cuda(), dtype, want_dgrad=False)
    w2p, _ = ops.pack_weight(w2.cuda(), dtype, want_dgrad=False)
    y16 = ops.decoder_layer_fwd(x, prep, rpi, g1, b1, bo, g2, b2, w1p, fb1, w2p, fb2, mlp).float().cpu()
    y8 = ops.decoder_layer_fwd(x, prep, rpi, g1, b1, bo, g2, b2, w1p, fb1, w2p, fb2, mlp, fp8=True).float().cpu()
    # torch fp32 evaluation: heads * 4 keys = the 32 rows of Kq; softmax over the 4 keys of a head
    xf = x.float().cpu()
    xn = F.layer_norm(xf, (D,), g1.cpu(), b1.cpu(), 1e-5).reshape(images, rpi, D)
    dots = torch.einsum("ipc,ihc->iph", xn, kq32)
    attn = torch.softmax(dots.reshape(images, rpi, 8, 4), -1).reshape(images, rpi, 32)
    x1 = torch.einsum("iph,ich->ipc", attn, voT32).reshape(rows, D) + bo.cpu() + xf
    l2 = F.layer_norm(x1, (D,), g2.cpu(), b2.cpu(), 1e-5)
    ref = x1 + F.gelu(l2 @ w1.t() + fb1.cpu()) @ w2.t() + fb2.cpu()
    rel = lambda u, v: float((u - v).norm() / v.norm())
    print("decoder layer (mlp %d): bf16 vs fp32 %.3e, fp8 vs fp32 %.3e, fp8 vs bf16 %.3e" % (mlp, rel(y16, ref), rel(y8, ref), rel(y8, y16)))
    assert rel(y16, ref) <= 1e-2
    assert rel(y8, y16) <= 3e-2 and rel(y8, ref) <= 3e-2
    assert torch.isfinite(y8).all()


@pytest.mark.parametrize("mlp", [64, 32])
@pytest.mark.parametrize("rpi", [256, 1024])
def test_decoder_layer_bf16_backward_against_torch_autograd(ops, mlp, rpi):
    """csrc/decoder_fused.hip, bf16, BACKWARD: dx, the operand gradients dKq / dVo^T and every parameter gradient of one fused
    cross-attention + MLP layer against torch autograd (fp32, CPU) of the same layer -- models/help_funcs.py:52-63 (FeedForward
    with nn.GELU, i.e. the erf form), :66-114 (Cross_Attention in the re-associated Kq / Vo form), :170-186 (the residual
    wiring of TransformerDecoder).  The kernel recomputes the forward chain from x in bf16 and uses the fitted logistic-quintic
    `gelu_fast` and its derivative (tools/gelu_fit.py: 2.7e-5 / 1.1e-4 from the erf form), so the bounds are bf16 bounds:
    relative L2 <= 1e-2 for y and dx, <= 1.5e-2 for every other tensor (measured on MI355X: y 2.3e-3, dx 2.5e-3, dKq 3.7e-3,
    dVo^T 3.0e-3, dW1 / dW2 3.0e-3, LayerNorm gradients <= 5.2e-3)."""
    torch.manual_seed(0)
    images, D = 3, 32
    rows = images * rpi
    dtype = torch.bfloat16
    x = rnd((rows, D), dtype, 2001, 1.0)
    dy = rnd((rows, D), dtype, 2002, 1.0)
    kq = rnd((images, 32, D), dtype, 2003, 0.3)
    voT = rnd((images, D, 32), dtype, 2004, 0.3)
    g1, b1 = 1 + 0.1 * rnd((D,), torch.float32, 2005), 0.1 * rnd((D,), torch.float32, 2006)
    g2, b2 = 1 + 0.1 * rnd((D,), torch.float32, 2007), 0.1 * rnd((D,), torch.float32, 2008)
    bo = 0.1 * rnd((D,), torch.float32, 2009)
    w1 = rnd((mlp, D), dtype, 2010, D ** -0.5)
    w2 = rnd((D, mlp), dtype, 2011, mlp ** -0.5)
    fb1, fb2 = 0.1 * rnd((mlp,), torch.float32, 2012), 0.1 * rnd((D,), torch.float32, 2013)
    # ---- torch autograd, fp32 ----
    leaves = [t.clone().requires_grad_(True) for t in (x, kq, voT, g1, b1, bo, g2, b2, w1, fb1, w2, fb2)]
    tx, tkq, tvoT, tg1, tb1, tbo, tg2, tb2, tw1, tfb1, tw2, tfb2 = leaves
    xn = F.layer_norm(tx, (D,), tg1, tb1, 1e-5).reshape(images, rpi, D)
    dots = torch.einsum("ipc,ihc->iph", xn, tkq)
    attn = torch.softmax(dots.reshape(images, rpi, 8, 4), -1).reshape(images, rpi, 32)
    x1 = torch.einsum("iph,ich->ipc", attn, tvoT).reshape(rows, D) + tbo + tx
    l2 = F.layer_norm(x1, (D,), tg2, tb2, 1e-5)
    y = x1 + F.gelu(l2 @ tw1.t() + tfb1) @ tw2.t() + tfb2
    y.backward(dy)
    # ---- the kernel ----

    class Prep:
        pass
    prep = Prep()
    prep.kq, prep.voT = dev(kq, dtype), dev(voT, dtype)
    prep.kqT, prep.vo = dev(kq.transpose(1, 2), dtype), dev(voT.transpose(1, 2), dtype)
    w1p, w1T = ops.pack_weight(w1.cuda(), dtype, want_dgrad=True)
    w2p, w2T = ops.pack_weight(w2.cuda(), dtype, want_dgrad=True)
    cu = lambda t: t.cuda().contiguous()
    grads = [torch.zeros_like(t).cuda() for t in (w1, w2, fb1, fb2, bo, g1, b1, g2, b2)]
    yk = ops.decoder_layer_fwd(dev(x, dtype), prep, rpi, cu(g1), cu(b1), cu(bo), cu(g2), cu(b2), w1p, cu(fb1), w2p, cu(fb2), mlp)
    dx, dkq, dvoT = ops.decoder_layer_bwd(dev(x, dtype), dev(dy, dtype), prep, rpi, cu(g1), cu(b1), cu(bo), cu(g2), cu(b2),
                                          w1p, w1T, cu(fb1), w2p, w2T, cu(fb2), grads, mlp)
    rel = lambda u, v: float((u.float().cpu() - v).norm() / v.norm())
    got = dict(y=(yk, y.detach()), dx=(dx, tx.grad), dkq=(dkq, tkq.grad), dvoT=(dvoT, tvoT.grad), dw1=(grads[0], tw1.grad),
               dw2=(grads[1], tw2.grad), db1=(grads[2], tfb1.grad), db2=(grads[3], tfb2.grad), dbo=(grads[4], tbo.grad),
               dln1_g=(grads[5], tg1.grad), dln1_b=(grads[6], tb1.grad), dln2_g=(grads[7], tg2.grad), dln2_b=(grads[8], tb2.grad))
    errs = {k: rel(u.reshape(v.shape), v) for k, (u, v) in got.items()}
    print("decoder layer backward (mlp %d, %d rows per image): " % (mlp, rpi) + ", ".join("%s %.2e" % kv for kv in errs.items()))
    assert errs["y"] <= 1e-2 and errs["dx"] <= 1e-2, errs
    for k, e in errs.items():
        assert e <= 1.5e-2, (k, e)


@pytest.mark.parametrize("cfg", [
    dict(n=4, cin=256, cout=64, h=16, w=16, stats=True, res=False, relu=False),      # Bottleneck conv1 (64-cout tile)
    dict(n=2, cin=64, cout=256, h=16, w=32, stats=True, res=False, relu=False),      # conv3, one K step
    dict(n=2, cin=1024, cout=256, h=16, w=16, stats=False, res=True, relu=True),     # layer3 conv1, eval form, 16 K steps
    dict(n=16, cin=256, cout=1024, h=32, w=32, stats=True, res=False, relu=False),   # layer3 conv3 at size
    dict(n=3, cin=128, cout=128, h=8, w=16, stats=False, res=False, relu=False),     # odd image count (M = 3 * 128)
    dict(n=64, cin=256, cout=1024, h=32, w=32, stats=True, res=False, relu=False),   # 256 x 256 tiles (8 waves; two statistics halves)
    dict(n=64, cin=1024, cout=256, h=32, w=32, stats=False, res=True, relu=True),    # ... 16 K steps, eval form
    dict(n=32, cin=512, cout=128, h=64, w=64, stats=True, res=False, relu=False),    # 256 x 128 tiles (layer2 conv1)
])
def test_conv1x1_kdeep_gemm_path(ops, cfg):
    """csrc/conv1x1_gemm.hip (direct-to-LDS loads, 64 channels per step, swizzled 128-byte rows) through dh_conv2d_fwd, bf16:
    output, BatchNorm partial sums, bias / residual / ReLU, and the data gradient (a 1x1 conv with transposed weights)"""
    dtype = torch.bfloat16
    N, Cin, Cout, H, W = cfg["n"], cfg["cin"], cfg["cout"], cfg["h"], cfg["w"]
    x = rnd((N, Cin, H, W), dtype, 1101).requires_grad_(True)
    w = rnd((Cout, Cin, 1, 1), dtype, 1102, Cin ** -0.5)
    b = rnd((Cout,), torch.float32, 1103, 0.1)
    want = F.conv2d(x, w, b)
    r = rnd(tuple(want.shape), dtype, 1104) if cfg["res"] else None
    out_ref = want if r is None else want + r
    if cfg["relu"]:
        out_ref = F.relu(out_ref)
    wp, wd = ops.pack_weight(w.cuda(), dtype, want_dgrad=True)
    out = ops.conv2d(dev(nhwc(x.detach()), dtype), wp, Cout, 1, 1, 0, bias=b.cuda(),
                     residual=dev(nhwc(r), dtype) if r is not None else None,
                     act=ops.ACT_RELU if cfg["relu"] else ops.ACT_NONE, want_stats=cfg["stats"])
    y = out[0] if cfg["stats"] else out
    close(nchw(y), out_ref.detach(), dtype, "1x1 GEMM conv out")
    if cfg["stats"]:
        tot = out[1].double().sum(2).float().cpu()
        close(tot[0, :Cout], out_ref.detach().sum((0, 2, 3)), dtype, "stats sum", scale=float(out_ref.detach().abs().sum((0, 2, 3)).max()))
        close(tot[1, :Cout], (out_ref.detach() ** 2).sum((0, 2, 3)), dtype, "stats sumsq")
    dy = rnd(tuple(want.shape), dtype, 1105)
    want.backward(dy)
    dx = ops.conv2d(dev(nhwc(dy), dtype), wd, Cin, 1, 1, 0)
    close(nchw(dx), x.grad, dtype, "1x1 GEMM data gradient", factor=2.0)


# ---- the register-resident-weights form of the 3x3 convolutions (csrc/conv_wreg.hip) --------------------------------
@pytest.mark.parametrize("cfg", [
    dict(n=64, cin=64, cout=64, h=64, w=64, stats=True),                         # layer1 convs (8-row statistics units)
    dict(n=64, cin=64, cout=64, h=64, w=64, res=True),                           # ... their data gradients with the shortcut
    dict(n=64, cin=64, cout=64, h=64, w=64, stats=True, bn_in=True),             # conv2 of a block: BatchNorm + ReLU on load
    dict(n=64, cin=128, cout=128, h=32, w=32, stats=True),                       # layer2 (16-row statistics units)
    dict(n=64, cin=128, cout=128, h=32, w=32, stats=True, bn_in=True),
    dict(n=64, cin=128, cout=256, h=32, w=32, stats=True),                       # layer3.0.conv1
    dict(n=64, cin=256, cout=128, h=32, w=32, res=True),                         # its data gradient
    dict(n=64, cin=256, cout=256, h=32, w=32, stats=True),                       # layer3
    dict(n=64, cin=256, cout=256, h=32, w=32, stats=True, bn_in=True),
    dict(n=64, cin=256, cout=256, h=32, w=32, res=True, relu=True, bias=True),   # eval form: bias + residual + ReLU
    dict(n=6, cin=64, cout=64, h=48, w=80, stats=True, bn_in=True),              # ragged stream: 180 tiles over 512 workgroups
    dict(n=10, cin=256, cout=64, h=40, w=48, res=True),                          # one output-channel block, odd tile count
    dict(n=8, cin=32, cout=32, h=256, w=256, stats=True),                        # classifier.0 (the 32 -> 32 kernel: row halves)
    dict(n=8, cin=32, cout=32, h=128, w=128, res=True, relu=True, bias=True),
    dict(n=3, cin=32, cout=32, h=72, w=112, stats=True),                         # ragged stream
])
def test_conv3x3_register_resident_weights_equals_tap_kernel(ops, cfg):
    """dh_conv2d_fwd on the persistent register-resident-weights kernel against the tap-oriented kernel (dh_conv_wreg_mode 0):
    the same MFMA products accumulated in the same order => bit-identical outputs; statistics to fp32 round-off (their
    partial sums are taken over other pixel subsets).  And against torch's fp32 convolution."""
    from dahitra_amd import _lib
    L = _lib.lib()
    dtype = torch.bfloat16
    N, cin, cout, H, W = cfg["n"], cfg["cin"], cfg["cout"], cfg["h"], cfg["w"]
    x = rnd((N, cin, H, W), dtype, 801)
    w = rnd((cout, cin, 3, 3), dtype, 802, scale=(cin * 9) ** -0.5)
    b = rnd((cout,), torch.float32, 803, 0.1) if cfg.get("bias") else None
    r = rnd((N, cout, H, W), dtype, 804) if cfg.get("res") else None
    groups = 2
    xin = x
    bn = None
    if cfg.get("bn_in"):
        scale = rnd((groups, cin), torch.float32, 805, 0.5) + 0.7
        shift = rnd((groups, cin), torch.float32, 806, 0.5)
        per = N // groups
        xin = torch.cat([F.relu(x[g * per:(g + 1) * per] * scale[g].view(1, -1, 1, 1) + shift[g].view(1, -1, 1, 1))
                         for g in range(groups)]).to(dtype).float()
        bn = (scale.cuda(), shift.cuda())
    want = F.conv2d(xin, w, b, 1, 1)
    if r is not None:
        want = want + r
    if cfg.get("relu"):
        want = F.relu(want)
    wp, _ = ops.pack_weight(w.cuda(), dtype, want_dgrad=False)
    plan = ops.PackPlan(torch.device("cuda"))
    wf, _ = plan.add(w.cuda(), dtype, want_dgrad=False, frag=True)       # the fragment-order copy the engine hands over
    plan.run()
    xd = dev(nhwc(x), dtype)
    arg = ops.BnInput(xd, bn[0], bn[1], groups) if bn else xd
    res = {}
    for mode in (0, 1):
        prev = L.dh_conv_wreg_mode(mode)
        try:
            res[mode] = ops.conv2d(arg, wp, cout, 3, 1, 1, bias=b.cuda() if b is not None else None,
                                   residual=dev(nhwc(r), dtype) if r is not None else None,
                                   act=ops.ACT_RELU if cfg.get("relu") else ops.ACT_NONE, want_stats=bool(cfg.get("stats")),
                                   w_frag=wf if (mode == 1 and cfg["n"] != 6) else None)
        finally:
            L.dh_conv_wreg_mode(prev)
    y0, y1 = (res[0][0], res[1][0]) if cfg.get("stats") else (res[0], res[1])
    assert torch.equal(y0, y1), "outputs differ: max |d| = %g" % float((y0.float() - y1.float()).abs().max())
    close(nchw(y1), want, dtype, "register-resident conv vs torch")
    if cfg.get("stats"):
        s0, s1 = res[0][1].double(), res[1][1].double()
        assert s0.shape == s1.shape
        per_g = s0.shape[2] // groups
        for g in range(groups):          # per BatchNorm group, as bn_finalize sums them
            a0, a1 = s0[:, :, g * per_g:(g + 1) * per_g].sum(2), s1[:, :, g * per_g:(g + 1) * per_g].sum(2)
            assert float((a0 - a1).abs().max()) <= 2e-5 * float(a0.abs().max())


@pytest.mark.parametrize("cfg", [dict(n=2, h=16, w=24, k=32), dict(n=32, h=64, w=64, k=32), dict(n=3, h=6, w=8, k=64)])
def test_conv3x3_data_gradient_through_bilinear_up4_without_the_fine_tensor(ops, cfg):
    """backward of conv3x3(bilinear_x4(|a - b|)) (models/networks.py:383-389): the fused pair dh_conv3x3_dgrad_up4 +
    dh_absdiff_up4_combine against (i) the two-kernel path it replaces (data gradient written at 4H x 4W in bf16, then the
    gather kernel) and (ii) torch autograd in fp32"""
    dtype = torch.bfloat16
    N, H, W, K = cfg["n"], cfg["h"], cfg["w"], cfg["k"]
    a = rnd((N, 32, H, W), dtype, 901).requires_grad_(True)
    b = rnd((N, 32, H, W), dtype, 902).requires_grad_(True)
    w = rnd((K, 32, 3, 3), dtype, 903, scale=(32 * 9) ** -0.5)
    dy = rnd((N, K, 4 * H, 4 * W), dtype, 904)
    up = F.interpolate((a - b).abs(), scale_factor=4, mode="bilinear", align_corners=False)
    F.conv2d(up, w, None, 1, 1).backward(dy)
    _, wd = ops.pack_weight(w.cuda(), dtype, want_dgrad=True, dgrad_inner=K)
    ad, bd, dyd = dev(nhwc(a.detach()), dtype), dev(nhwc(b.detach()), dtype), dev(nhwc(dy), dtype)
    da, db = ops.conv3x3_dgrad_through_up4(dyd, wd, ad, bd)
    dfine = ops.conv2d(dyd, wd, 32, 3, 1, 1)                         # the path it replaces
    da2, db2 = ops.absdiff_upsample4_bwd(ad, bd, dfine)
    scale = float(a.grad.abs().max())
    close(nchw(da), a.grad, dtype, "da (fused) vs torch", scale=scale)
    close(nchw(db), b.grad, dtype, "db (fused) vs torch", scale=scale)
    assert float((da.float() - da2.float()).abs().max()) <= 2.0 ** -6 * scale      # the old path rounds the fine gradient to bf16
    assert torch.equal(db, -da) or float((db.float() + da.float()).abs().max()) == 0.0
    # more accurate than the path it replaces (fp32 from the accumulators to the coarse sum)
    e_new = float((nchw(da).float().cpu() - a.grad).abs().mean())
    e_old = float((nchw(da2).float().cpu() - a.grad).abs().mean())
    assert e_new <= e_old * 1.05 + 1e-9, (e_new, e_old)


# ---- cross-attention operand preparation on the matrix cores (csrc/tokens.hip: xattn_prep_mfma_kernel / _bwd_) ------------------
@pytest.mark.parametrize("cfg", [dict(heads=8, dh=64, S=64, layers=8), dict(heads=4, dh=64, S=8, layers=4),
                                 dict(heads=1, dh=32, S=12, layers=1)])
def test_cross_attention_prep_on_the_matrix_cores_matches_the_scalar_kernels(ops, cfg):
    """ops.XattnPrepStack with `masters` (four images per workgroup = the 16 columns of an MFMA, bf16 operands, fp32
    accumulation) against the one-workgroup-per-image fp32-FMA kernels it replaces: saved LayerNorm rows bit-level, k / v /
    Kq / Vo and every gradient of the backward to bf16 operand rounding (help_funcs.py:66-114 re-associated)"""
    heads, dh, S, layers = cfg["heads"], cfg["dh"], cfg["S"], cfg["layers"]
    B, L, inner, dt = S // 2, 4, heads * dh, torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(77)
    tok = torch.randn(B, 2 * L, 32, device="cuda", generator=g)
    bstride, sstride = 2 * L * 32, L * 32
    stride = 4 * inner * 32 + 64
    arena = torch.randn(layers * stride, device="cuda", generator=g) * 0.2
    offs = dict(wq=0, wk=inner * 32, wv=2 * inner * 32, wo=3 * inner * 32, g=4 * inner * 32, b=4 * inner * 32 + 32)
    shp = dict(wq=(inner, 32), wk=(inner, 32), wv=(inner, 32), wo=(32, inner), g=(32,), b=(32,))

    def view(buf, l, name):
        n = int(np.prod(shp[name]))
        return buf[l * stride + offs[name]: l * stride + offs[name] + n].view(*shp[name])
    T = {n: torch.stack([view(arena, l, n).t().contiguous().to(dt).reshape(-1) for l in range(layers)]) for n in ("wq", "wk", "wv", "wo")}
    p0 = {n: view(arena, 0, n) for n in shp}
    mk = lambda masters: ops.XattnPrepStack(tok, bstride, sstride, B, S, L, heads, dh, layers, stride, p0["g"], p0["b"], p0["wq"],
                                            T["wk"], T["wv"], T["wo"], dt, masters=masters)
    old, new = mk(None), mk((p0["wk"], p0["wv"], p0["wo"], T["wq"]))
    assert new.mfma and not old.mfma
    for n in ("mn", "mstats"):
        assert float((getattr(old, n) - getattr(new, n)).abs().max()) <= 1e-6
    for n in ("k", "v", "kq", "kqT", "vo", "voT"):
        a, b = getattr(old, n).float(), getattr(new, n).float()
        assert float((a - b).abs().max()) <= 1e-2 * float(a.abs().max()), n
    res = {}
    for name, st in (("old", old), ("new", new)):
        g2 = torch.Generator(device="cuda").manual_seed(5)
        st.dkq.copy_(torch.randn(st.dkq.shape, device="cuda", generator=g2))
        st.dvoT.copy_(torch.randn(st.dvoT.shape, device="cuda", generator=g2))
        st.dkq[:, :, heads * L:, :] = 0
        st.dvoT[:, :, :, heads * L:] = 0
        dtok, garena = torch.zeros_like(tok), torch.zeros_like(arena)
        st.backward(tok, dtok, p0["g"], T["wq"], p0["wk"], p0["wv"], p0["wo"], view(garena, 0, "g"), view(garena, 0, "b"),
                    view(garena, 0, "wq"), view(garena, 0, "wk"), view(garena, 0, "wv"), view(garena, 0, "wo"))
        res[name] = (dtok, garena)
    for i, n in enumerate(("token gradient", "parameter gradients")):
        a, b = res["old"][i], res["new"][i]
        assert float((a - b).abs().max()) <= 1.5e-2 * float(a.abs().max()), n


def test_conv3x3_over_a_channel_concatenation_in_place(ops):
    """conv_layer2_0(torch.cat([a_128, b_128], 1)) (models/networks.py:1344) without the concatenated tensor: forward with
    BatchNorm statistics, data gradient (written as the two halves of the [2B] gradient) and weight gradient read / write the
    two batch halves in place (ops.SplitCat); each must equal -- bit for bit: same kernels, same order of products -- the
    path through the materialised concatenation, which itself is checked against torch autograd here."""
    B, H, W, C = 4, 128, 128, 64
    dtype = torch.bfloat16
    assert ops.conv3x3_split_supported(B, H, W, 2 * C, 128, dtype)
    assert not ops.conv3x3_split_supported(B, H, W, 2 * C, 128, torch.float32)
    t = rnd((2 * B, C, H, W), dtype, 1201)
    w = rnd((128, 2 * C, 3, 3), dtype, 1202, (2 * C * 9) ** -0.5)
    dy = rnd((B, 128, H, W), dtype, 1203)
    td, dyd = dev(nhwc(t), dtype), dev(nhwc(dy), dtype)
    plan = ops.PackPlan(td.device)
    f, d = plan.add(w.cuda(), dtype, want_dgrad=True, dgrad_inner=128)
    ff, dd = plan.add(w.cuda(), dtype, want_dgrad=True, dgrad_inner=128, frag=True)
    plan.run()
    sc = ops.SplitCat(td)
    assert tuple(sc.shape) == (B, H, W, 2 * C)
    cat = sc.materialize()
    assert torch.equal(cat[..., :C], td[:B]) and torch.equal(cat[..., C:], td[B:])
    # forward (+ statistics)
    y_ref, st_ref = ops.conv2d(cat, f, 128, 3, 1, 1, want_stats=True, w_frag=ff)
    y, st = ops.conv3x3_split(sc, f, ff, 128, want_stats=True)
    assert torch.equal(y, y_ref) and torch.equal(st, st_ref)
    xc = torch.cat([t[:B], t[B:]], 1).requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    yt = F.conv2d(xc, wt, None, 1, 1)
    close(nchw(y), yt.detach(), dtype, "split conv forward")
    yt.backward(dy)
    # data gradient into the two halves
    dx_ref = ops.conv2d(dyd, d, 2 * C, 3, 1, 1, w_frag=dd)
    dx = ops.conv3x3_split(dyd, d, dd, 2 * C, split_out=True)
    assert tuple(dx.shape) == (2 * B, H, W, C)
    assert torch.equal(dx[:B], dx_ref[..., :C]) and torch.equal(dx[B:], dx_ref[..., C:])
    close(nchw(dx[:B]), xc.grad[:, :C], dtype, "split conv data gradient (first half)", factor=2.0)
    close(nchw(dx[B:]), xc.grad[:, C:], dtype, "split conv data gradient (second half)", factor=2.0)
    # weight gradient
    dw_ref = torch.zeros(128, 2 * C, 3, 3, device="cuda")
    dw = torch.full_like(dw_ref, 0.25)
    ops.conv2d_wgrad(cat, dyd, dw_ref, 3, 1, 1)
    ops.conv2d_wgrad(sc, dyd, dw, 3, 1, 1, accumulate=True)
    assert torch.equal(dw - 0.25, dw_ref) or float((dw - 0.25 - dw_ref).abs().max()) <= 1e-6 * float(dw_ref.abs().max())
    close(dw_ref, wt.grad, dtype, "split conv weight gradient", factor=4.0)


def test_head_data_gradient_with_the_relu_in_front_of_it_folded_in(ops):
    """dh_head_dgrad3x3_relu: the class head's data gradient already masked by the ReLU that produced the head's input
    (classifier(conv_layer2(...)), models/networks.py:1351-1355) == the plain data gradient followed by the activation's backward"""
    N, H, W = 3, 64, 48
    dtype = torch.bfloat16
    dl = dev(rnd((N, H, W, 8), dtype, 1301), dtype)
    dl[..., 2:] = 0
    w = rnd((2, 32, 3, 3), torch.float32, 1302, 0.1).cuda()
    out = dev(rnd((N, H, W, 32), dtype, 1303), dtype).clamp_(min=0)
    want = ops.act_bwd(ops.head_dgrad3x3(dl, w, 2), out, ops.ACT_RELU)
    got = ops.head_dgrad3x3(dl, w, 2, relu_out=out)
    assert torch.equal(got, want)
    assert float((got != 0).float().mean()) < float((ops.head_dgrad3x3(dl, w, 2) != 0).float().mean())


@pytest.mark.parametrize("cfg", [dict(n=3, h=64, w=48, ncls=2), dict(n=2, h=9, w=21, ncls=1), dict(n=8, h=128, w=128, ncls=2),
                                 dict(n=2, h=40, w=24, ncls=5), dict(n=1, h=64, w=64, ncls=8)])
@pytest.mark.parametrize("dtype,mma", [pytest.param(torch.bfloat16, 0, id="bfloat16"), pytest.param(torch.float32, 1, id="bf16x3")])
def test_head_behind_a_relu_data_weight_and_bias_gradient_in_one_pass(ops, cfg, dtype, mma):
    """dh_head_relu_bwd against torch autograd of conv2d(relu_out, W, b) with the gradient masked by relu_out > 0
    (classifier(conv_layer2(...)), models/networks.py:1351-1355) and against the kernels it replaces"""
    N, H, W, ncls = cfg["n"], cfg["h"], cfg["w"], cfg["ncls"]
    pre = rnd((N, 32, H, W), dtype, 1311).float().requires_grad_(True)
    w = rnd((ncls, 32, 3, 3), torch.float32, 1312, 0.1).requires_grad_(True)
    hb = torch.zeros(ncls, requires_grad=True)
    dlog = rnd((N, ncls, H, W), dtype, 1313)
    F.conv2d(torch.relu(pre), w, hb, 1, 1).backward(dlog.float())
    out = dev(nhwc(torch.relu(pre).detach()), dtype)
    dlp = ops.head_dlogits_pack(dlog.float().cuda().contiguous(), dtype)
    dw, db = torch.full((ncls, 32, 3, 3), 0.25, device="cuda"), torch.full((ncls,), 1.0, device="cuda")
    dx = ops.head_relu_bwd(dlp, w.detach().cuda(), ncls, out, dw, db, accumulate=True)
    close(nchw(dx), pre.grad, dtype, "dx vs autograd", factor=2.0)
    close(dw - 0.25, w.grad, dtype, "dw vs autograd", factor=2.0)
    close(db - 1.0, hb.grad, dtype, "db vs autograd", factor=2.0)
    dl = ops.nchw_to_nhwc(dlog.float().cuda().contiguous(), dtype, cpad=8 if (dtype == torch.bfloat16 or ncls > 4) else 4)
    if dtype == torch.bfloat16 and ncls <= 2:
        assert torch.equal(dx, ops.head_dgrad3x3(dl, w.detach().cuda(), ncls, relu_out=out))      # same products, same order
    else:
        close(dx, ops.head_dgrad3x3(dl, w.detach().cuda(), ncls, relu_out=out).cpu(), dtype, "dx vs head_dgrad3x3", factor=1.0)
    dw0 = torch.zeros_like(dw)
    ops.conv2d_wgrad(out, dl, dw0, 3, 1, 1, accumulate=False, cout_real=ncls)
    dw1, db1 = torch.zeros_like(dw), torch.zeros_like(db)
    ops.head_relu_bwd(dlp, w.detach().cuda(), ncls, out, dw1, db1, accumulate=False)
    close(dw1, dw0.cpu(), dtype, "dw vs conv2d_wgrad", factor=1.0)


@pytest.mark.parametrize("dtype", DTYPES)
def test_coarse_grid_gradient_added_at_the_even_positions(ops, dtype):
    """dh_add_coarse: x[n, 2y, 2x, :] += coarse[n, y, x, :] -- with the coarse-grid 1x1 product it is the data gradient of a 1x1
    stride-2 convolution (the shortcut of a stride-2 Bottleneck, models/resnet.py:106-118) == zero insertion + 1x1 convolution"""
    for (N, H, W, C) in ((2, 12, 20, 64), (3, 7, 9, 32)):
        OH, OW = (H + 1) // 2, (W + 1) // 2
        x = rnd((N, H, W, C), dtype, 2501)
        c = rnd((N, OH, OW, C), dtype, 2502)
        want = x.clone()
        want[:, ::2, ::2] += c
        got = ops.add_coarse_(dev(x, dtype), dev(c, dtype))
        close(got, want.to(dtype).float(), dtype, "add_coarse", factor=0.5)
    # against the fine-grid path: autograd of a 1x1 stride-2 convolution
    N, H, W, Cin, Cout = 2, 16, 32, 64, 128
    xx = rnd((N, Cin, H, W), dtype, 2503).requires_grad_(True)
    w = rnd((Cout, Cin, 1, 1), dtype, 2504, scale=Cin ** -0.5)
    y = F.conv2d(xx, w, None, 2)
    dy = rnd(tuple(y.shape), dtype, 2505)
    y.backward(dy)
    _, wd = ops.pack_weight(w.cuda(), dtype, want_dgrad=True)
    coarse = ops.conv2d(dev(nhwc(dy), dtype), wd, Cin, 1, 1, 0)
    fine = torch.zeros(N, H, W, Cin, device="cuda", dtype=dtype)
    ops.add_coarse_(fine, coarse)
    close(nchw(fine), xx.grad, dtype, "1x1 stride-2 data gradient on the coarse grid", factor=2.0)


def test_job_table_of_elementwise_kernels_equals_their_single_launches(ops):
    """ops.EncoderBatch(ew=True): add_pos / add_pos_bwd / cat_halves / split_halves / absdiff_halves(_bwd) calls of independent
    levels recorded and issued as ONE dh_ew_multi launch -- bit-equal to the single launches (three levels' worth of jobs, more
    jobs than one table holds, and a dtype the table does not take)"""
    dtype = torch.bfloat16
    lv = [(4, 16, 16), (4, 8, 8), (4, 4, 4)]          # (2B, h, w) of three levels
    x = [dev(rnd((n, h, w, 32), dtype, 2400 + i), dtype) for i, (n, h, w) in enumerate(lv)]
    pos = [rnd((1, 32, h, w), torch.float32, 2410 + i).cuda() for i, (n, h, w) in enumerate(lv)]
    tok = [rnd((2, 2, 4 * 32), torch.float32, 2420 + i).cuda() for i in range(3)]
    dout = [rnd((2, 4 * 32), torch.float32, 2430 + i).cuda() for i in range(3)]
    xf = dev(rnd((4, 6, 6, 32), torch.float32, 2440), torch.float32)        # fp32 activations: not a table job, issued at once

    def run():
        r = []
        for i in range(3):
            r.append(ops.add_pos(x[i], pos[i]))
            r.append(ops.cat_halves(x[i]))
            d = torch.zeros(2, 4 * 32, device="cuda")
            ops.absdiff_halves(tok[i], d)
            r.append(d)
        r.append(ops.add_pos(xf, pos[0][:, :, :6, :6].contiguous()))
        return r

    def run_bwd(cats):
        r = []
        for i in range(3):
            g = torch.full((1, 32, lv[i][1], lv[i][2]), 0.5, device="cuda")
            ops.add_pos_bwd(x[i], g, accumulate=True)
            r.append(g)
            r.append(ops.split_halves(cats[i]))
            dt_ = torch.ones(2, 2, 4 * 32, device="cuda")
            ops.absdiff_halves_bwd(tok[i], dout[i], dt_)
            r.append(dt_)
        return r
    want = run()
    cats = [want[3 * i + 1] for i in range(3)]
    want_b = run_bwd(cats)
    with ops.EncoderBatch(decoder=False, ew=True) as eb:
        got = run()
        assert ops._EW_BATCH is not None and len(ops._EW_BATCH) == 9          # the fp32 add_pos went out at once
        eb.launch()
        assert len(ops._EW_BATCH) == 0
        got_b = run_bwd(cats)
        for _ in range(2):          # 9 + 6 more jobs: the table (12) flushes itself
            ops.add_pos_bwd(x[0], torch.zeros(1, 32, 16, 16, device="cuda"), accumulate=False)
            ops.add_pos_bwd(x[1], torch.zeros(1, 32, 8, 8, device="cuda"), accumulate=False)
            ops.add_pos_bwd(x[2], torch.zeros(1, 32, 4, 4, device="cuda"), accumulate=False)
    assert ops._EW_BATCH is None
    for a_, b_ in zip(want + want_b, got + got_b):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("cfg", [
    dict(n=2, h=16, w=16),          # 64 x 64 fine map
    dict(n=3, h=6, w=10),           # 24 x 40: ragged tiles in both directions, odd batch
    dict(n=1, h=1, w=2),            # one coarse row: every bilinear index clamps
    dict(n=32, h=64, w=64),         # the bench's classifier.0: 32 x 256 x 256
])
def test_classifier0_on_the_upsampled_absdiff_map_formed_on_load(ops, cfg):
    """abs(x1 - x2) -> nn.Upsample(4, 'bilinear') -> classifier.0 (models/networks.py:383-389, help_funcs.py:9) with the
    upsampled map formed inside the convolution's loads (ops.Up4Input): equal -- bit for bit for the forward and its
    BatchNorm partial sums -- to the two-kernel path through the materialised map, which is checked against
    F.interpolate(..., 'bilinear', align_corners=False) + F.conv2d in fp32 and their autograd weight gradient."""
    N, h, w = cfg["n"], cfg["h"], cfg["w"]
    dtype = torch.bfloat16
    a, b = rnd((N, 32, h, w), dtype, 1401), rnd((N, 32, h, w), dtype, 1402)
    wt = rnd((32, 32, 3, 3), dtype, 1403, (32 * 9) ** -0.5).requires_grad_(True)
    bias = rnd((32,), torch.float32, 1404, 0.1)
    up = F.interpolate((a - b).abs(), scale_factor=4, mode="bilinear", align_corners=False)
    y_t = F.conv2d(up, wt, None, 1, 1)
    dy = rnd(tuple(y_t.shape), dtype, 1405)
    y_t.backward(dy)
    ad, bd, dyd = dev(nhwc(a), dtype), dev(nhwc(b), dtype), dev(nhwc(dy), dtype)
    wp, _ = ops.pack_weight(wt.detach().cuda(), dtype, want_dgrad=False)
    u = ops.Up4Input(ad, bd)
    assert tuple(u.shape) == (N, 4 * h, 4 * w, 32)
    mat = u.materialize()
    y_ref, st_ref = ops.conv2d(mat, wp, 32, 3, 1, 1, want_stats=True)
    y, st = ops.conv2d(u, wp, 32, 3, 1, 1, want_stats=True)
    assert torch.equal(y, y_ref), float((y.float() - y_ref.float()).abs().max())
    # (the materialised path may take the register-resident-weights kernel, whose per-tile sums add the two row halves of a
    # tile in another order: equal up to fp32 rounding there, bit-identical against the tap kernel)
    assert torch.equal(st, st_ref) or float((st - st_ref).abs().max()) <= 1e-5 * float(st_ref.abs().max())
    close(nchw(y), y_t.detach(), dtype, "conv over the upsampled |a - b| (formed on load)", factor=2.0)
    # bias + ReLU epilogue (the eval-mode form: BatchNorm folded into the weights, its shift as the bias)
    y2 = ops.conv2d(u, wp, 32, 3, 1, 1, bias=bias.cuda(), act=ops.ACT_RELU)
    assert torch.equal(y2, ops.conv2d(mat, wp, 32, 3, 1, 1, bias=bias.cuda(), act=ops.ACT_RELU))
    # weight gradient (Up4Input: through the materialised map), twice: reproducible bit for bit
    dw_ref = torch.zeros(32, 32, 3, 3, device="cuda")
    dw, dw2 = torch.full_like(dw_ref, -0.5), torch.full_like(dw_ref, -0.5)
    ops.conv2d_wgrad(mat, dyd, dw_ref, 3, 1, 1)
    ops.conv2d_wgrad(u, dyd, dw, 3, 1, 1, accumulate=True)
    ops.conv2d_wgrad(u, dyd, dw2, 3, 1, 1, accumulate=True)
    assert torch.equal(dw, dw2)
    err = float((dw + 0.5 - dw_ref).abs().max())
    assert err <= 2e-5 * float(dw_ref.abs().max()), err
    close(dw_ref, wt.grad, dtype, "weight gradient over the upsampled |a - b|", factor=4.0)


@pytest.mark.parametrize("cfg", [dict(n=2, h=16, w=16), dict(n=5, h=8, w=12), dict(n=1, h=2, w=4), dict(n=3, h=32, w=64)])
def test_classifier0_upsample_fused_into_the_weights_resident_stream(ops, cfg):
    """north_star "bilinear upsample fused with the seg head", forward: conv3x3_up4_wreg32_kernel (csrc/conv_wreg.hip) -- the
    persistent 32 -> 32 stream whose halo images are interpolated in LDS from the two coarse maps, one piece per lane and step,
    three small stages ahead of the matrix work -- against the materialised path (dh_absdiff_upsample4_fwd, then the same
    convolution): bit for bit, at sizes that give a workgroup one tile, a few, and a ragged count (dh_conv_wreg_mode(1) makes the
    stream eligible from 16 tiles on); the per-tile BatchNorm sums to fp32 rounding (this stream adds a tile's two row halves)"""
    from dahitra_amd import _lib
    N, h, w = cfg["n"], cfg["h"], cfg["w"]
    dtype = torch.bfloat16
    a, b = rnd((N, 32, h, w), dtype, 1501), rnd((N, 32, h, w), dtype, 1502)
    wt = rnd((32, 32, 3, 3), dtype, 1503, (32 * 9) ** -0.5)
    bias = rnd((32,), torch.float32, 1504, 0.1)
    ad, bd = dev(nhwc(a), dtype), dev(nhwc(b), dtype)
    wp, _ = ops.pack_weight(wt.cuda(), dtype, want_dgrad=False)
    u = ops.Up4Input(ad, bd)
    mat = u.materialize()
    L = _lib.lib()
    prev = L.dh_conv_wreg_mode(0)                      # reference: the tap kernel on the materialised map
    try:
        y_ref, st_ref = ops.conv2d(mat, wp, 32, 3, 1, 1, want_stats=True)
        yb_ref = ops.conv2d(mat, wp, 32, 3, 1, 1, bias=bias.cuda())
        L.dh_conv_wreg_mode(1)                         # the fused stream wherever it can run
        y, st = ops.conv2d(u, wp, 32, 3, 1, 1, want_stats=True)
        yb = ops.conv2d(u, wp, 32, 3, 1, 1, bias=bias.cuda())
        y2, st2 = ops.conv2d(u, wp, 32, 3, 1, 1, want_stats=True)
    finally:
        L.dh_conv_wreg_mode(prev)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref), float((y.float() - y_ref.float()).abs().max())
    assert torch.equal(yb, yb_ref)
    assert torch.equal(y2, y) and torch.equal(st2, st)             # reproducible
    assert float((st - st_ref).abs().max()) <= 1e-5 * float(st_ref.abs().max())
    up = F.interpolate((a - b).abs(), scale_factor=4, mode="bilinear", align_corners=False)
    close(nchw(y), F.conv2d(up, wt, None, 1, 1), dtype, "conv over the upsampled |a - b| (interpolated in LDS)", factor=2.0)


@pytest.mark.parametrize("cfg", [dict(images=6, rpi=256, depth=8, mlp=32), dict(images=4, rpi=1024, depth=4, mlp=32),
                                 dict(images=2, rpi=4096, depth=3, mlp=64)])
def test_decoder_stack_in_one_launch_equals_layer_by_layer(ops, cfg):
    """ops.decoder_stack_fwd / _bwd (all layers of a cross-attention decoder stack in ONE launch per direction: a workgroup takes
    its pixel rows through every layer, help_funcs.py:170-186) against the same layers launched one by one: every layer's
    output, the data gradient and every layer's parameter-gradient partials, bit for bit."""
    images, rpi, depth, mlp = cfg["images"], cfg["rpi"], cfg["depth"], cfg["mlp"]
    D, dt = 32, torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(77 + depth)
    rn = lambda *s, sc=1.0: torch.randn(*s, device="cuda", generator=g) * sc
    rows = images * rpi
    x, dy = rn(rows, D).to(dt), rn(rows, D).to(dt)

    class Stack:
        pass
    st = Stack()
    st.layers = depth
    kq, voT = rn(depth, images, 32, D, sc=0.3), rn(depth, images, D, 32, sc=0.3)
    st.kq, st.voT = kq.to(dt), voT.to(dt)
    st.vo, st.kqT = voT.transpose(2, 3).contiguous().to(dt), kq.transpose(2, 3).contiguous().to(dt)
    # fp32 parameter vectors of all layers in one arena at a constant stride, as the net's flat parameter arena holds them
    PS = 7 * D + 2 * mlp + 5
    arena = rn(depth, PS, sc=0.1)
    off = dict(g1=0, b1=D, bo=2 * D, g2=3 * D, b2=4 * D, fb1=5 * D, fb2=5 * D + mlp)
    arena[:, off["g1"]:off["g1"] + D] += 1.0
    arena[:, off["g2"]:off["g2"] + D] += 1.0
    par = lambda l, n, ln: arena[l, off[n]:off[n] + ln]
    params = lambda l: (par(l, "g1", D), par(l, "b1", D), par(l, "bo", D), par(l, "g2", D), par(l, "b2", D), par(l, "fb1", mlp), par(l, "fb2", D))
    w1, w2 = rn(depth, mlp, D, sc=D ** -0.5), rn(depth, D, mlp, sc=mlp ** -0.5)
    w1s, w1Ts = w1.to(dt).contiguous(), w1.transpose(1, 2).contiguous().to(dt)
    w2s, w2Ts = w2.to(dt).contiguous(), w2.transpose(1, 2).contiguous().to(dt)
    pf = ops.decoder_layer_bwd_partial_floats(rows, rpi, mlp)

    class Prep:
        pass

    def prep(l):
        q = Prep()
        q.kq, q.voT, q.vo, q.kqT = st.kq[l], st.voT[l], st.vo[l], st.kqT[l]
        return q
    # layer by layer
    xs, cur = [x], x
    for l in range(depth):
        g1, b1, bo, g2, b2, fb1, fb2 = params(l)
        cur = ops.decoder_layer_fwd(cur, prep(l), rpi, g1, b1, bo, g2, b2, w1s[l], fb1, w2s[l], fb2, mlp)
        xs.append(cur)
    # (zeroed: the workspace is sized for the smallest blocks a re-planned batched launch may use, these launches fill a part of it)
    part_ref = torch.zeros(depth, pf, dtype=torch.float32, device="cuda")
    d = dy
    for l in range(depth - 1, -1, -1):
        g1, b1, bo, g2, b2, fb1, fb2 = params(l)
        d, _, _ = ops.decoder_layer_bwd(xs[l], d, prep(l), rpi, g1, b1, bo, g2, b2, w1s[l], w1Ts[l], fb1, w2s[l], w2Ts[l], fb2, None,
                                        mlp, partial=part_ref[l])
    # one launch per direction
    ys = ops.decoder_stack_fwd(x, st, rpi, params(0), w1s, w2s, PS, mlp)
    for l in range(depth):
        assert torch.equal(ys[l], xs[l + 1]), "layer %d output" % l
    part = torch.zeros(depth, pf, dtype=torch.float32, device="cuda")
    dx = ops.decoder_stack_bwd(x, ys, dy, st, rpi, params(0), w1s, w1Ts, w2s, w2Ts, PS, mlp, part)
    assert torch.equal(dx, d)
    assert torch.equal(part, part_ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [dict(n=4, h=16, w=24, c=64, groups=2), dict(n=2, h=8, w=8, c=256, groups=1),
                                 dict(n=64, h=64, w=64, c=64, groups=2)])       # the last: the persistent one-launch form (bf16)
def test_batchnorm_residual_relu_mask_as_bytes_equals_the_tensor_form(ops, cfg, dtype):
    """dh_bn_apply_bits / dh_bn_bwd_bits / dh_bn_bwd_persist_bits (bf16): the ReLU mask of out = relu(bn(y) + identity) as one
    byte per 8 elements -- the forward output is unchanged, the bytes are the mask of that output, and the backward that reads
    them equals the backward that reads the post-activation tensor BIT FOR BIT (two-pass and persistent forms)"""
    V = 8 if dtype == torch.bfloat16 else 4
    N, H, W, C, G = cfg["n"], cfg["h"], cfg["w"], cfg["c"], cfg["groups"]
    y = dev(rnd((N, H, W, C), dtype, 3001), dtype)
    res = dev(rnd((N, H, W, C), dtype, 3002), dtype)
    dout = dev(rnd((N, H, W, C), dtype, 3003), dtype)
    scale = (1 + 0.2 * rnd((G, C), torch.float32, 3004)).cuda()
    shift = (0.2 * rnd((G, C), torch.float32, 3005)).cuda()
    mean = (0.1 * rnd((G, C), torch.float32, 3006)).cuda()
    invstd = (1 + 0.1 * rnd((G, C), torch.float32, 3007)).abs().cuda()
    gamma = (1 + 0.1 * rnd((C,), torch.float32, 3008)).cuda()
    out0 = ops.bn_apply(y, scale, shift, G, ops.ACT_RELU, res)
    out1, bits = ops.bn_apply(y, scale, shift, G, ops.ACT_RELU, res, want_bits=True)
    assert bits is not None and bits.numel() == y.numel() // V and torch.equal(out0, out1)
    want = (out0.float().reshape(-1, V) > 0).to(torch.int32)
    got = (bits.to(torch.int32).unsqueeze(1) >> torch.arange(V, device="cuda", dtype=torch.int32)) & 1
    assert torch.equal(got, want)
    persist = bool(ops._lib.lib().dh_bn_bwd_persist_preferred(1 if dtype == torch.bfloat16 else 0, N * H * W, C, G))
    assert persist == (N == 64 and dtype == torch.bfloat16)
    for want_dres in (True, False):
        outs = []
        for b in (None, bits):
            dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
            r = ops.bn_bwd(dout, out0 if b is None else None, y, mean, invstd, gamma, dg, db, G, want_dres=want_dres, bits=b)
            outs.append((r if want_dres else (r,)) + (dg, db))
        for u, v in zip(*outs):
            assert torch.equal(u, v)
    ops.bn_persist_check(y.device)
