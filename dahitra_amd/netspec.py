"""net_G names, hyper-parameters and state-dict layout of the supported change-detection nets.

Mirrors the reference factory models/networks.py:130-168 (define_G) and the constructors it calls
(BASE_Transformer networks.py:260-310, BASE_Transformer_UNet networks.py:1142-1249, the ResNet-18
trunk models/resnet.py:125-204).  Key names and registration order are the reference's, so that a
`best_ckpt.pt` written by either side loads into the other (models/trainer.py:150-158)."""

DIM = 32
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5
ATTN_SCALE = DIM ** -0.5       # Attention / Cross_Attention use dim**-0.5 (help_funcs.py:71,122)

NET_CONFIGS = {
    "base_transformer_pos_s4": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=1, dec_dim_head=64),
    "base_transformer_pos_s4_dd8": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=8, dec_dim_head=64),
    "base_transformer_pos_s4_dd8_o5": dict(kind="bit", n_class=5, token_len=4, enc_depth=1, dec_depth=8,
                                           dec_dim_head=64),
    "base_transformer_pos_s4_dd8_dedim8": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=8,
                                               dec_dim_head=8),
    "base_transformer_pos_s4_dd8_t8_e2d4": dict(kind="bit", n_class=2, token_len=8, enc_depth=2, dec_depth=4,
                                                dec_dim_head=8),
    "newUNetTrans": dict(kind="unet", n_class=2, token_len=4, enc_depth=1),
    # ResNet-50 trunk: NOT a define_G name in the reference -- only reachable through the constructor
    # BASE_Transformer(..., backbone='resnet50') (networks.py:192-195); define_G below refuses it too.
    "base_transformer_pos_s4_resnet50": dict(kind="bit", n_class=2, token_len=4, enc_depth=1, dec_depth=1,
                                             dec_dim_head=64, backbone="resnet50", ctor_only=True),
    # xBD copy of the hierarchical model (xBD_code/zoo/model_transformer_encoding.py:242-449, built at
    # xBD_code/train.py:44-45).  One 6-channel input, 5 output channels, difference-only decoder pass, ModuleList
    # alias keys, `layer == 5/4/3` positional quirk.  The first is train.py's model (1024x1024 input only), the
    # second the same constructor with with_decoder_pos=None (any multiple of 64).  Constructor access only:
    # dahitra_amd.models.xbd.BASE_Transformer_UNet.
    "xbd_unet_transformer": dict(kind="xbd", n_class=5, token_len=4, enc_depth=1, decoder_pos=True, ctor_only=True),
    "xbd_unet_transformer_nodecpos": dict(kind="xbd", n_class=5, token_len=4, enc_depth=1, decoder_pos=False,
                                          ctor_only=True),
}
RESNET50_BLOCKS = (3, 4, 6, 3)
BIT_HEADS, BIT_DIM_HEAD = 8, 64
# hierarchical model, level suffix -> trunk channels / heads / decoder depth / dim_head / map size
UNET_LEVELS = {
    5: dict(cin=256, heads=4, dec_depth=4, dim_head=64, size=16),
    4: dict(cin=128, heads=4, dec_depth=4, dim_head=64, size=32),
    3: dict(cin=64, heads=8, dec_depth=8, dim_head=64, size=64),
    2: dict(cin=64, heads=1, dec_depth=1, dim_head=32, size=64),
}


def get_config(net_G):
    if net_G not in NET_CONFIGS:
        raise NotImplementedError("Generator model name [%s] is not recognized" % net_G)
    return dict(NET_CONFIGS[net_G], net_G=net_G)


def _bn(p, c):
    return [(p + ".weight", (c,), "bn_w"), (p + ".bias", (c,), "bn_b"), (p + ".running_mean", (c,), "bn_rm"),
            (p + ".running_var", (c,), "bn_rv"), (p + ".num_batches_tracked", (), "bn_nbt")]


def _resnet18(p="resnet"):
    s = [(p + ".conv1.weight", (64, 3, 7, 7), "conv_w")] + _bn(p + ".bn1", 64)
    cin = 64
    for li, planes in ((1, 64), (2, 128), (3, 256), (4, 512)):
        for b in (0, 1):
            q = "%s.layer%d.%d" % (p, li, b)
            s += [(q + ".conv1.weight", (planes, cin, 3, 3), "conv_w")] + _bn(q + ".bn1", planes)
            s += [(q + ".conv2.weight", (planes, planes, 3, 3), "conv_w")] + _bn(q + ".bn2", planes)
            if b == 0 and cin != planes:
                s += [(q + ".downsample.0.weight", (planes, cin, 1, 1), "conv_w")] + _bn(q + ".downsample.1", planes)
            cin = planes
    return s + [(p + ".fc.weight", (1000, 512), "lin_w"), (p + ".fc.bias", (1000,), "bias")]


def _resnet50(p="resnet"):
    """Bottleneck trunk (models/resnet.py:76-122, 261-270): expansion 4, a 1x1 downsample in every first block."""
    s = [(p + ".conv1.weight", (64, 3, 7, 7), "conv_w")] + _bn(p + ".bn1", 64)
    cin = 64
    for li, planes, blocks in zip((1, 2, 3, 4), (64, 128, 256, 512), RESNET50_BLOCKS):
        for b in range(blocks):
            q = "%s.layer%d.%d" % (p, li, b)
            s += [(q + ".conv1.weight", (planes, cin, 1, 1), "conv_w")] + _bn(q + ".bn1", planes)
            s += [(q + ".conv2.weight", (planes, planes, 3, 3), "conv_w")] + _bn(q + ".bn2", planes)
            s += [(q + ".conv3.weight", (4 * planes, planes, 1, 1), "conv_w")] + _bn(q + ".bn3", 4 * planes)
            if b == 0:
                s += [(q + ".downsample.0.weight", (4 * planes, cin, 1, 1), "conv_w")] + \
                     _bn(q + ".downsample.1", 4 * planes)
            cin = 4 * planes
    return s + [(p + ".fc.weight", (1000, 2048), "lin_w"), (p + ".fc.bias", (1000,), "bias")]


def _xformer(p, depth, heads, dim_head, mlp, cross):
    inner = heads * dim_head
    s = []
    for i in range(depth):
        a, f = "%s.layers.%d.0.fn" % (p, i), "%s.layers.%d.1.fn" % (p, i)
        s += [(a + ".norm.weight", (DIM,), "ln_w"), (a + ".norm.bias", (DIM,), "ln_b")]
        if cross:
            s += [(a + ".fn.to_%s.weight" % n, (inner, DIM), "lin_w") for n in "qkv"]
        else:
            s += [(a + ".fn.to_qkv.weight", (3 * inner, DIM), "lin_w")]
        s += [(a + ".fn.to_out.0.weight", (DIM, inner), "lin_w"), (a + ".fn.to_out.0.bias", (DIM,), "bias"),
              (f + ".norm.weight", (DIM,), "ln_w"), (f + ".norm.bias", (DIM,), "ln_b"),
              (f + ".fn.net.0.weight", (mlp, DIM), "lin_w"), (f + ".fn.net.0.bias", (mlp,), "bias"),
              (f + ".fn.net.3.weight", (DIM, mlp), "lin_w"), (f + ".fn.net.3.bias", (DIM,), "bias")]
    return s


def state_spec(net_G):
    """[(key, shape, role)] in the reference module's state_dict() order."""
    cfg = get_config(net_G)
    L = cfg["token_len"]
    if cfg["kind"] == "bit":
        r50 = cfg.get("backbone") == "resnet50"
        s = [("pos_embedding", (1, 2 * L, DIM), "pos")] + (_resnet50() if r50 else _resnet18())
        s += [("classifier.0.weight", (32, 32, 3, 3), "conv_w")] + _bn("classifier.1", 32)
        s += [("classifier.3.weight", (cfg["n_class"], 32, 3, 3), "conv_w"), ("classifier.3.bias", (cfg["n_class"],), "bias"),
              ("conv_pred.weight", (32, 1024 if r50 else 256, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias"),
              ("conv_a.weight", (L, 32, 1, 1), "conv_w")]
        s += _xformer("transformer", cfg["enc_depth"], BIT_HEADS, BIT_DIM_HEAD, 2 * DIM, False)
        s += _xformer("transformer_decoder", cfg["dec_depth"], BIT_HEADS, cfg["dec_dim_head"], 2 * DIM, True)
        return s
    if cfg["kind"] == "xbd":
        return _xbd(cfg)
    s = [("pos_embedding_%d" % l, (1, 2 * L, DIM), "pos") for l in (5, 4, 3, 2)]
    s += [("pos_embedding_decoder_%d" % l, (1, DIM, UNET_LEVELS[l]["size"], UNET_LEVELS[l]["size"]), "pos")
          for l in (5, 4, 3, 2)]
    s += _resnet18()
    s += [("conv_pred.weight", (32, 384, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias")]
    s += [("conv_squeeze_%d.0.weight" % l, (DIM, UNET_LEVELS[l]["cin"], 1, 1), "conv_w") for l in (5, 4, 3, 2)]
    s += [("conv_token_%d.weight" % l, (L, DIM, 1, 1), "conv_w") for l in (5, 4, 3, 2)]
    s += [("conv_decode_%d.weight" % l, (DIM, 2 * DIM, 3, 3), "conv_w") for l in (5, 4, 3, 2)]
    for l in (5, 4, 3, 2):
        lv = UNET_LEVELS[l]
        s += _xformer("transformer_%d" % l, cfg["enc_depth"], lv["heads"], lv["dim_head"], DIM, False)
        s += _xformer("transformer_decoder_%d" % l, lv["dec_depth"], lv["heads"], lv["dim_head"], DIM, True)
    s += [("conv_layer2_0.0.weight", (128, 128, 3, 3), "conv_w")] + _bn("conv_layer2_0.1", 128)
    s += [("conv_layer2_0.3.weight", (32, 128, 3, 3), "conv_w"), ("conv_layer2_0.3.bias", (32,), "bias")]
    for l in (2, 3, 4):
        s += [("conv_layer%d.0.weight" % l, (32, 32, 3, 3), "conv_w"), ("conv_layer%d.0.bias" % l, (32,), "bias")]
    return s + [("classifier.weight", (cfg["n_class"], 32, 3, 3), "conv_w"), ("classifier.bias", (cfg["n_class"],), "bias")]


def _xbd(cfg):
    """State dict of the xBD model.  Its nn.ModuleList holders (model_transformer_encoding.py:285-334) list the
    per-level modules in the order 2,3,4,5, so every such tensor appears twice: under its own name and as
    `<holder>.<index>...`; the second occurrence has role "alias:<own name>" and shares storage."""
    L = cfg["token_len"]
    order, held = (5, 4, 3, 2), (2, 3, 4, 5)
    s = [("pos_embedding_%d" % l, (1, 2 * L, DIM), "pos") for l in (5, 4, 3)]
    if cfg["decoder_pos"]:
        s += [("pos_embedding_decoder_%d" % l, (1, DIM, UNET_LEVELS[l]["size"], UNET_LEVELS[l]["size"]), "pos")
              for l in (5, 4, 3)]
    s += _resnet18() + [("conv_pred.weight", (32, 384, 3, 3), "conv_w"), ("conv_pred.bias", (32,), "bias")]

    def aliases(holder, own_prefix, entries_of):
        out = []
        for i, l in enumerate(held):
            pre = own_prefix % l
            out += [(holder + ".%d" % i + k[len(pre):], shp, "alias:" + k) for k, shp, _ in entries_of(l)]
        return out

    for own, holder, entries_of in (
            ("conv_squeeze_%d", "conv_squeeze_layers",
             lambda l: [("conv_squeeze_%d.0.weight" % l, (DIM, UNET_LEVELS[l]["cin"], 1, 1), "conv_w")]),
            ("conv_token_%d", "conv_tokens_layers", lambda l: [("conv_token_%d.weight" % l, (L, DIM, 1, 1), "conv_w")]),
            ("conv_decode_%d", "conv_decode_layers",
             lambda l: [("conv_decode_%d.weight" % l, (DIM, 2 * DIM, 3, 3), "conv_w")])):
        for l in order:
            s += entries_of(l)
        s += aliases(holder, own, entries_of)

    def enc(l):
        lv = UNET_LEVELS[l]
        return _xformer("transformer_%d" % l, cfg["enc_depth"], lv["heads"], lv["dim_head"], DIM, False)

    def dec(l):
        lv = UNET_LEVELS[l]
        return _xformer("transformer_decoder_%d" % l, lv["dec_depth"], lv["heads"], lv["dim_head"], DIM, True)

    for l in order:
        s += enc(l) + dec(l)
    s += aliases("transformer_layers", "transformer_%d", enc)
    s += aliases("transformer_decoder_layers", "transformer_decoder_%d", dec)
    s += [("conv_layer2_0.0.weight", (128, 128, 3, 3), "conv_w")] + _bn("conv_layer2_0.1", 128)
    s += [("conv_layer2_0.3.weight", (32, 128, 3, 3), "conv_w"), ("conv_layer2_0.3.bias", (32,), "bias")]
    for l in (2, 3, 4):
        s += [("conv_layer%d.0.weight" % l, (32, 32, 3, 3), "conv_w"), ("conv_layer%d.0.bias" % l, (32,), "bias")]
    return s + [("classifier.weight", (cfg["n_class"], 32, 3, 3), "conv_w"), ("classifier.bias", (cfg["n_class"],), "bias")]


def is_alias(role):
    return role.startswith("alias:")


def is_buffer(role):
    return role in ("bn_rm", "bn_rv", "bn_nbt")


def unused_prefixes(net_G):
    """Parameters the forward never touches, hence without gradient (SURVEY.md 8c 'Unused-param
    contract': 17 tensors for the BiT nets, 48 for newUNetTrans)."""
    cfg = get_config(net_G)
    pre = ["resnet.layer4.", "resnet.fc."]
    if cfg["kind"] == "unet":
        pre += ["conv_pred.", "conv_squeeze_2.", "conv_token_2.", "conv_decode_2.", "pos_embedding_2",
                "pos_embedding_decoder_2", "transformer_2.", "transformer_decoder_2."]
    if cfg["kind"] == "xbd":
        # only the level-5 call adds positional embeddings, and it picks the *_3 ones (layer index 3)
        pre += ["conv_pred.", "conv_squeeze_2.", "conv_token_2.", "conv_decode_2.", "transformer_2.",
                "transformer_decoder_2.", "pos_embedding_5", "pos_embedding_4", "pos_embedding_decoder_5",
                "pos_embedding_decoder_4"]
    return pre


def is_active(net_G, key):
    return not any(key.startswith(p) for p in unused_prefixes(net_G))
