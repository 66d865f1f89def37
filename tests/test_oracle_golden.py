"""The oracle (oracle/cdnet_ref.py) against fixtures produced by the reference itself
(tests/golden/*.npz, generator oracle/make_golden.py).  Runs anywhere, CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import cdnet_ref as O

FWD = ["base_transformer_pos_s4", "base_transformer_pos_s4_dd8", "base_transformer_pos_s4_dd8_o5",
       "base_transformer_pos_s4_dd8_dedim8", "base_transformer_pos_s4_dd8_t8_e2d4", "newUNetTrans",
       "base_transformer_pos_s4_resnet50"]


def test_state_keys_match_reference(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for name in FWD:
        spec = O.state_spec(name)
        assert [[k, list(s)] for k, s, _ in spec] == ref[name], name


def test_param_counts():
    # SURVEY.md section 8 a1 (probed from the reference)
    want = {"base_transformer_pos_s4": 11913290, "base_transformer_pos_s4_dd8": 12402506,
            "base_transformer_pos_s4_dd8_dedim8": 11943754, "newUNetTrans": 13381226,
            "base_transformer_pos_s4_resnet50": 26001994}
    for name, n in want.items():
        got = sum(int(np.prod(s)) if len(s) else 1 for k, s, r in O.state_spec(name)
                  if not O.is_buffer(r))
        assert got == n, (name, got)


def test_deterministic_state_is_rng_independent():
    torch.manual_seed(123)
    a = O.deterministic_state("base_transformer_pos_s4")
    torch.manual_seed(999)
    b = O.deterministic_state("base_transformer_pos_s4")
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert abs(float(a["conv_a.weight"].std()) - (3.0 / 32) ** 0.5 / 3 ** 0.5) < 0.05


@pytest.mark.parametrize("name", FWD)
def test_forward_matches_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, "fwd_%s.npz" % name))
    cfg = O.get_config(name)
    bs, size, stride = int(g["batch"]), int(g["size"]), int(g["stride"])
    a, b, lab = O.synthetic_batch(bs, size, n_class=cfg["n_class"])
    for mode in ("eval", "train"):
        sd = O.deterministic_state(name)
        with torch.no_grad():
            y = O.forward(sd, name, a, b, training=(mode == "train"))
        want = torch.from_numpy(g["logits_" + mode])
        got = y[..., ::stride, ::stride]
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 1e-5 * scale, (name, mode)
        assert abs(float(y.double().sum()) - float(g["sum_" + mode])) <= 1e-6 * float(g["abssum_" + mode])
        mask = torch.argmax(y, 1).numpy().astype(np.uint8)
        packed = np.packbits(mask) if cfg["n_class"] == 2 else mask
        # masks must agree wherever the reference margin exceeds the fp32 tie band
        if cfg["n_class"] == 2:
            diff = np.unpackbits(packed)[:mask.size] != np.unpackbits(g["mask_" + mode])[:mask.size]
        else:
            diff = (packed.ravel() != np.asarray(g["mask_" + mode]).ravel())
        assert diff.sum() == 0, (name, mode, int(diff.sum()))
        if mode == "train":
            assert np.allclose(sd["resnet.bn1.running_mean"].numpy(), g["bn1_running_mean"], atol=1e-6)
            assert np.allclose(sd["resnet.bn1.running_var"].numpy(), g["bn1_running_var"], atol=1e-6)
            assert abs(float(O.focal_loss(y, lab)) - float(g["focal"])) < 1e-6


@pytest.mark.parametrize("name", ["base_transformer_pos_s4", "newUNetTrans", "base_transformer_pos_s4_resnet50"])
def test_train_steps_match_golden(name, golden_dir):
    torch.set_num_threads(8)
    g = np.load(os.path.join(golden_dir, "train_%s.npz" % name))
    cfg = O.get_config(name)
    a, b, lab = O.synthetic_batch(int(g["batch"]), int(g["size"]), n_class=cfg["n_class"])
    st = O.TrainState(name, O.deterministic_state(name), lr=float(g["lr"]))
    losses = []
    for it in range(int(g["steps"])):
        if it == 0:
            logits = O.forward(st.sd, name, a, b, training=True)
            loss = O.focal_loss(logits, lab)
            loss.backward()
            nograd = sorted(k for k in O.trainable_keys(name) if st.sd[k].grad is None)
            assert nograd == sorted(g["nograd_keys"].tolist())
            for k, v in zip(g["gradnorm_keys"].tolist(), g["gradnorm_vals"].tolist()):
                got = float(st.sd[k].grad.double().norm())
                assert abs(got - v) <= 2e-4 * max(v, 1e-8) + 1e-9, (k, got, v)
            for k in g.files:
                if k.startswith("grad0/"):
                    w = torch.from_numpy(g[k])
                    e = float((st.sd[k[6:]].grad - w).abs().max())
                    assert e <= 2e-4 * float(w.abs().max()) + 1e-8, (k, e)
            # rebuild so that BN buffers / step count start from scratch
            st = O.TrainState(name, O.deterministic_state(name), lr=float(g["lr"]))
        _, l = st.step(a, b, lab)
        losses.append(l)
    assert np.allclose(losses, g["losses"], rtol=2e-4, atol=1e-7), (losses, g["losses"])
    for k, v in zip(g["finalnorm_keys"].tolist(), g["finalnorm_vals"].tolist()):
        got = float(st.sd[k].detach().double().norm())
        assert abs(got - v) <= 1e-4 * max(v, 1e-8) + 1e-7, (k, got, v)
    for k in g.files:
        if k.startswith("final/"):
            w = torch.from_numpy(g[k])
            e = float((st.sd[k[6:]].detach() - w).abs().max())
            assert e <= 5e-4 * float(w.abs().max()) + 1e-7, (k, e)
    assert int(st.sd["resnet.bn1.num_batches_tracked"]) == int(g["nbt"])


def test_focal_loss_small_known_answer():
    # hand-computed: two classes, one pixel, logits (0, 0), label 1 -> p = .5
    # loss = (1+1e-6) * 0.5*0.25*ln2 + 1e-6 * 0.5*0.25*ln2
    logits = torch.zeros(1, 2, 1, 1)
    lab = torch.ones(1, 1, 1, 1, dtype=torch.int64)
    want = (1 + 2e-6) * 0.125 * np.log(2.0)
    assert abs(float(O.focal_loss(logits, lab)) - want) < 1e-7


@pytest.mark.parametrize("name", ["xbd_unet_transformer_nodecpos"])
def test_xbd_train_steps_match_golden(name, golden_dir):
    """xBD 5-class step: forward, ComboLoss, clip, hand-rolled AdamW against the fixture the
    reference produced (oracle/make_golden.py XBD_CASES).  The 1024x1024 fixture is checked on the GPU side
    (tests/test_xbd_gpu.py) and, in the build container, in tests/test_oracle_vs_reference.py."""
    g = np.load(os.path.join(golden_dir, "xbd_%s.npz" % name))
    bs, size, stride = int(g["batch"]), int(g["size"]), int(g["stride"])
    a, b, lab = O.synthetic_batch(bs, size, seed=11, n_class=5)
    x6, msk = torch.cat([a, b], 1), O.xbd_masks(lab)
    with torch.no_grad():
        y = O.forward(O.deterministic_state(name), name, x6, None, training=False)
    assert float((y[..., ::stride, ::stride] - torch.from_numpy(g["logits_eval"])).abs().max()) <= 1e-5
    st = O.XbdTrainState(name, O.deterministic_state(name), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
    losses, norms = [], []
    for it in range(int(g["steps"])):
        yo, lo = st.step(x6, msk)
        if it == 0:
            assert float((yo[..., ::stride, ::stride] - torch.from_numpy(g["logits_train"])).abs().max()) <= 1e-5
            assert sorted(k for k in O.trainable_keys(name) if st.sd[k].grad is None) == sorted(g["nograd_keys"].tolist())
        losses.append(lo)
        norms.append(st.last_total_norm)
    assert np.allclose(losses, g["losses"], rtol=1e-5), (losses, g["losses"])
    assert np.allclose(norms, g["total_norms"], rtol=1e-4), (norms, g["total_norms"])
    for k, v in zip(g["finalnorm_keys"].tolist(), g["finalnorm_vals"].tolist()):
        assert abs(float(st.sd[k].double().norm()) - v) <= 1e-6 * max(v, 1e-8) + 1e-9, k


def test_xbd_state_has_the_modulelist_aliases(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "state_keys.json")))
    for name in ("xbd_unet_transformer", "xbd_unet_transformer_nodecpos"):
        spec = O.state_spec(name)
        assert [[k, list(s)] for k, s, _ in spec] == ref[name], name
        sd = O.deterministic_state(name)
        assert sd["conv_squeeze_layers.3.0.weight"] is sd["conv_squeeze_5.0.weight"]
        assert sd["transformer_decoder_layers.1.layers.7.1.fn.fn.net.3.bias"] is \
            sd["transformer_decoder_3.layers.7.1.fn.fn.net.3.bias"]
    n = sum(int(np.prod(s)) for k, s, r in O.state_spec("xbd_unet_transformer") if not O.is_buffer(r) and not O.is_alias(r))
    assert n == 13250765


@pytest.mark.parametrize("name", ["base_transformer_pos_s4", "newUNetTrans"])
def test_large_margin_fixture_matches_oracle(name, golden_dir):
    """the antisymmetric-head fixture (written by the reference, oracle/make_golden.py MARGIN_CASES): the oracle gives
    the same masks at EVERY pixel and the stored per-pixel margins put < 0.2 % of them in the 4e-4 tie band"""
    g = np.load(os.path.join(golden_dir, "margin_%s.npz" % name))
    a, b, _ = O.synthetic_batch(int(g["batch"]), int(g["size"]), seed=int(g["seed"]))
    for mode in ("eval", "train"):
        with torch.no_grad():
            y = O.forward(O.large_margin_state(name), name, a, b, training=(mode == "train"))
        mask = torch.argmax(y, 1).numpy().astype(np.uint8)
        assert np.array_equal(np.packbits(mask), g["mask_" + mode])
        margin = (y[:, 0] - y[:, 1]).abs().numpy()
        assert np.allclose(margin, g["margin_" + mode].astype(np.float32), rtol=2e-3, atol=1e-6)
        assert float((g["margin_" + mode].astype(np.float32) <= 4e-4 * float(g["scale_" + mode])).mean()) < 0.002
        assert abs(float(y.double().sum()) - float(g["sum_" + mode])) <= 1e-6 * float(g["abssum_" + mode])


def test_resnet50_1024_fixture_matches_oracle_eval(golden_dir):
    """BASELINE configs[4] shape (ResNet-50 trunk, 1024x1024, written by the reference): the oracle's eval forward"""
    g = np.load(os.path.join(golden_dir, "fwd1024_base_transformer_pos_s4_resnet50.npz"))
    name = "base_transformer_pos_s4_resnet50"
    a, b, _ = O.synthetic_batch(int(g["batch"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        y = O.forward(O.deterministic_state(name), name, a, b, training=False)
    st = int(g["stride"])
    assert float((y[..., ::st, ::st] - torch.from_numpy(g["logits_eval"])).abs().max()) <= 1e-5 * float(g["scale_eval"])
    assert abs(float(y.double().sum()) - float(g["sum_eval"])) <= 1e-6 * float(g["abssum_eval"])


def test_committed_gradient_noise_floor_is_reproducible(golden_dir):
    """tests/golden/grad_noise_floor.json (tools/grad_noise_floor.py: oracle gradients, float32 against float64) is what the GPU
    gradient test takes its bounds from: re-measured here for base_transformer_pos_s4.  The exact figures move with the
    thread count (summation order), hence a factor-3 window; and the floor of this well-conditioned fixture is small --
    nowhere near the 6e-2 per-tensor bound the gradient tests used before it was measured."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import grad_noise_floor as G
    name = "base_transformer_pos_s4"
    want = json.load(open(os.path.join(golden_dir, "grad_noise_floor.json")))
    assert set(G.NETS) <= set(want)
    got = G.measure(name, threads=4)
    for key in ("rel_l2", "rel_max"):
        for q in ("median", "max"):
            a, b = got[key][q], want[name][key][q]
            assert b / 3 <= a <= 3 * b, (key, q, a, b)
    assert got["tensors"] == want[name]["tensors"] and got["case"] == want[name]["case"]
    assert want[name]["rel_l2"]["max"] < 5e-3 and want[name]["rel_l2"]["median"] < 1e-4
    assert got["cos_min"] > 0.99999
