"""Mutation checks of the gradient-parity CHECKERS (tests/_bounds.py), on the CPU: "green" must mean something.  The GPU tests
compare the HIP gradients with the oracle's through these helpers; here the helpers are fed the oracle's own gradients
 - displaced by the measured fp32 noise floor (what a correct second implementation looks like): must pass;
 - a weight gradient off by 1e-3 of its norm (what a wrong tile in one weight-gradient kernel looks like): the KERNEL-level
   checker (`close`: one kernel against torch on tie-free data) must reject it; at model level a flipped ReLU / max-pool tie moves
   the tensors around it by as much, so that checker tolerates a FEW such tensors and rejects more, or anything beyond tie noise;
 - with one gradient norm off by 1 %, one stored small tensor off by 1 % of its maximum: must fail."""
import os
import types

import numpy as np
import pytest
import torch

import cdnet_ref as O
import _bounds as B

NAME = "base_transformer_pos_s4"


@pytest.fixture(scope="module")
def oracle_grads(golden_dir):
    fl = B.floor(NAME, golden_dir)
    torch.set_num_threads(min(8, torch.get_num_threads()))
    a, b, lab = O.synthetic_batch(fl["case"]["batch"], fl["case"]["size"], seed=fl["case"]["seed"])
    st = O.TrainState(NAME, O.deterministic_state(NAME), lr=0.01)
    O.focal_loss(O.forward(st.sd, NAME, a, b, training=True), lab).backward()
    ref = {k: st.sd[k].grad.clone() for k in O.trainable_keys(NAME) if st.sd[k].grad is not None}
    return fl, ref


def _displaced(ref, fl, times):
    """every tensor moved by `times` x its own measured fp32-vs-fp64 distance, in a random direction"""
    g = torch.Generator().manual_seed(5)
    out = {}
    for k, r in ref.items():
        u = torch.randn(r.shape, generator=g, dtype=torch.float64)
        d = fl["per_tensor"].get(k, {"rel_l2": fl["rel_l2"]["median"]})["rel_l2"]
        out[k] = (r.double() + times * d * float(r.double().norm()) * u / u.norm()).float()
    return out


def test_gradients_at_the_noise_floor_pass(oracle_grads):
    fl, ref = oracle_grads
    B.assert_grads_at_floor(_displaced(ref, fl, 2.0), ref, fl, "oracle displaced by 2 x its floor")


def test_kernel_level_check_rejects_a_weight_gradient_off_by_1e_3(oracle_grads):
    """what the kernel tests (tests/test_kernels_gpu.py: test_conv2d_dgrad_and_wgrad, factor 4) apply to a weight gradient in
    fp32: a tensor off by 1e-3 -- in a random direction, or as a pure scale error -- is rejected; one that differs by fp32
    summation-order noise passes"""
    _, ref = oracle_grads
    want = ref["resnet.layer3.0.conv1.weight"]
    u = torch.randn(want.shape, generator=torch.Generator().manual_seed(6))
    B.close(want + 2e-6 * float(want.abs().max()) * u / u.abs().max(), want, torch.float32, "wgrad", factor=4.0)
    with pytest.raises(AssertionError, match="wgrad"):
        B.close(want + 1e-3 * float(want.norm()) * u / u.norm(), want, torch.float32, "wgrad", factor=4.0)
    with pytest.raises(AssertionError, match="wgrad"):
        B.close(want * (1.0 + 1e-3), want, torch.float32, "wgrad", factor=4.0)


def test_model_level_check_rejects_wrong_gradients_beyond_tie_noise(oracle_grads):
    """the oracle-based model check: tensors may sit at their own floor, a FEW (3) may carry one flipped tie's worth of noise
    (1e-3: which ties flip differs from run to run) -- more than a few tensors at 1e-3, or any tensor beyond the worst-tensor
    bound (a pure 1 % scale error leaves the cosine at 1: the distance must catch it), turn it red"""
    fl, ref = oracle_grads
    low = [k for k, v in fl["per_tensor"].items() if v["rel_l2"] < 1e-5 and k in ref and ref[k].numel() >= 64]
    assert len(low) >= 8

    def mutate(got, key, rel, seed):
        u = torch.randn(ref[key].shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
        got[key] = (got[key].double() + rel * float(ref[key].double().norm()) * u / u.norm()).float()
    got = _displaced(ref, fl, 2.0)
    for i, k in enumerate(low[:3]):                               # three tensors at 1e-3: what a flipped tie looks like
        mutate(got, k, 1e-3, 10 + i)
    r = B.assert_grads_at_floor(got, ref, fl, "three tie-sized outliers")
    assert len(r["outliers"]) == 3
    mutate(got, low[3], 1e-3, 20)                                 # a fourth: no longer "a few"
    with pytest.raises(AssertionError, match="beyond"):
        B.assert_grads_at_floor(got, ref, fl, "four outliers")
    got = _displaced(ref, fl, 2.0)
    got["resnet.layer3.1.conv2.weight"] = got["resnet.layer3.1.conv2.weight"] * 1.01
    with pytest.raises(AssertionError):
        B.assert_grads_at_floor(got, ref, fl, "scaled by 1 %")


def test_fixture_helpers_reject_a_one_percent_error(oracle_grads):
    fl, ref = oracle_grads
    params = {k: types.SimpleNamespace(grad=v.clone()) for k, v in ref.items()}
    keys = sorted(ref)
    vals = [float(ref[k].double().norm()) for k in keys]
    assert B.assert_grad_norms(params, keys, vals, 4e-3) == 0.0
    params[keys[3]].grad = params[keys[3]].grad * 1.01
    with pytest.raises(AssertionError, match="grad norm"):
        B.assert_grad_norms(params, keys, vals, 4e-3)
    params[keys[3]].grad = ref[keys[3]].clone()

    class Fix(dict):
        files = property(lambda self: list(self))
    small = [k for k in keys if ref[k].numel() <= 1024]
    fix = Fix({"grad0/" + k: ref[k].numpy() for k in small})
    n, worst = B.assert_stored_grads(params, fix, 3e-3)
    assert n == len(small) and worst["tie"] == 0.0 and worst["rest"] == 0.0
    k = next(k for k in small if not B.is_tie_sized(k, ref[k].numel()))          # e.g. the 2 x 32 x 3 x 3 class-head weight
    g = params[k].grad.clone().flatten()
    g[0] += 0.01 * float(ref[k].abs().max())
    params[k].grad = g.view_as(ref[k])
    with pytest.raises(AssertionError, match="grad " + k.replace(".", r"\.")):
        B.assert_stored_grads(params, fix, 3e-3)
    # ... while a tie-sized tensor may move by percents (one flipped ReLU tie), not by more
    params[k].grad = ref[k].clone()
    t = next(k for k in small if B.is_tie_sized(k, ref[k].numel()) and float(ref[k].abs().max()) > 0)
    g = params[t].grad.clone().flatten()
    g[0] += 0.03 * float(ref[t].abs().max())
    params[t].grad = g.view_as(ref[t])
    B.assert_stored_grads(params, fix, 3e-3)
    g[0] += 0.1 * float(ref[t].abs().max())
    with pytest.raises(AssertionError):
        B.assert_stored_grads(params, fix, 3e-3)
