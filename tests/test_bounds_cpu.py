"""Mutation checks of the gradient-parity CHECKERS (tests/_bounds.py), on the CPU: "green" must mean something.  The GPU tests
compare the HIP gradients with the oracle's through these helpers; here the helpers are fed the oracle's own gradients
 - displaced by the measured fp32 noise floor (what a correct second implementation looks like): must pass;
 - with ONE convolution weight gradient off by 1e-3 of its norm (what a wrong tile in one weight-gradient kernel looks like):
   must fail, naming that tensor (a tensor downstream of the case's flipping ties: upstream of them the floor itself is 3e-4);
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


def test_one_weight_gradient_off_by_1e_3_turns_the_check_red(oracle_grads):
    fl, ref = oracle_grads
    got = _displaced(ref, fl, 2.0)
    key = "resnet.layer3.0.conv1.weight"
    assert fl["per_tensor"][key]["rel_l2"] < 1e-4                 # a tensor whose own floor is far below the mutation
    u = torch.randn(ref[key].shape, generator=torch.Generator().manual_seed(6), dtype=torch.float64)
    got[key] = (got[key].double() + 1e-3 * float(ref[key].double().norm()) * u / u.norm()).float()
    with pytest.raises(AssertionError, match="resnet.layer3.0.conv1.weight"):
        B.assert_grads_at_floor(got, ref, fl, "mutated")


def test_a_scaled_weight_gradient_turns_the_check_red(oracle_grads):
    """a pure scale error leaves the cosine at 1: the relative distance must catch it"""
    fl, ref = oracle_grads
    got = _displaced(ref, fl, 2.0)
    got["resnet.layer3.1.conv2.weight"] = got["resnet.layer3.1.conv2.weight"] * (1.0 + 1e-3)
    with pytest.raises(AssertionError):
        B.assert_grads_at_floor(got, ref, fl, "scaled")


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
