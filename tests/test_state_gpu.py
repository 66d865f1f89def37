"""Training-state round trips through the C-ABI path: optimizer state_dict interchange with torch.optim.AdamW
(the reference resumes with optimizer_G.load_state_dict, models/trainer.py:116), CDTrainer resume, lifetime of the
buffers a captured HIP graph points into, several outstanding forwards."""
import types

import numpy as np
import pytest
import torch

import cdnet_ref as O

pytestmark = pytest.mark.gpu
NAME = "base_transformer_pos_s4"


def make_net(dtype="fp32", name=NAME):
    from dahitra_amd.models.networks import define_G
    net = define_G(types.SimpleNamespace(net_G=name, compute_dtype=dtype), gpu_ids=[0])
    net.load_state_dict(O.deterministic_state(name))
    return net.train()


def hip_step(net, opt, a, b, lab):
    from dahitra_amd.models import losses
    y = net(a, b)
    opt.zero_grad()
    loss = losses.focal_loss(y, lab)
    loss.backward()
    opt.step()
    return float(loss)


@pytest.mark.parametrize("capturable", [False, True])
def test_optimizer_state_dict_round_trips_and_matches_torch_adamw(capturable):
    """save after 2 steps -> load into a fresh optimizer -> step: identical to stepping on; the saved per-parameter
    state has torch.optim.AdamW's layout and the REAL step count; a torch.optim.AdamW fed the same gradients lands on
    the same parameters and moments."""
    from dahitra_amd.optim import AdamW
    a, b, lab = (t.cuda() for t in O.synthetic_batch(2, 64, seed=3))
    net = make_net()
    opt = AdamW(net.parameters(), lr=0.01, weight_decay=0.01, capturable=capturable)
    # a torch optimizer over clones, fed the HIP gradients
    names = [k for k, p in net.named_parameters()]
    ref_p = {k: p.detach().clone().requires_grad_(True) for k, p in net.named_parameters()}
    topt = torch.optim.AdamW([ref_p[k] for k in names], lr=0.01, betas=(0.9, 0.999), weight_decay=0.01)
    for _ in range(2):
        from dahitra_amd.models import losses
        y = net(a, b)
        opt.zero_grad()
        losses.focal_loss(y, lab).backward()
        for k, p in net.named_parameters():
            ref_p[k].grad = None if p.grad is None else p.grad.detach().clone()
        opt.step()
        topt.step()
    sd = opt.state_dict()
    tsd = topt.state_dict()
    assert sorted(sd["state"].keys()) == sorted(tsd["state"].keys())           # the same (grad-carrying) parameters
    for i, s in sd["state"].items():
        assert float(s["step"]) == 2.0
        assert set(("step", "exp_avg", "exp_avg_sq")) <= set(s.keys())
        t = tsd["state"][i]
        # (fused multiply-adds in the kernel vs torch's separate mul / addcmul: a few fp32 ulps)
        assert float((s["exp_avg"] - t["exp_avg"]).abs().max()) <= 1e-5 * float(t["exp_avg"].abs().max()) + 1e-12
        assert float((s["exp_avg_sq"] - t["exp_avg_sq"]).abs().max()) <= 1e-4 * float(t["exp_avg_sq"].abs().max()) + 1e-20
    for k, p in net.named_parameters():
        assert float((p.detach() - ref_p[k].detach()).abs().max()) <= 2e-6, k
    # ---- resume: fresh net + fresh optimizer from the saved dicts, against continuing with the original ------------
    import copy
    net_sd = copy.deepcopy(net.state_dict())
    opt_sd = copy.deepcopy(sd)
    net2 = make_net()
    net2.load_state_dict(net_sd)
    opt2 = AdamW(net2.parameters(), lr=0.01, weight_decay=0.01, capturable=capturable)
    opt2.load_state_dict(opt_sd)
    l1 = hip_step(net, opt, a, b, lab)
    l2 = hip_step(net2, opt2, a, b, lab)
    assert l1 == l2
    assert opt2.step_count(net2) == 3 and opt.step_count(net) == 3
    assert torch.equal(net._arena.flat, net2._arena.flat)
    # and loading a torch.optim.AdamW state (the reference's checkpoint) works the same way
    net3 = make_net()
    net3.load_state_dict(net_sd)
    opt3 = AdamW(net3.parameters(), lr=0.01, weight_decay=0.01, capturable=capturable)
    opt3.load_state_dict(copy.deepcopy(tsd))
    l3 = hip_step(net3, opt3, a, b, lab)
    assert abs(l3 - l1) <= 1e-6 * max(1.0, abs(l1))
    assert float((net3._arena.flat - net._arena.flat).abs().max()) <= 2e-6


def test_trainer_resumes_from_its_checkpoint(tmp_path):
    from dahitra_amd.models.trainer import CDTrainer
    a, b, lab = O.synthetic_batch(2, 64, seed=5)
    batch = {"A": a, "B": b, "L": lab}
    args = types.SimpleNamespace(net_G=NAME, gpu_ids=[0], lr=0.01, batch_size=2, max_epochs=4, n_class=2,
                                 lr_policy="linear", compute_dtype="fp32", checkpoint_dir=str(tmp_path))
    tr = CDTrainer(args, dataloaders={"train": [batch], "val": [batch]})
    tr.net_G.load_state_dict(O.deterministic_state(NAME))
    tr.max_num_epochs = 2
    tr.train_models()                                   # 2 epochs; the best (first, at least) epoch is saved
    ckpt = torch.load(tmp_path / "best_ckpt.pt", map_location="cpu")
    assert float(next(iter(ckpt["optimizer_G_state_dict"]["state"].values()))["step"]) == ckpt["epoch_id"] + 1
    tr2 = CDTrainer(args, dataloaders={"train": [batch], "val": [batch]})
    assert tr2._load_checkpoint()
    assert tr2.epoch_to_start == ckpt["epoch_id"] + 1 and tr2.best_val_acc == ckpt["best_val_acc"]
    assert tr2.exp_lr_scheduler_G.last_epoch == ckpt["exp_lr_scheduler_G_state_dict"]["last_epoch"]
    # one more step from the resumed state == one more step of a trainer that never stopped at that point
    tr3 = CDTrainer(args, dataloaders=None)
    tr3.net_G.load_state_dict(ckpt["model_G_state_dict"])
    tr3.optimizer_G.load_state_dict(ckpt["optimizer_G_state_dict"])
    tr3.exp_lr_scheduler_G.load_state_dict(ckpt["exp_lr_scheduler_G_state_dict"])
    tr2.net_G.train(), tr3.net_G.train()
    l2, l3 = float(tr2.train_step(batch)), float(tr3.train_step(batch))
    assert l2 == l3 and torch.equal(tr2.net_G._arena.flat, tr3.net_G._arena.flat)
    assert tr2.optimizer_G.step_count(tr2.net_G) == ckpt["epoch_id"] + 2


def test_graph_replay_survives_a_larger_eager_step_in_between():
    """the captured graph keeps raw pointers into the shared workspace / weight-gradient slabs / job tables: a bigger
    eager step afterwards must not free or rewrite what the graph reads (replay -> bigger eager step -> replay)"""
    from dahitra_amd.graph import GraphedTrainStep
    from dahitra_amd.optim import AdamW
    a, b, lab = (t.cuda() for t in O.synthetic_batch(2, 64, seed=11))
    big = tuple(t.cuda() for t in O.synthetic_batch(5, 128, seed=12))

    def run(disturb):
        net = make_net("bf16")
        opt = AdamW(net.parameters(), lr=0.002, capturable=True)
        step = GraphedTrainStep(net, opt, a, b, lab)
        ls = [float(step(a, b, lab))]
        if disturb:
            other = make_net("bf16")                                    # another net in the process, larger shapes
            oopt = AdamW(other.parameters(), lr=0.002)
            hip_step(other, oopt, *big)
            with torch.no_grad():
                net.eval()
                net(big[0], big[1])                                     # eval forward of the SAME net at a larger size
                net.train()
            junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]      # recycle freed blocks
            del junk
        ls.append(float(step(a, b, lab)))
        ls.append(float(step(a, b, lab)))
        return ls, net._arena.flat.clone()

    l0, f0 = run(False)
    l1, f1 = run(True)
    assert l0 == l1, (l0, l1)
    assert torch.equal(f0, f1)


def test_two_outstanding_forwards_and_a_no_grad_forward_between():
    """each grad-enabled forward owns its backward (autograd ctx), as the reference's autograd graph does"""
    from dahitra_amd.models import losses
    a1, b1, lab1 = (t.cuda() for t in O.synthetic_batch(2, 64, seed=21))
    a2, b2, lab2 = (t.cuda() for t in O.synthetic_batch(2, 64, seed=22))
    net = make_net()
    y1 = net(a1, b1)
    y2 = net(a2, b2)
    with torch.no_grad():
        net(a2, b2)                        # e.g. a metrics pass between forward and backward
    net.eval()
    with torch.no_grad():
        net(a1, b1)
    net.train()
    losses.focal_loss(y1, lab1).backward()
    g1 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    net.zero_grad(set_to_none=True)
    losses.focal_loss(y2, lab2).backward()
    g2 = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    with pytest.raises(RuntimeError):
        losses.focal_loss(y2, lab2).backward()            # its saved activations were released
    # references: one forward/backward at a time on a fresh net (BN running stats do not enter train-mode outputs)
    for (a, b, lab), got in (((a1, b1, lab1), g1), ((a2, b2, lab2), g2)):
        ref = make_net()
        losses.focal_loss(ref(a, b), lab).backward()
        for k, p in ref.named_parameters():
            if p.grad is not None:
                assert torch.equal(p.grad, got[k]), k


def test_graphed_step_keeps_an_optimizer_state_loaded_before_the_first_step():
    """resume order of the reference (models/trainer.py:113-116): load_state_dict() on a FRESH optimizer, then train.  The
    graphed step's warm-up snapshot / restore must carry the loaded moments and step count (they live only per parameter
    until the first step adopts them into the flat arenas)"""
    from dahitra_amd.graph import GraphedTrainStep
    from dahitra_amd.optim import AdamW
    a, b, lab = (t.cuda() for t in O.synthetic_batch(2, 64, seed=5))
    net = make_net()
    opt = AdamW(net.parameters(), lr=0.01, weight_decay=0.01, capturable=True)
    for _ in range(3):
        hip_step(net, opt, a, b, lab)
    import copy
    sd = copy.deepcopy(opt.state_dict())           # (state_dict() hands out the live moment views, as torch's does)
    params = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # continue eagerly for one step: the reference trajectory
    hip_step(net, opt, a, b, lab)
    want = {k: p.detach().clone() for k, p in net.named_parameters()}

    net2 = make_net()
    net2.load_state_dict(params)
    opt2 = AdamW(net2.parameters(), lr=0.01, weight_decay=0.01, capturable=True)
    opt2.load_state_dict(sd)                       # before any step: no flat state exists yet
    step = GraphedTrainStep(net2, opt2, a, b, lab)
    assert opt2.step_count(net2) == 3, "the warm-up restore lost the loaded step count"
    st = opt2._flat_state[id(net2)]
    ref = opt._flat_state[id(net)]
    assert float(st.m.abs().max()) > 0 and float(st.v.abs().max()) > 0, "the loaded Adam moments were zeroed"
    step(a, b, lab)
    assert opt2.step_count(net2) == 4
    for k, p in net2.named_parameters():
        if p.grad is None:
            continue
        d = float((p.detach() - want[k]).abs().max())
        assert d <= 1e-5 * max(float(want[k].abs().max()), 1e-3) + 1e-7, (k, d)
