"""One tiny train step of the hot path on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke()).  The oracle is only the checker here."""
import types

import torch


def run(net_G="base_transformer_pos_s4", batch=2, size=64, verbose=True):
    import cdnet_ref as O                      # oracle (test infrastructure)
    from dahitra_amd.models import losses
    from dahitra_amd.models.networks import define_G
    from dahitra_amd.optim import AdamW

    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    cfg = O.get_config(net_G)
    sd = O.deterministic_state(net_G)
    a, b, lab = O.synthetic_batch(batch, size, n_class=cfg["n_class"])
    net = define_G(types.SimpleNamespace(net_G=net_G, compute_dtype="fp32"), gpu_ids=[0])
    net.load_state_dict(sd)
    net.train()
    opt = AdamW(net.parameters(), lr=0.01, betas=(0.9, 0.999), weight_decay=0.01)
    st = O.TrainState(net_G, sd, lr=0.01)
    for it in range(2):
        # teacher forcing: every step starts from the oracle's current weights / BN buffers, because an
        # Adam trajectory amplifies fp32-noise-level gradient differences (first update = lr*sign(g))
        net.load_state_dict({k: v.detach() for k, v in st.sd.items()})
        logits = net(a.cuda(), b.cuda())
        opt.zero_grad()
        loss = losses.focal_loss(logits, lab.cuda())
        loss.backward()
        opt.step()
        ref_logits, ref_loss = st.step(a, b, lab)
        err = float((logits.detach().cpu() - ref_logits).abs().max()) / float(ref_logits.abs().max())
        if verbose:
            print("smoke step %d: loss %.6f (oracle %.6f), logits rel err %.2e" % (it, float(loss), ref_loss, err))
        assert err < 1e-3, "logits differ from the oracle: %g" % err
        assert abs(float(loss) - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    torch.cuda.synchronize()
