from dahitra_amd.models.losses import cross_entropy, diceloss, focal_loss  # noqa: F401
