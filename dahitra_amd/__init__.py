"""dahitra_amd -- MI355X (gfx950) native implementation of the DAHiTra change-detection hot path.

    from dahitra_amd.models.networks import define_G      # reference: models/networks.py:130
    from dahitra_amd.models import losses                 # reference: models/losses.py
    from dahitra_amd.models.trainer import CDTrainer      # reference: models/trainer.py
    from dahitra_amd.optim import AdamW

All arithmetic runs in hand-written HIP kernels (dahitra_amd/csrc, C ABI in include/dahitra_hip.h).
"""
__version__ = "0.1.0"
