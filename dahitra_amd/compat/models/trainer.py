"""`from models.trainer import *` (main_cd.py:3) must provide CDTrainer, utils and os, as the reference's module does."""
import os  # noqa: F401

from dahitra_amd import utils  # noqa: F401
from dahitra_amd.models.networks import *  # noqa: F401,F403
from dahitra_amd.models.trainer import CDTrainer  # noqa: F401
