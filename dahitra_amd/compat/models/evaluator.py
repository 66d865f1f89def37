"""`from models.evaluator import *` (eval_cd.py:3, main_cd.py:21) provides CDEvaluator, utils and os."""
import os  # noqa: F401

from dahitra_amd import utils  # noqa: F401
from dahitra_amd.models.evaluator import CDEvaluator, cm2score  # noqa: F401
from dahitra_amd.models.networks import *  # noqa: F401,F403
