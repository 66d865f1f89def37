from dahitra_amd.datasets.CD_dataset import *  # noqa: F401,F403
from dahitra_amd.datasets.CD_dataset import CDDataset, ImageDataset  # noqa: F401
