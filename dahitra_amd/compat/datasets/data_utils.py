from dahitra_amd.datasets.data_utils import CDDataAugmentation  # noqa: F401
