from dahitra_amd.data_config import DataConfig  # noqa: F401
