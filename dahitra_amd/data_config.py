"""Dataset roots by name -- the reference's data_config.py (DataConfig().get_data_config(name) -> .root_dir,
.label_transform).  DAHITRA_DATA_ROOT, if set, is prepended to the relative roots."""
import os


class DataConfig:
    data_name = ""
    root_dir = ""
    label_transform = "norm"

    _ROOTS = {"xBDataset": "data/xbd/", "quick_start": "samples", "LEVIR": "data/LEVIR_CD/"}

    def get_data_config(self, data_name):
        self.data_name = data_name
        if data_name not in self._ROOTS:
            raise TypeError('%s has not defined' % data_name)
        self.root_dir = os.path.join(os.environ.get("DAHITRA_DATA_ROOT", ""), self._ROOTS[data_name])
        return self
