from dahitra_amd.utils import de_norm, get_device, get_loader, get_loaders, make_numpy_grid  # noqa: F401
