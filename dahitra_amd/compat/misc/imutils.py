from dahitra_amd.misc.imutils import save_image  # noqa: F401
