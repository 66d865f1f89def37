from dahitra_amd.misc.logger_tool import Logger, Timer  # noqa: F401
