from dahitra_amd.misc.metric_tool import *  # noqa: F401,F403
