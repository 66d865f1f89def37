from dahitra_amd.models.networks import *  # noqa: F401,F403
from dahitra_amd.models.networks import define_G, get_scheduler, init_net, init_weights  # noqa: F401
