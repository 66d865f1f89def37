"""Host-side mirror of the reference's `models` package for the change-detection hot path."""
from .networks import define_G, get_scheduler, init_net, init_weights, CDNet  # noqa: F401
