from .builder import build_vision_sampler, TextGuidedSampler  # noqa: F401
from .sampler import Resampler, IdentityMap  # noqa: F401
