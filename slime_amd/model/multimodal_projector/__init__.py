from .builder import build_vision_projector, GatedBlock, HipMlp  # noqa: F401
