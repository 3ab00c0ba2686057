from .builder import build_vision_tower  # noqa: F401
from .clip_encoder import CLIPVisionTower  # noqa: F401
