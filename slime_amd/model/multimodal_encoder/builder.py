"""``build_vision_tower`` -- same contract as llava/model/multimodal_encoder/builder.py:5-11."""
import os

from .clip_encoder import CLIPVisionTower, SYNTHETIC_PREFIX


def build_vision_tower(vision_tower_cfg, **kwargs):
    """Reads ``mm_vision_tower`` (else ``vision_tower``) from the config; accepts an existing path,
    hub-style names starting with ``openai`` / ``laion`` or containing ``ShareGPT4V`` (resolved from the
    local HF cache: there is no network here), plus the offline stand-in ``synthetic:<seed>``.
    Anything else raises ``ValueError('Unknown vision tower: ...')`` like the reference."""
    vision_tower = getattr(vision_tower_cfg, "mm_vision_tower", getattr(vision_tower_cfg, "vision_tower", None))
    if vision_tower is None:
        raise ValueError("Unknown vision tower: None")
    if (os.path.exists(vision_tower) or vision_tower.startswith("openai") or vision_tower.startswith("laion")
            or "ShareGPT4V" in vision_tower or vision_tower.startswith(SYNTHETIC_PREFIX)):
        return CLIPVisionTower(vision_tower, args=vision_tower_cfg, **kwargs)
    raise ValueError(f"Unknown vision tower: {vision_tower}")
