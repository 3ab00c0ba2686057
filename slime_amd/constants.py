"""Model constants shared with the reference's token-splicing code (llava/constants.py:6-13)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"

# CLIP-ViT-L/14-336 geometry the slicer is hard-wired to (llava/process_image.py:11-21)
PATCH_SIZE = 14
PATCH_NUM_WIDTH = 24
PATCH_NUM_HEIGHT = 24
IMAGE_WIDTH = PATCH_SIZE * PATCH_NUM_WIDTH      # 336
IMAGE_HEIGHT = PATCH_SIZE * PATCH_NUM_HEIGHT    # 336

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
