"""MI355X-native SigLIP-2 vision-encoder path (hand-written gfx950 HIP kernels behind a C ABI).

Imported as ``siglip_amd`` via ``__graft_entry__.load_package()`` (the directory name has hyphens).
"""
from . import config, weights, lib, heads, optim, weights_io, preprocess  # noqa: F401
from .config import SiglipVisionConfig, get_config, NAMED_CONFIGS  # noqa: F401
from .encoder import (SiglipVisionModelHIP, OpenClipStyleEncoder, create_model_and_transforms,  # noqa: F401
                      VisionModelOutput)
from .ddp import GradBucketReducer  # noqa: F401
from .optim import FusedAdamW, global_grad_norm, ExponentialMovingAverage  # noqa: F401
