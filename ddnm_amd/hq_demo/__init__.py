"""MI355X engine for the reference's `hq_demo` (arbitrary-size restoration: DDPM-posterior DDNM sampler over
256x256 tiles with the mask-shift trick).  Module names follow hq_demo/guided_diffusion/."""
from .gaussian_diffusion import SpacedDiffusion, tile_plan  # noqa: F401
from .respace import space_timesteps  # noqa: F401
from .scheduler import get_schedule_jump  # noqa: F401
from .script_util import (classifier_defaults, create_classifier, create_model_and_diffusion,  # noqa: F401
                          model_and_diffusion_defaults, select_args)
