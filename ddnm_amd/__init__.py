"""ddnm_amd -- MI355X-native (gfx950) engine for the DDNM sampling hot path.

Layout mirrors the slice of wyhuai/DDNM it replaces:
  ddnm_amd.functions.svd_ddnm        ddnm_diffusion            (functions/svd_ddnm.py)
  ddnm_amd.functions.svd_operators   A_functions classes       (functions/svd_operators.py)
  ddnm_amd.guided_diffusion.models   Model (CelebA-HQ UNet)    (guided_diffusion/models.py)
  ddnm_amd.guided_diffusion.diffusion Diffusion runner         (guided_diffusion/diffusion.py)
  ddnm_amd.csrc + include/ddnm_hip.h hand-written HIP kernels behind a C ABI
"""
__version__ = "0.1.0"
