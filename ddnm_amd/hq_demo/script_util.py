"""Factories of hq_demo/guided_diffusion/script_util.py on the HIP engine (same names and keyword sets)."""
from ..guided_diffusion import classifier as _classifier
from ..guided_diffusion import unet as _unet
from . import gaussian_diffusion as gd
from .respace import space_timesteps

NUM_CLASSES = 1000


def diffusion_defaults():
    return dict(learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="", use_kl=False,
                predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False)


def classifier_defaults():
    return dict(image_size=64, classifier_use_fp16=False, classifier_width=128, classifier_depth=2,
                classifier_attention_resolutions="32,16,8", classifier_use_scale_shift_norm=True,
                classifier_resblock_updown=True, classifier_pool="attention")


def model_and_diffusion_defaults():
    res = dict(image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
               num_head_channels=-1, attention_resolutions="16,8", channel_mult="", dropout=0.0, class_cond=False,
               use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False, use_fp16=False,
               use_new_attention_order=False)
    res.update(diffusion_defaults())
    return res


def select_args(args_dict, keys):
    return {k: args_dict[k] for k in keys}


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks, channel_mult, num_heads,
                               num_head_channels, num_heads_upsample, attention_resolutions, dropout, diffusion_steps,
                               noise_schedule, timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm, resblock_updown, use_fp16,
                               use_new_attention_order, conf=None):
    """script_util.py:83-147: (noise predictor, respaced diffusion)."""
    model = _unet.create_model(image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
                               class_cond=class_cond, use_checkpoint=use_checkpoint,
                               attention_resolutions=attention_resolutions, num_heads=num_heads,
                               num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                               use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
                               use_fp16=use_fp16, use_new_attention_order=use_new_attention_order)
    diffusion = create_gaussian_diffusion(steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule,
                                          use_kl=use_kl, predict_xstart=predict_xstart,
                                          rescale_timesteps=rescale_timesteps,
                                          rescale_learned_sigmas=rescale_learned_sigmas,
                                          timestep_respacing=timestep_respacing, conf=conf)
    return model, diffusion


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear", use_kl=False,
                              predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
                              timestep_respacing="", conf=None):
    """script_util.py:258-306.  The sampler is epsilon-prediction only (every shipped config)."""
    if predict_xstart or rescale_timesteps:
        raise NotImplementedError("predict_xstart / rescale_timesteps are not used by the hq_demo configs")
    if conf is not None and conf.get("respace_interpolate"):
        raise NotImplementedError("respace_interpolate is not used by the hq_demo configs")
    betas = gd.get_named_beta_schedule(noise_schedule, steps, use_scale=True)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return gd.SpacedDiffusion(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas, conf=conf)


def create_classifier(image_size, classifier_use_fp16, classifier_width, classifier_depth,
                      classifier_attention_resolutions, classifier_use_scale_shift_norm, classifier_resblock_updown,
                      classifier_pool):
    return _classifier.create_classifier(image_size, classifier_use_fp16, classifier_width, classifier_depth,
                                         classifier_attention_resolutions, classifier_use_scale_shift_norm,
                                         classifier_resblock_updown, classifier_pool)
