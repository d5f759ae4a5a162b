#!/usr/bin/env python
"""Generate tests/golden/hq_demo.npz by running the REAL reference `hq_demo` sampler
(/root/reference/hq_demo, CPU) on small seeded cases.

    python tests/golden/make_golden_hq.py          # ~2-3 min of CPU in the build container

Own script (not part of make_golden.py) because hq_demo ships its own top-level `guided_diffusion` package,
which cannot share a process with the main reference's package of the same name.
Shims: stub `blobfile` / `torchvision` (imported, unused here), 'cuda' -> CPU, a noise tape for `randn_like`,
image writes redirected to a scratch directory.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
HQ = os.path.join(os.environ.get("DDNM_REFERENCE_ROOT", "/root/reference"), "hq_demo")

from oracle import cases, hq_cases, ref_import, weights  # noqa: E402


def load_reference():
    for name in ("blobfile", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, HQ)
    import importlib
    ns = types.SimpleNamespace()
    ns.gd = importlib.import_module("guided_diffusion.gaussian_diffusion")
    ns.script_util = importlib.import_module("guided_diffusion.script_util")
    ns.scheduler = importlib.import_module("guided_diffusion.scheduler")
    ns.respace = importlib.import_module("guided_diffusion.respace")
    ns.conf_base = importlib.import_module("conf_mgt.conf_base")
    sys.path.remove(HQ)
    return ns


def make_conf(ns, case):
    conf = ns.conf_base.Default_Conf()
    conf.update(hq_cases.conf_dict(case))
    return conf


def main():
    ns = load_reference()
    out = {}
    # ---- host logic: schedules and respaced tables
    for key, kw in hq_cases.SCHEDULES.items():
        out[f"schedule_{key}"] = np.array(ns.scheduler.get_schedule_jump(**kw))
    for key, (steps, resp) in hq_cases.RESPACINGS.items():
        out[f"space_{key}"] = np.array(sorted(ns.respace.space_timesteps(steps, resp)))
    scratch = tempfile.mkdtemp(prefix="hq_golden_")
    os.chdir(scratch)
    for name, case in hq_cases.CASES_ALL.items():
        conf = make_conf(ns, case)
        model, diffusion = ns.script_util.create_model_and_diffusion(
            **ns.script_util.select_args(conf, ns.script_util.model_and_diffusion_defaults().keys()), conf=conf)
        mcfg = hq_cases.model_config(case)
        model.load_state_dict(weights.adm_state_dict(mcfg, cases.SEED))
        model.eval()
        if name == list(hq_cases.CASES)[0]:
            out["betas_respaced"] = np.asarray(diffusion.betas)
            out["timestep_map"] = np.asarray(diffusion.timestep_map)
            out["posterior_variance"] = np.asarray(diffusion.posterior_variance)
            out["posterior_mean_coef1"] = np.asarray(diffusion.posterior_mean_coef1)
            out["posterior_mean_coef2"] = np.asarray(diffusion.posterior_mean_coef2)
        cond_fn = None
        if case.get("classifier"):
            import torch.nn.functional as F
            cc = hq_cases.classifier_config(case)
            classifier = ns.script_util.create_classifier(
                **ns.script_util.select_args(conf, ns.script_util.classifier_defaults().keys()))
            classifier.load_state_dict(weights.classifier_state_dict(cc))
            classifier.eval()

            def cond_fn(x, t, y=None, gt=None, **kwargs):           # hq_demo/main.py:87-94
                with torch.enable_grad():
                    x_in = x.detach().requires_grad_(True)
                    logits = classifier(x_in, t)
                    log_probs = F.log_softmax(logits, dim=-1)
                    selected = log_probs[range(len(logits)), y.view(-1)]
                    return torch.autograd.grad(selected.sum(), x_in)[0] * conf.classifier_scale

        def model_fn(x, t, y=None, gt=None, **kwargs):              # hq_demo/main.py:98-100
            return model(x, t, y if conf.class_cond else None, gt=gt)

        gt, x_init, tape = hq_cases.inputs(case)
        kwargs = {"gt": gt.clone(), "scale": case["scale"], "deg": case["deg"], "resize_y": case["resize_y"],
                  "sigma_y": case["sigma_y"], "save_path": name, "y": torch.full((1,), case["class"], dtype=torch.long)}
        if case.get("face"):
            kwargs["gt_keep_mask"] = hq_cases.keep_mask(case)
        with torch.no_grad(), ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
            res = diffusion.p_sample_loop_progressive(model_fn, (1, 3, 256, 256), noise=x_init.clone(),
                                                      clip_denoised=conf.clip_denoised, model_kwargs=kwargs,
                                                      cond_fn=cond_fn, device="cpu", progress=False, conf=conf)
        final = res["sample"]
        print(name, tuple(final.shape), float(final.abs().mean()))
        out[f"{name}_final"] = final.numpy()[:, :, ::4, ::4].copy()
        out[f"{name}_final_stats"] = np.array([final.double().mean().item(), final.double().std().item(),
                                               final.double().abs().sum().item()])
        out[f"{name}_last_sample"] = res["x0_t"].numpy()[:, :, ::8, ::8].copy()
    np.savez_compressed(os.path.join(HERE, "hq_demo.npz"), **out)


if __name__ == "__main__":
    main()
