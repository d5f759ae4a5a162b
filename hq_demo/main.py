#!/usr/bin/env python
"""`python hq_demo/main.py --resize_y --config confs/inet256.yml --path_y <image> --class 950 --deg sr_averagepooling
--scale 4 -i orange` -- the reference's arbitrary-size demo (hq_demo/main.py, hq_demo/evaluation.sh) on the MI355X
engine.  Same flags; results are written under results/<save_path>/{y,Apy,final,<tile>}/ like the reference.

Checkpoints are the public guided-diffusion files named by the YAML (`model_path`, `classifier_path`);
DDNM_RANDOM_WEIGHTS=1 substitutes seeded random weights (no network in the build environment).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ddnm_amd.hq_demo import (classifier_defaults, create_classifier, create_model_and_diffusion,  # noqa: E402
                              model_and_diffusion_defaults, select_args)
from ddnm_amd.hq_demo.conf import Default_Conf, yamlread  # noqa: E402


def load_image(path):
    """ToTensor + Normalize(0.5, 0.5) of hq_demo/main.py:118-125."""
    from PIL import Image
    arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1)[None].contiguous() * 2 - 1


def load_keep_mask(conf):
    """`gt_keep_mask` of the first evaluation item (image_datasets.py:154-181): RGB mask / 255, CHW."""
    ds = conf.eval_dataset()
    if not ds or not ds.get("mask_path") or not os.path.isdir(os.path.expanduser(ds["mask_path"])):
        return None
    from PIL import Image
    root = os.path.expanduser(ds["mask_path"])
    names = sorted(n for n in os.listdir(root) if n.split(".")[-1].lower() in ("jpg", "jpeg", "png", "gif"))
    if not names:
        return None
    size = ds.get("image_size", 256)
    img = Image.open(os.path.join(root, names[0])).convert("RGB")
    while min(*img.size) >= 2 * size:                                    # center_crop_arr (:198-218)
        img = img.resize(tuple(x // 2 for x in img.size), resample=Image.BOX)
    s = size / min(*img.size)
    img = img.resize(tuple(round(x * s) for x in img.size), resample=Image.BICUBIC)
    arr = np.array(img)
    cy, cx = (arr.shape[0] - size) // 2, (arr.shape[1] - size) // 2
    arr = arr[cy:cy + size, cx:cx + size].astype(np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1)[None].contiguous()


def _state_dict(path, random_from):
    if os.environ.get("DDNM_RANDOM_WEIGHTS") == "1":
        return random_from.random_state_dict(1234)
    return torch.load(os.path.expanduser(path), map_location="cpu")


def main(conf, args):
    print("Start", conf["name"])
    device = torch.device(conf.get("device") or "cuda")
    model, diffusion = create_model_and_diffusion(**select_args(conf, model_and_diffusion_defaults().keys()), conf=conf)
    model.load_state_dict(_state_dict(conf.model_path, model))
    if conf.use_fp16:
        model.convert_to_fp16()
    model.eval()
    cond_fn = None
    if (conf.classifier_scale or 0) > 0 and conf.classifier_path:
        print("loading classifier...")
        classifier = create_classifier(**select_args(conf, classifier_defaults().keys()))
        classifier.load_state_dict(_state_dict(conf.classifier_path, classifier))
        if conf.classifier_use_fp16:
            classifier.convert_to_fp16()
        classifier.eval()
        scale = float(conf.classifier_scale)

        def cond_fn(x, t, y=None, gt=None, **kwargs):
            assert y is not None
            g = classifier.log_prob_grad(x, t.float(), y)           # explicit backward on the HIP engine
            return g if scale == 1.0 else g * scale

    def model_fn(x, t, y=None, gt=None, **kwargs):
        assert y is not None
        return model(x, t, y) if conf.class_cond else model(x, t)

    print("sampling...")
    gt = load_image(args.get("path_y")).to(device)
    model_kwargs = {"gt": gt, "scale": args.get("scale"), "deg": args.get("deg"), "resize_y": args.get("resize_y"),
                    "sigma_y": args.get("sigma_y"), "save_path": args.get("save_path")}
    mask = load_keep_mask(conf)
    if mask is not None:
        model_kwargs["gt_keep_mask"] = mask.to(device)
    batch_size = gt.shape[0]
    model_kwargs["y"] = torch.ones(batch_size, dtype=torch.long, device=device) * args.get("class")
    result = diffusion.p_sample_loop(model_fn, (batch_size, 3, conf.image_size, conf.image_size),
                                     clip_denoised=conf.clip_denoised, model_kwargs=model_kwargs, cond_fn=cond_fn,
                                     device=device, progress=conf.show_progress, return_all=True, conf=conf)
    print("sampling complete")
    return result


def parse(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config", type=str, required=False, default="confs/inet256.yml")
    parser.add_argument("--deg", type=str, required=False, default="sr_averagepooling")
    parser.add_argument("--sigma_y", type=float, required=False, default=0.)
    parser.add_argument("-i", "--save_path", type=str, required=False, default="demo")
    parser.add_argument("--scale", type=int, required=False, default=4)
    parser.add_argument("--resize_y", default=False, action="store_true")
    parser.add_argument("--path_y", type=str, required=False, default="data/datasets/gts/inet256/orange.png")
    parser.add_argument("--class", type=int, required=False, default=950)
    return vars(parser.parse_args(argv))


if __name__ == "__main__":
    cli = parse()
    conf_arg = Default_Conf()
    conf_arg.update(yamlread(cli.get("config")))
    main(conf_arg, cli)
