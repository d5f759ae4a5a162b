#!/usr/bin/env python
"""DDNM command line on MI355X -- same flags and YAML configs as the reference's main.py.

    python main.py --ni --config celeba_hq.yml --path_y celeba_hq --eta 0.85 \
                   --deg sr_bicubic --deg_scale 4 --sigma_y 0. -i celeba_sr_bc_4

Flag names, defaults and side effects follow the reference (main.py:22-147): the image folder is
`<exp>/image_samples/<-i>`, `--ni` overwrites it without asking, seeds are set for torch / numpy /
the device generator, and any exception inside the run is logged while the process still exits 0
(main.py:164-170).  Multi-GPU: like the reference, which takes every visible GPU from the plain command through
nn.DataParallel (guided_diffusion/diffusion.py:140,164,180), `python main.py ...` on a node with N > 1 visible GPUs
re-executes itself as N ranks (one process per GPU, `torch.distributed.run`; DDNM_GPUS=n picks another count,
DDNM_GPUS=1 stays in this process; loader batches smaller than the rank count -- the shipped batch_size 1 -- are dealt whole
to the ranks, larger ones are split by image and gathered); an explicit `python -m torch.distributed.run --nproc-per-node N main.py ...`
works as before.

`--path_y synthetic:N` (ours) replaces the dataset by N seeded uniform-noise images.
"""
import argparse
import logging
import os
import shutil
import sys
import traceback

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

FLAGS = [
    # (names, kwargs) in the order of the reference parser
    (("--config",), dict(type=str, required=True, help="YAML under configs/ (or an absolute path)")),
    (("--seed",), dict(type=int, default=1234, help="seed of torch / numpy / the device generator")),
    (("--exp",), dict(type=str, default="exp", help="root of the run directory (<exp>/image_samples, <exp>/datasets, <exp>/logs)")),
    (("--deg",), dict(type=str, required=True, help="degradation operator, e.g. sr_bicubic, colorization, inpainting, cs_walshhadamard, deblur_gauss")),
    (("--path_y",), dict(type=str, required=True, help="dataset sub-folder below <exp>/datasets, or synthetic:N")),
    (("--sigma_y",), dict(type=float, default=0.0, help="std of the measurement noise on the [0,1] scale (> 0 selects DDNM+)")),
    (("--eta",), dict(type=float, default=0.85, help="eta of the DDIM-style update")),
    (("--simplified",), dict(action="store_true", help="inlined A / A^+ loop of the reference (batch size 1)")),
    (("-i", "--image_folder"), dict(type=str, default="images", help="output folder below <exp>/image_samples")),
    (("--deg_scale",), dict(type=float, default=0.0, help="operator parameter: SR factor, CS ratio, ...")),
    (("--verbose",), dict(type=str, default="info", help="logging level name")),
    (("--ni",), dict(action="store_true", help="never prompt: an existing output folder is replaced")),
    (("--subset_start",), dict(type=int, default=-1)),
    (("--subset_end",), dict(type=int, default=-1)),
    (("-n", "--noise_type"), dict(type=str, default="gaussian", help="accepted for compatibility (unused, like in the reference)")),
    (("--add_noise",), dict(action="store_true")),
]


def to_namespace(tree):
    ns = argparse.Namespace()
    for key, val in tree.items():
        setattr(ns, key, to_namespace(val) if isinstance(val, dict) else val)
    return ns


def parse_args_and_config(argv=None):
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for names, kw in FLAGS:
        parser.add_argument(*names, **kw)
    args = parser.parse_args(argv)

    cfg_path = args.config if os.path.isabs(args.config) else os.path.join("configs", args.config)
    if not os.path.exists(cfg_path):
        cfg_path = os.path.join(HERE, "configs", args.config)
    with open(cfg_path, "r") as f:
        config = to_namespace(yaml.safe_load(f))

    level = getattr(logging, args.verbose.upper(), None)
    if not isinstance(level, int):
        raise ValueError("level {} not supported".format(args.verbose))
    handler = logging.StreamHandler()
    handler.setFormatter(logging.Formatter("%(levelname)s - %(filename)s - %(asctime)s - %(message)s"))
    root = logging.getLogger()
    root.addHandler(handler)
    root.setLevel(level)

    args.image_folder = os.path.join(args.exp, "image_samples", args.image_folder)
    if int(os.environ.get("WORLD_SIZE", 1)) == 1:        # single process: here, like the reference (main.py:112-135)
        if not prepare_image_folder(args, 0):
            sys.exit(0)

    device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    logging.info("Using device: {}".format(device))
    config.device = device

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(args.seed)
    return args, config


def prepare_image_folder(args, rank):
    """The reference's overwrite protocol (main.py:112-135), decided by rank 0 ALONE: the other ranks of a
    multi-process run never look at the folder (a rank that arrived after rank 0's makedirs used to see the fresh
    folder, print "Program halted" and leave rank 0 waiting in the first barrier) -- they learn the verdict through
    `ddist.broadcast_flag` once the process group exists.  Returns False when the run must stop."""
    if rank != 0:
        return True
    os.makedirs(os.path.join(args.exp, "image_samples"), exist_ok=True)
    if os.path.exists(args.image_folder):
        overwrite = args.ni
        if not overwrite and int(os.environ.get("WORLD_SIZE", 1)) > 1:
            print("Output image folder exists; pass --ni to overwrite it in a multi-process run. Program halted.")
            return False
        if not overwrite:
            answer = input(f"Image folder {args.image_folder} already exists. Overwrite? (Y/N)")
            overwrite = answer.upper() == "Y"
        if not overwrite:
            print("Output image folder exists. Program halted.")
            return False
        shutil.rmtree(args.image_folder)
    os.makedirs(args.image_folder)
    return True


def maybe_self_launch(argv):
    """Plain `python main.py ...` with several visible GPUs: re-exec as one rank per GPU (does not return then).  The
    arguments are parsed and the output-folder question is settled HERE, in the parent, before any rank exists: the
    interactive Y/N prompt of the reference (main.py:112-135) keeps working, and a refused overwrite spawns nothing."""
    if "WORLD_SIZE" in os.environ or not torch.cuda.is_available():
        return
    ndev = torch.cuda.device_count()
    n = int(os.environ.get("DDNM_GPUS", "0")) or ndev
    if n <= 1:
        return
    parser = argparse.ArgumentParser(add_help=False)
    for names, kw in FLAGS:
        parser.add_argument(*names, **kw)
    pre, _ = parser.parse_known_args(sys.argv[1:] if argv is None else argv)
    pre.image_folder = os.path.join(pre.exp, "image_samples", pre.image_folder)
    extra = []
    if os.path.exists(pre.image_folder) and not pre.ni:
        answer = input(f"Image folder {pre.image_folder} already exists. Overwrite? (Y/N)")
        if answer.upper() != "Y":
            print("Output image folder exists. Program halted.")
            sys.exit(0)
        extra = ["--ni"]                      # the answer travels to the ranks
    if n > ndev and os.environ.get("DDNM_DIST_BACKEND") != "gloo":      # gloo: the 1-GPU test mode, ranks share the device
        sys.stderr.write(f"[main] DDNM_GPUS={n} but only {ndev} GPU(s) are visible\n")
        sys.exit(2)
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv) + extra
    sys.exit(subprocess.call(cmd))


def main(argv=None):
    maybe_self_launch(argv)
    args, config = parse_args_and_config(argv)
    try:
        from ddnm_amd import dist as ddist
        from ddnm_amd.guided_diffusion.diffusion import Diffusion
        rank, _, world = ddist.init()
        if world > 1:
            # rank 0's verdict on the output folder; the broadcast is also the "folder exists" barrier
            if not ddist.broadcast_flag(prepare_image_folder(args, rank)):
                ddist.shutdown()
                return 0
        runner = Diffusion(args, config)
        runner.sample(args.simplified)
    except Exception:
        logging.error(traceback.format_exc())
    return 0


if __name__ == "__main__":
    sys.exit(main())
