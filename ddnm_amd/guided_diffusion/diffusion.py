"""Runner of the DDNM CLI on MI355X: `Diffusion(args, config).sample(simplified)`.

Same seam as the reference's `guided_diffusion/diffusion.py::Diffusion` (:79-610): it is built
from the argparse namespace + YAML namespace of `main.py`, constructs the noise predictor and the
degradation operator for `--deg`, forms `y = A(x_orig)`, draws `x_T`, runs the reverse loop and
reports PSNR.  The hot path underneath is ours: `ddnm_amd.guided_diffusion.models.Model`,
`ddnm_amd.functions.svd_operators.*`, `ddnm_amd.functions.svd_ddnm.ddnm_diffusion`.

Deliberate differences at the boundary (not in the results):
  * checkpoints are never downloaded (no network): `exp/logs/celeba/celeba_hq.ckpt` must exist,
    or `DDNM_RANDOM_WEIGHTS=1` selects seeded random weights of the same architecture;
  * images are read/written with PIL (torchvision is not required);
  * multi-GPU is one process per GPU (ddnm_amd.dist), not `torch.nn.DataParallel` (:140,164): every rank sees the
    same batches, restores the images [lo, hi) of each batch (its slice of x_T and of the per-step noise, both drawn
    for the WHOLE batch from a per-batch generator so the result does not depend on the number of ranks), and the
    restored shards meet in ONE RCCL all_gather per batch; rank 0 writes the PNGs and reports the PSNR.
"""
import os
import random

import numpy as np
import torch
import torch.utils.data as data

from .. import dist as ddist
from .. import ops
from ..functions.svd_ddnm import _AlphaTable, ddnm_diffusion, ddnm_plus_diffusion, get_schedule_jump
from ..functions.svd_operators import (Colorization, Denoising, Inpainting, SuperResolution, build_operator,
                                      mask_color_sr)
from .models import Model


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """float64 schedules of diffusion.py:46-76."""
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        s = np.linspace(-6, 6, n)
        betas = 1 / (np.exp(-s) + 1) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


class ImageFolder(data.Dataset):
    """torchvision.datasets.ImageFolder + ToTensor: classes = sorted sub-directories, files sorted inside each.
    transform "resize": Resize([S,S]) bilinear (datasets/__init__.py:138-150, CelebA_HQ / FFHQ);
    transform "center_crop_arr": the guided-diffusion crop (`out_of_dist` LSUN / ImageNet folders, :113-119,177-183)."""

    def __init__(self, root, image_size, transform="resize"):
        from PIL import Image  # noqa: F401
        self.size, self.transform = image_size, transform
        self.items = []
        classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        for ci, c in enumerate(classes):
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c))):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXT):
                        self.items.append((os.path.join(dirpath, f), ci))
        if not self.items:
            raise FileNotFoundError(f"no images under {root}")

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        path, cls = self.items[i]
        img = Image.open(path).convert("RGB")
        if self.transform == "center_crop_arr":
            arr = center_crop_arr(img, self.size)
        else:
            arr = np.asarray(img.resize((self.size, self.size), Image.BILINEAR), dtype=np.uint8)
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255.0)
        return x, cls


def center_crop_arr(pil_image, image_size=256):
    """datasets/__init__.py:29-44 (from openai/guided-diffusion): box-downsample while >= 2x, bicubic to the
    short side, centre crop."""
    from PIL import Image
    while min(*pil_image.size) >= 2 * image_size:
        pil_image = pil_image.resize(tuple(x // 2 for x in pil_image.size), resample=Image.BOX)
    scale = image_size / min(*pil_image.size)
    pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
    arr = np.array(pil_image)
    cy, cx = (arr.shape[0] - image_size) // 2, (arr.shape[1] - image_size) // 2
    return arr[cy:cy + image_size, cx:cx + image_size]


def center_crop_long_edge(pil_image):
    """datasets/imagenet_subset.py:5-23 (`CenterCropLongEdge`): torchvision's center_crop to min(w, h) -- the offsets
    are int(round((dim - s) / 2.0)) like torchvision.transforms.functional.center_crop."""
    w, h = pil_image.size
    s = min(w, h)
    left, top = int(round((w - s) / 2.0)), int(round((h - s) / 2.0))
    return pil_image.crop((left, top, left + s, top + s))


class ImageList(data.Dataset):
    """datasets/imagenet_subset.py::ImageDataset(normalize=False) (:48-103), what `subset_1k: true` of the ImageNet
    configs selects (datasets/__init__.py:169-175): list file of '<name> <label>' (label -1 when absent),
    CenterCropLongEdge + Resize(image_size) (bilinear) + ToTensor."""

    def __init__(self, root, list_file, image_size):
        self.root, self.size = root, image_size
        self.items = []
        with open(list_file) as f:
            for line in f:
                parts = line.rstrip().split()
                if parts:
                    self.items.append((parts[0], int(parts[1]) if len(parts) == 2 else -1))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        name, label = self.items[i]
        with open(self.root + "/" + name, "rb") as f:
            img = Image.open(f).convert("RGB")
        img = center_crop_long_edge(img).resize((self.size, self.size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
        return x, label


class SyntheticImages(data.Dataset):
    """Seeded U[0,1] images (BASELINE metric inputs; `--path_y synthetic:N`)."""

    def __init__(self, n, image_size, seed):
        self.n, self.size, self.seed = n, image_size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        return torch.rand(3, self.size, self.size, generator=g), 0


def save_image(x, path):
    """torchvision.utils.save_image for one [3,H,W] image in [0,1]."""
    from PIL import Image
    arr = x.detach().float().cpu().mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def data_transform(config, X):
    """datasets/__init__.py:201-216 for the configs on the hot path (rescaled: 2x - 1)."""
    if getattr(config.data, "uniform_dequantization", False) or getattr(config.data, "gaussian_dequantization", False):
        raise NotImplementedError("dequantization transforms are training-time options")
    if config.data.rescaled:
        return 2 * X - 1.0
    raise NotImplementedError("only rescaled data is on the DDNM hot path")


# functions/ckpt_util.py:15-24 (EMA checkpoints of the DDIM authors; `church_outdoor` -> `church`, :56-57)
LSUN_CKPT = {"bedroom": "ema_diffusion_lsun_bedroom_model/model-2388000.ckpt",
             "cat": "ema_diffusion_lsun_cat_model/model-1761000.ckpt",
             "church": "ema_diffusion_lsun_church_model/model-4432000.ckpt"}


def simple_checkpoint_path(config, exp):
    """Checkpoint of a `model.type: simple` config (guided_diffusion/diffusion.py:115-136): celeba_hq ->
    exp/logs/celeba/celeba_hq.ckpt; LSUN category -> the `ema_lsun_<category>` file get_ckpt_path resolves under
    $XDG_CACHE_HOME or exp/logs/ + diffusion_models_converted/ (ckpt_util.py:55-66).  Nothing is downloaded here."""
    ds = config.data.dataset
    if ds == "CelebA_HQ":
        return os.path.join(exp, "logs/celeba/celeba_hq.ckpt")
    if ds == "LSUN":
        cat = str(config.data.category).replace("church_outdoor", "church")
        if cat not in LSUN_CKPT:
            raise ValueError(f"no checkpoint known for LSUN category {config.data.category!r}")
        cachedir = os.environ.get("XDG_CACHE_HOME", os.path.join(exp, "logs/"))
        return os.path.join(cachedir, "diffusion_models_converted", LSUN_CKPT[cat])
    raise ValueError(f"no checkpoint family for dataset {ds!r} with model.type=simple")   # CIFAR10 is not a DDNM config


def _mix64(seed, k):
    """64-bit key of (run seed, batch index): splitmix64 finaliser, so that neighbouring seeds / batches share no key bits."""
    z = (int(seed) * 0x9E3779B97F4A7C15 + int(k) * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


class BatchNoise:
    """Noise of one loader batch, independent of how the batch is sharded: x_T and the tensor of loop iteration k are
    drawn for ALL `n` images from a generator seeded by (seed, batch index) -- in a fixed order -- and sliced to
    [lo, hi).  Indexing is sequential (`tape[k]` for k = 0, 1, ...), which is how `ddnm_diffusion(noise=...)` reads it."""

    def __init__(self, seed, batch_index, n, shape, lo, hi, device):
        self.g = torch.Generator(device=device)
        self.g.manual_seed((int(seed) * 1000003 + int(batch_index)) % (2 ** 63 - 1))
        self.full, self.lo, self.hi, self.device = (n,) + tuple(shape), lo, hi, device
        self.next_k = 0

    def draw(self, shape=None):
        return torch.randn(self.full if shape is None else shape, generator=self.g, device=self.device)

    def x_T(self):
        return self.draw()[self.lo:self.hi].contiguous()

    def __getitem__(self, k):
        if k != self.next_k:
            raise IndexError(f"BatchNoise is a sequential tape: asked for {k}, expected {self.next_k}")
        self.next_k += 1
        return self.draw()[self.lo:self.hi].contiguous()


class Diffusion(object):
    def __init__(self, args, config, device=None):
        self.args, self.config = args, config
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("ddnm_amd needs an MI355X: the hot path has no CPU fallback")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.model_var_type = config.model.var_type
        betas = get_beta_schedule(beta_schedule=config.diffusion.beta_schedule, beta_start=config.diffusion.beta_start,
                                  beta_end=config.diffusion.beta_end,
                                  num_diffusion_timesteps=config.diffusion.num_diffusion_timesteps)
        self.betas = torch.from_numpy(betas).float().to(self.device)
        self.num_timesteps = self.betas.shape[0]

    # ------------------------------------------------------------------ model construction
    def _build_model(self):
        cfg = self.config
        if cfg.model.type == "simple":
            model = Model(cfg, device=self.device)
            ckpt = simple_checkpoint_path(cfg, self.args.exp)
            if os.path.exists(ckpt):
                model.load_state_dict(torch.load(ckpt, map_location="cpu"))
            elif os.environ.get("DDNM_RANDOM_WEIGHTS") == "1":
                print(f"[ddnm_amd] {ckpt} not found; DDNM_RANDOM_WEIGHTS=1 -> seeded random weights")
                model.load_state_dict(model.random_state_dict(self.args.seed))
            else:
                raise FileNotFoundError(f"{ckpt} not found (no network here: place the checkpoint there, or set "
                                        "DDNM_RANDOM_WEIGHTS=1 for seeded random weights)")
            return model
        if cfg.model.type == "openai":
            from .unet import create_model
            model = create_model(**vars(cfg.model))
            model.device = self.device
            if cfg.model.use_fp16:
                model.convert_to_fp16()          # 3x3 torso convolutions on fp16 MFMA operands (unet.py here)
            if cfg.model.class_cond:
                ckpt = os.path.join(self.args.exp, "logs/imagenet/%dx%d_diffusion.pt" % (
                    cfg.data.image_size, cfg.data.image_size))
            else:
                ckpt = os.path.join(self.args.exp, "logs/imagenet/256x256_diffusion_uncond.pt")
            if os.path.exists(ckpt):
                model.load_state_dict(torch.load(ckpt, map_location="cpu"))
            elif os.environ.get("DDNM_RANDOM_WEIGHTS") == "1":
                print(f"[ddnm_amd] {ckpt} not found; DDNM_RANDOM_WEIGHTS=1 -> seeded random weights")
                model.load_state_dict(model.random_state_dict(self.args.seed))
            else:
                raise FileNotFoundError(f"{ckpt} not found (no network here: place the checkpoint there, or set "
                                        "DDNM_RANDOM_WEIGHTS=1 for seeded random weights)")
            if cfg.model.class_cond:
                # noisy classifier + guidance gradient (diffusion.py:166-191)
                from .classifier import args_to_dict, classifier_defaults, create_classifier, make_cond_fn
                ckpt = os.path.join(self.args.exp, "logs/imagenet/%dx%d_classifier.pt" % (
                    cfg.data.image_size, cfg.data.image_size))
                classifier = create_classifier(**args_to_dict(cfg.classifier, classifier_defaults().keys()))
                classifier.device = self.device
                if os.path.exists(ckpt):
                    classifier.load_state_dict(torch.load(ckpt, map_location="cpu"))
                elif os.environ.get("DDNM_RANDOM_WEIGHTS") == "1":
                    g = torch.Generator().manual_seed(self.args.seed + 1)
                    sd = {k: (torch.randn(v, generator=g) * (1.0 / max(1, int(np.prod(v[1:])))) ** 0.5 if len(v) > 1
                              else (1.0 + 0.1 * torch.randn(v, generator=g) if k.endswith("weight") else
                                    0.05 * torch.randn(v, generator=g)))
                          for k, v in classifier.state_dict_shapes().items()}
                    classifier.load_state_dict(sd)
                else:
                    raise FileNotFoundError(f"{ckpt} not found (set DDNM_RANDOM_WEIGHTS=1 for seeded random weights)")
                if cfg.classifier.classifier_use_fp16:
                    classifier.convert_to_fp16()
                self._cls_fn = make_cond_fn(classifier, cfg.classifier.classifier_scale)
            return model
        raise ValueError(cfg.model.type)

    def sample(self, simplified):
        model = self._build_model()
        # batch sizes of 1 and 2 (the reference's shipped `sampling.batch_size: 1`) replay the forward from a hipGraph:
        # eager, the host needs longer to enqueue a forward's ~250-300 launches than the GPU to run them
        # (DDNM_GRAPH_MAX_BATCH=0 switches it off; bench.py `latency` reports both)
        model.auto_graphs(int(os.environ.get("DDNM_GRAPH_MAX_BATCH", "2")))
        tt = self.config.time_travel
        print(("Run Simplified DDNM, without SVD." if simplified else "Run SVD-based DDNM."),
              f"{tt.T_sampling} sampling steps.", f"travel_length = {tt.travel_length},",
              f"travel_repeat = {tt.travel_repeat}.", f"Task: {self.args.deg}.")
        cls_fn = getattr(self, "_cls_fn", None)
        if simplified:
            self.simplified_ddnm_plus(model, cls_fn)
        else:
            self.svd_based_ddnm_plus(model, cls_fn)

    # ------------------------------------------------------------------ data
    def _loader(self):
        args, config = self.args, self.config
        if str(args.path_y).startswith("synthetic"):
            n = int(str(args.path_y).split(":")[1]) if ":" in str(args.path_y) else config.sampling.batch_size
            ds = SyntheticImages(n, config.data.image_size, args.seed)
        elif config.data.dataset in ("CelebA_HQ", "FFHQ"):
            if getattr(config.data, "out_of_dist", False):          # datasets/__init__.py:137-143
                ds = ImageFolder(os.path.join(args.exp, "datasets", "ood_celeba"), config.data.image_size)
            else:
                ds = ImageFolder(os.path.join(args.exp, "datasets", args.path_y), config.data.image_size)
                # datasets/__init__.py:152-167: indices shuffled with numpy seed 2019, all of them are "test"
                idx = list(range(len(ds)))
                state = np.random.get_state()
                np.random.seed(2019)
                np.random.shuffle(idx)
                np.random.set_state(state)
                ds = data.Subset(ds, idx)
        elif config.data.dataset == "LSUN":
            if getattr(config.data, "out_of_dist", False):          # datasets/__init__.py:112-119
                ds = ImageFolder(os.path.join(args.exp, "datasets", f"ood_{config.data.category}"), config.data.image_size,
                                 transform="center_crop_arr")
            else:
                raise NotImplementedError("the LSUN lmdb validation sets (datasets/__init__.py:120-134) need the `lmdb` "
                                          "package; use out_of_dist: true with an image folder exp/datasets/ood_<category>")
        elif config.data.dataset == "ImageNet":
            if getattr(config.data, "subset_1k", False):
                # datasets/__init__.py:169-175: exp/imagenet_val_1k.txt lists "<file> <label>" under exp/datasets/imagenet/imagenet
                ds = ImageList(os.path.join(args.exp, "datasets", "imagenet", "imagenet"),
                               os.path.join(args.exp, "imagenet_val_1k.txt"), config.data.image_size)
            elif getattr(config.data, "out_of_dist", False):        # :176-183
                ds = ImageFolder(os.path.join(args.exp, "datasets", "ood"), config.data.image_size,
                                 transform="center_crop_arr")
            else:
                raise NotImplementedError("torchvision.datasets.ImageNet(split='val') (datasets/__init__.py:184-190) is "
                                          "not rebuilt here; use subset_1k: true or out_of_dist: true")
        else:
            raise NotImplementedError(f"dataset {config.data.dataset}")
        if args.subset_start >= 0 and args.subset_end > 0:
            assert args.subset_end > args.subset_start
            ds = data.Subset(ds, range(args.subset_start, args.subset_end))
        else:
            args.subset_start, args.subset_end = 0, len(ds)
        print(f"Dataset has size {len(ds)}")

        def seed_worker(worker_id):
            s = args.seed % 2 ** 32
            np.random.seed(s)
            random.seed(s)

        g = torch.Generator()
        g.manual_seed(args.seed)
        return data.DataLoader(ds, batch_size=config.sampling.batch_size, shuffle=True, num_workers=0,
                               worker_init_fn=seed_worker, generator=g)

    # ------------------------------------------------------------------ SVD path (diffusion.py:419-610)
    def svd_based_ddnm_plus(self, model, cls_fn):
        args, config = self.args, self.config
        loader = self._loader()
        A_funcs = build_operator(args.deg, args.deg_scale, config, self.device)
        args.sigma_y = 2 * args.sigma_y          # scaling to [-1, 1] (:524)
        sigma_y = args.sigma_y
        rank, world = ddist.world_state()      # the INITIALISED group (env-only sharding without a group loses shards)
        print(f"Start from {args.subset_start}")
        idx_so_far = args.subset_start
        psnr_sum, n_done = 0.0, 0
        C, S = config.data.channels, config.data.image_size
        os.makedirs(os.path.join(args.image_folder, "Apy"), exist_ok=True)      # (rank 0 made the image folder before the first barrier)
        # Sharding (one process per GPU): a loader batch of at least `world` images is split by image index and gathered once;
        # SMALLER batches -- every shipped config has sampling.batch_size 1 -- are dealt whole, round-robin, each rank writing
        # its own images (file names carry the global image index) and the PSNR reduced at the end (splitting a
        # batch of one left N - 1 GPUs idle).  Either way an image's result does not depend on the rank count: its noise is a
        # function of (seed, batch index, image index in the batch) only.
        deal = world > 1 and config.sampling.batch_size < world
        philox = os.environ.get("DDNM_NOISE") != "torch"      # in-kernel Philox draws (default) | ATen tape (BatchNoise)
        for bi, (x_orig, classes) in enumerate(loader):
            b = x_orig.shape[0]
            if deal and bi % world != rank:
                idx_so_far += b
                continue
            # (split mode: every rank sees the same batch -- same loader seed; rank r restores images [lo, hi) of it)
            x_orig = data_transform(config, x_orig.to(self.device)).contiguous()
            lo, hi = (0, b) if deal else ddist.shard_range(b, rank, world)
            writer = deal or rank == 0
            if philox:
                # no ATen RNG launch in the loop, nothing drawn for other ranks' images: key = (seed, batch), counter =
                # (element, iteration, image index in the batch); bench.py times this path
                noise = ops.PhiloxNoise(_mix64(args.seed, bi), image_base=lo)
                x = noise.tensor(ops.PhiloxNoise.XT_ITER, torch.empty(max(hi - lo, 1), C, S, S, device=self.device))[:hi - lo]
            else:
                noise = BatchNoise(args.seed, bi, b, (C, S, S), lo, hi, self.device)
                x = noise.x_T()
            y = A_funcs.A(x_orig)                # the whole batch: operators are cheap, and the writer needs A^+ y of all
            if args.add_noise:
                if philox:
                    y = y + ops.PhiloxNoise(_mix64(args.seed, bi) ^ 0x5DEECE66D, 0).tensor(0, y.reshape(b, -1)).reshape(y.shape) * sigma_y
                else:
                    y = y + noise.draw(tuple(y.shape)) * sigma_y
            if writer:
                Apy = A_funcs.A_pinv(y).view(b, C, S, S)
                if args.deg[:6] == "deblur":
                    Apy = y.view(b, C, S, S)
                elif args.deg == "colorization":
                    Apy = y.view(b, 1, S, S).repeat(1, 3, 1, 1)
                elif args.deg == "inpainting":
                    Apy = Apy + A_funcs.A_pinv(A_funcs.A(torch.ones_like(Apy))).reshape(*Apy.shape) - 1
                for i in range(b):
                    save_image(ops.finalize_psnr(Apy[i:i + 1].contiguous())[0][0],
                               os.path.join(args.image_folder, f"Apy/Apy_{idx_so_far + i}.png"))
                    save_image(ops.finalize_psnr(x_orig[i:i + 1])[0][0],
                               os.path.join(args.image_folder, f"Apy/orig_{idx_so_far + i}.png"))
            if hi > lo:
                y_loc = y.reshape(b, -1)[lo:hi].contiguous()
                with torch.no_grad():
                    if sigma_y == 0.0:       # noise-free case, DDNM (diffusion.py:587-588)
                        xs, _ = ddnm_diffusion(x, model, self.betas, args.eta, A_funcs, y_loc, cls_fn=cls_fn,
                                               classes=classes, config=config, noise=noise, return_cpu=False)
                    else:                    # noisy case, DDNM+ (:589-590)
                        xs, _ = ddnm_plus_diffusion(x, model, self.betas, args.eta, A_funcs, y_loc, sigma_y, cls_fn=cls_fn,
                                                    classes=classes, config=config, noise=noise, return_cpu=False)
                x_loc = xs[0]
            else:
                x_loc = x                    # empty shard (fewer images than ranks): takes part in the gather only
            x_all = x_loc if deal else ddist.gather_images(x_loc, n_total=b)      # the path's single collective (one per batch)
            if writer:
                img, psnr = ops.finalize_psnr(x_all.contiguous(), x_orig)
                for j in range(b):
                    save_image(img[j], os.path.join(args.image_folder, f"{idx_so_far + j}_{0}.png"))
                psnr_sum += float(psnr.sum())
                n_done += b
                print("PSNR: %.2f" % (psnr_sum / n_done))
            idx_so_far += b
        if deal:
            psnr_sum, n_done = ddist.reduce_sum(psnr_sum, self.device), int(ddist.reduce_sum(n_done, self.device))
        if rank == 0:
            print("Total Average PSNR: %.2f" % (psnr_sum / max(n_done, 1)))
            print("Number of samples: %d" % n_done)
        ddist.barrier()
        return psnr_sum / max(n_done, 1)

    # ------------------------------------------------------------------ simplified path (diffusion.py:211-415)
    def _simplified_operator(self):
        args, config = self.args, self.config
        d, dev = config.data.image_size, self.device
        if args.deg == "colorization":
            return Colorization(d, dev, weights=(1 / 3, 1 / 3, 1 / 3))          # color2gray / gray2color :33-42
        if args.deg == "denoising":
            return Denoising(config.data.channels, d, dev)
        if args.deg == "sr_averagepooling":
            return SuperResolution(config.data.channels, d, round(args.deg_scale), dev)   # AdaptiveAvgPool2d / MeanUpsample
        if args.deg == "inpainting":
            mask = torch.from_numpy(np.load("exp/inp_masks/mask.npy")).reshape(-1)      # A = Ap = z * mask
            r = torch.nonzero(mask == 0).long().reshape(-1) * 3
            return Inpainting(config.data.channels, d, torch.cat([r, r + 1, r + 2], 0), dev)
        if args.deg in ("mask_color_sr", "diy"):                                         # :260-290
            mask = torch.from_numpy(np.load("exp/inp_masks/mask.npy"))
            return mask_color_sr(config.data.channels, d, mask, round(args.deg_scale), dev)
        raise NotImplementedError("degradation type not supported")

    def simplified_ddnm_plus(self, model, cls_fn):
        args, config = self.args, self.config
        loader = self._loader()
        print("args.deg:", args.deg)
        op = self._simplified_operator()
        args.sigma_y = 2 * args.sigma_y
        sigma_y = args.sigma_y
        rank, world = ddist.world_state()      # the INITIALISED group (env-only sharding without a group loses shards)
        print(f"Start from {args.subset_start}")
        idx_so_far = args.subset_start
        psnr_sum, n_done = 0.0, 0
        os.makedirs(os.path.join(args.image_folder, "Apy"), exist_ok=True)
        for bi, (x_orig, classes) in enumerate(loader):
            if config.sampling.batch_size != 1:
                raise ValueError("please change the config file to set batch size as 1")
            if bi % world != rank:               # batch size is 1 here (:308-309): whole images are dealt round-robin
                idx_so_far += x_orig.shape[0]
                continue
            x_orig = data_transform(config, x_orig.to(self.device)).contiguous()
            y = op.A(x_orig)
            Apy = op.A_pinv(y).view(*x_orig.shape)
            save_image(ops.finalize_psnr(Apy.contiguous())[0][0], os.path.join(args.image_folder, f"Apy/Apy_{idx_so_far}.png"))
            save_image(ops.finalize_psnr(x_orig)[0][0], os.path.join(args.image_folder, f"Apy/orig_{idx_so_far}.png"))
            x = torch.randn(y.shape[0], config.data.channels, config.data.image_size, config.data.image_size,
                            device=self.device)
            x = simplified_loop(x, model, self.betas, args.eta, op, y, sigma_y, config)
            img, psnr = ops.finalize_psnr(x, x_orig)
            # the reference names the file with the stale loop variable j = -1 (:402); reproduced
            save_image(img[0], os.path.join(args.image_folder, f"{idx_so_far + (-1)}_{0}.png"))
            psnr_sum += float(psnr[0])
            idx_so_far += y.shape[0]
            n_done += y.shape[0]
            print("PSNR: %.2f" % (psnr_sum / n_done))
        psnr_sum, n_done = ddist.reduce_sum(psnr_sum, self.device), int(ddist.reduce_sum(n_done, self.device))
        if rank == 0:
            print("Total Average PSNR: %.2f" % (psnr_sum / max(n_done, 1)))
            print("Number of samples: %d" % n_done)
        return psnr_sum / max(n_done, 1)


def simplified_loop(x, model, betas, eta, op, y, sigma_y, config, noise=None):
    """The loop inlined in the reference at diffusion.py:333-397: Eq. 19 lambda_t / gamma_t with
    sigma_t = sqrt(1 - alpha_bar'^2) (sic, :356) and the whole noise term scaled by gamma_t (:384)."""
    tt = config.time_travel
    skip = config.diffusion.num_diffusion_timesteps // tt.T_sampling
    times = get_schedule_jump(tt.T_sampling, tt.travel_length, tt.travel_repeat)
    alpha = _AlphaTable(betas)
    n = x.shape[0]
    y = y.reshape(n, -1).float().contiguous()
    xt = x.float().contiguous()
    x0_t = torch.empty_like(xt)
    bufs = [torch.empty_like(xt), torch.empty_like(xt)]
    if hasattr(op, "begin_run"):
        op.begin_run(y)
    with torch.no_grad():
        for k, (i, j) in enumerate(zip(times[:-1], times[1:])):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = alpha(j)
            out = bufs[k & 1]
            nz = torch.randn_like(xt) if noise is None else noise[k]
            if j < i:
                at = alpha(i)
                sigma_t = (1 - at_next ** 2).sqrt()
                et = model(xt, torch.full((n,), float(i), device=xt.device))
                if et.size(1) == 6:
                    et = et[:, :3]
                if sigma_t >= at_next * sigma_y:
                    lambda_t = 1.0
                    gamma_t = float((sigma_t ** 2 - (at_next * sigma_y) ** 2).sqrt())
                else:
                    lambda_t = float(sigma_t / (at_next * sigma_y))
                    gamma_t = 0.0
                s = ops.step_scalars(at, at_next, eta, lam=lambda_t, gamma=gamma_t)
                op.ddnm_step(xt, et, nz, y, s, x0_t, out)
            else:
                ops.renoise(x0_t, nz, float(at_next.sqrt()), float((1 - at_next).sqrt()), out=out)
            xt = out
    return xt
