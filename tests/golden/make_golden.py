#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from
/root/reference, CPU) on the seeded cases of oracle/cases.py.

Run once in the build container:   python tests/golden/make_golden.py [--full]
The reference ships no golden vectors of its own, so these
files are what pins the oracle restatement -- and, through it, the HIP engine --
to the reference's behaviour.  `--full` additionally runs the 100-step
celeba_hq / sr_bicubic 4x case (BASELINE config 2 at B=1, about 2-3 min of CPU).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import cases, ref_import, schedule  # noqa: E402

OPS = ["sr_averagepooling", "sr_bicubic", "colorization", "inpainting", "cs_walshhadamard", "denoising"]


def ref_operator(R, name, d, mask=None, ratio=4):
    from oracle import operators as O
    if name == "sr_averagepooling":
        return R.SuperResolution(3, d, ratio, "cpu")
    if name == "sr_bicubic":
        k = O.bicubic_kernel(4)            # restates diffusion.py:485-499 (checked below)
        return R.SRConv(k / k.sum(), 3, d, "cpu", stride=4)
    if name == "colorization":
        return R.Colorization(d, "cpu")
    if name == "inpainting":
        mask = cases.random_mask(d) if mask is None else mask
        m = mask.reshape(-1)
        r = torch.nonzero(m == 0).long().reshape(-1) * 3          # diffusion.py:465-470
        op = R.Inpainting.__new__(R.Inpainting)                    # skip the O(n*m) python loop (:330)
        op.channels, op.img_dim = 3, d
        missing = torch.cat([r, r + 1, r + 2], dim=0)
        op._singulars = torch.ones(3 * d * d - missing.shape[0])
        op.missing_indices = missing
        keep = torch.ones(3 * d * d, dtype=torch.bool)
        keep[missing] = False
        op.kept_indices = torch.nonzero(keep).reshape(-1).long()   # == the list comprehension of :330
        return op
    if name == "cs_walshhadamard":
        return R.WalshHadamardCS(3, d, 4, cases.wh_perm(d), "cpu")
    if name == "denoising":
        return R.Denoising(3, d, "cpu")
    from oracle import operators as O
    if name == "deblur_uni":                                                 # diffusion.py:500-503
        return R.Deblurring(torch.Tensor([1 / 9] * 9), 3, d, "cpu")
    if name == "deblur_gauss":                                               # :504-509
        k = O.gaussian_taps(10, 2)
        return R.Deblurring(k / k.sum(), 3, d, "cpu")
    if name == "deblur_aniso":                                               # :510-521
        k2, k1 = O.gaussian_taps(20, 4), O.gaussian_taps(1, 4)
        return R.Deblurring2D(k1 / k1.sum(), k2 / k2.sum(), 3, d, "cpu")
    raise ValueError(name)


def sub(t, step=8):
    return t[..., ::step, ::step].contiguous().numpy()


def make_classifier():
    """EncoderUNetModel (create_classifier) of the reference + the cond_fn gradient of diffusion.py:183-189
    (torch autograd), fp32, seeded weights: key list, logits and input gradients at three sizes."""
    import torch.nn.functional as F
    from oracle import weights
    ns = ref_import.load()
    out = {}
    for kind, kw in (("small", dict(image_size=32, classifier_depth=1, classifier_attention_resolutions="16,8")),
                     ("mid", dict(image_size=64, classifier_depth=1)), ("full", dict())):
        cc = weights.classifier_config(**kw)
        if cc.image_size == 32:       # create_classifier has no 32-px table; same class, built directly
            ref = ns.unet.EncoderUNetModel(image_size=32, in_channels=3, model_channels=128, out_channels=1000,
                                           num_res_blocks=1, attention_resolutions=(2, 4), channel_mult=(1, 2),
                                           use_fp16=False, num_head_channels=64, use_scale_shift_norm=True,
                                           resblock_updown=True, pool="attention")
        else:
            ref = ns.script_util.create_classifier(**{k: getattr(cc, k) for k in ns.script_util.classifier_defaults()})
        if kind == "full":
            json.dump([[k, list(v.shape)] for k, v in ref.state_dict().items()],
                      open(os.path.join(HERE, "classifier_state_dict_keys.json"), "w"))
        ref.load_state_dict(weights.classifier_state_dict(cc))
        ref.eval()
        g = torch.Generator().manual_seed(cases.SEED + 11)
        r = cc.image_size
        x = torch.randn(2, 3, r, r, generator=g)
        t = torch.tensor([430.0, 10.0])
        y = torch.tensor([951, 17])
        with torch.enable_grad():
            xin = x.detach().requires_grad_(True)
            logits = ref(xin, t)
            sel = F.log_softmax(logits, dim=-1)[range(2), y]
            grad = torch.autograd.grad(sel.sum(), xin)[0]
        out[f"{kind}_logits"] = logits.detach().numpy()
        out[f"{kind}_grad"] = grad.numpy() if kind != "full" else sub(grad, 4)
        out[f"{kind}_grad_norm"] = np.array([grad.double().norm().item()])
    np.savez_compressed(os.path.join(HERE, "classifier.npz"), **out)


def make_deblur():
    """Deblurring / Deblurring2D of the reference: A, A_pinv at 64^2 and one 12-step sampler run (small net)."""
    ns = ref_import.load()
    R = ns.svd_operators
    out = {}
    x = cases.operator_input(64, 2)
    for name in ("deblur_uni", "deblur_gauss", "deblur_aniso"):
        op = ref_operator(R, name, 64)
        y = op.A(x)
        out[f"{name}_y"], out[f"{name}_pinv"] = y.numpy(), op.A_pinv(y.clone()).numpy()
    cfg, sd = cases.celeba_net("small")
    ref = ns.models.Model(cfg)
    ref.load_state_dict(sd)
    ref.eval()
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 12, 1, 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, 12)
    op = ref_operator(R, "deblur_gauss", cfg.data.image_size)
    y = op.A(x_orig)
    with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
        xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, y, cls_fn=None, classes=None,
                                             config=cfg)
    out["sampler_gauss_x"], out["sampler_gauss_x0"] = xs[0].numpy(), x0s[0].numpy()
    np.savez_compressed(os.path.join(HERE, "deblur.npz"), **out)


def make_cs():
    """CS (block-based compressed sensing) of the reference at 64^2, ratio 0.25: A, A_pinv and one 12-step sampler
    run on the small celeba net.  The constructor draws its Gaussian matrix from the global RNG (:107)."""
    from oracle import operators as O
    ns = ref_import.load()
    R = ns.svd_operators
    torch.manual_seed(cases.SEED + 21)
    op = R.CS(3, 64, 0.25, "cpu")
    _, _, V = torch.svd(O.gauss_matrix(cases.SEED + 21), some=False)
    assert torch.equal(V, op.V_small), "oracle.gauss_matrix must reproduce the reference's draw"
    x = cases.operator_input(64, 2)
    y = op.A(x)
    out = {"y": y.numpy(), "pinv": op.A_pinv(y.clone()).numpy(), "singulars_sum": np.array([op.singulars().sum().item()])}
    cfg, sd = cases.celeba_net("small")
    ref = ns.models.Model(cfg)
    ref.load_state_dict(sd)
    ref.eval()
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 12, 1, 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 2, 12)
    torch.manual_seed(cases.SEED + 21)
    op = R.CS(3, cfg.data.image_size, 0.25, "cpu")
    yy = op.A(x_orig)
    with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
        xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, yy, cls_fn=None, classes=None,
                                             config=cfg)
    out["sampler_x"], out["sampler_x0"] = xs[0].numpy(), x0s[0].numpy()
    np.savez_compressed(os.path.join(HERE, "cs_blockbased.npz"), **out)


def make_simplified():
    """Drive the REAL `Diffusion.simplified_ddnm_plus` (guided_diffusion/diffusion.py:211-415) -- the loop is inlined
    in a method full of dataset / image I/O, so the method runs on a bare instance with: a one-image fake dataset,
    `tvu.save_image` capturing tensors, `data_transform` = 2x-1 and an identity `inverse_data_transform` (so the
    captured result is unclamped), the 'cuda'->CPU shim and the noise tape (x_T is the first `torch.randn`)."""
    import importlib
    import types
    ref_import.load()
    sys.path.insert(0, ref_import.REF_ROOT)
    try:
        ds = types.ModuleType("datasets")
        ds.data_transform = lambda config, X: 2 * X - 1.0                 # datasets/__init__.py:201-215, rescaled
        ds.inverse_data_transform = lambda config, X: X                   # identity: keep the result unclamped
        ds.get_dataset = None
        sys.modules["datasets"] = ds
        ck = types.ModuleType("functions.ckpt_util")
        ck.get_ckpt_path = ck.download = None
        sys.modules.setdefault("functions.ckpt_util", ck)
        D = importlib.import_module("guided_diffusion.diffusion")
    finally:
        sys.path.remove(ref_import.REF_ROOT)
        sys.modules.pop("datasets", None)
    ns = ref_import.load()
    out = {}
    saved = {}
    D.tvu.save_image = lambda t, path, **kw: saved.__setitem__(os.path.basename(path), t.detach().clone())
    cwd = os.getcwd()
    os.chdir(ref_import.REF_ROOT)                     # exp/inp_masks/mask.npy is opened relative to the cwd (:257)
    try:
        for case in cases.SIMPLIFIED_CASES:
            name, deg, res, sigma_y = case["name"], case["deg"], case["res"], case["sigma_y"]
            cfg, sd = cases.simplified_net(res)
            tt = cfg.time_travel
            tt.T_sampling, tt.travel_length, tt.travel_repeat = case["T"], case["travel"][0], case["travel"][1]
            n_it = len(schedule.jump_times(tt.T_sampling, tt.travel_length, tt.travel_repeat)) - 1
            x_orig, x_T, tape = cases.sampler_case(cfg, 1, n_it)
            x01 = (x_orig + 1) / 2                                        # the loader yields [0,1] images

            class OneImage(torch.utils.data.Dataset):
                def __len__(self):
                    return 1

                def __getitem__(self, i):
                    return x01[0], 0
            D.get_dataset = lambda args, config: (None, OneImage())
            model = ns.models.Model(cfg)
            model.load_state_dict(sd)
            model.eval()
            runner = object.__new__(D.Diffusion)
            runner.args = argparse.Namespace(deg=deg, deg_scale=4.0, sigma_y=sigma_y / 2, subset_start=-1, subset_end=-1,
                                             seed=1234, image_folder=os.path.join("/tmp", "ddnm_simplified_golden", name),
                                             eta=0.85)
            runner.config, runner.device, runner.betas = cfg, torch.device("cpu"), cases.betas()
            saved.clear()
            orig_randn = torch.randn
            first = {"done": False}

            def randn(*a, **k):                       # x_T (:323-329) is the only torch.randn of the method
                if not first["done"] and tuple(a[:4]) == tuple(x_T.shape):
                    first["done"] = True
                    return x_T.clone()
                k.pop("device", None)
                return orig_randn(*a, **k)
            torch.randn = randn
            try:
                with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
                    runner.simplified_ddnm_plus(model, None)
            finally:
                torch.randn = orig_randn
            final = [v for k, v in saved.items() if k == "-1_0.png"]       # the stale loop variable j = -1 (:402)
            assert len(final) == 1 and first["done"], (list(saved), first)
            full = final[0][None]
            out[f"{name}_x"] = (full if res == 32 else full[..., ::4, ::4]).contiguous().numpy()
            out[f"{name}_stats"] = np.array([full.double().mean().item(), full.double().std().item(),
                                             full.double().abs().sum().item()])
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "simplified.npz"), **out)


def make_plus():
    """ddnm_plus_diffusion (functions/svd_ddnm.py:80-164) of the reference: small celeba net, sigma_y = 0.2
    (doubled value, as the runner passes it), 20 steps with time travel, every operator that has Lambda."""
    ns = ref_import.load()
    R = ns.svd_operators
    cfg, sd = cases.celeba_net("small")
    ref = ns.models.Model(cfg)
    ref.load_state_dict(sd)
    ref.eval()
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    d = cfg.data.image_size
    out = {}
    for name in ("sr_averagepooling", "colorization", "inpainting", "cs_walshhadamard", "denoising"):
        x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
        op = ref_operator(R, name, d)
        y = op.A(x_orig)
        gy = torch.Generator().manual_seed(cases.SEED + 9)
        y = y + 0.2 * torch.randn(y.shape, generator=gy)           # noisy measurement (diffusion.py:549-550)
        with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
            xs, x0s = ns.svd_ddnm.ddnm_plus_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, y, 0.2, cls_fn=None,
                                                      classes=None, config=cfg)
        out[f"{name}_x"], out[f"{name}_x0"], out[f"{name}_y"] = xs[0].numpy(), x0s[0].numpy(), y.numpy()
    np.savez_compressed(os.path.join(HERE, "ddnm_plus_small.npz"), **out)
    make_plus_deblur(ns, R, ref, cfg, n_it)


def make_plus_deblur(ns=None, R=None, ref=None, cfg=None, n_it=None):
    """Same DDNM+ case for Deblurring (the only blur operator with Lambda / Lambda_noise, :1016-1091), own file."""
    if ns is None:
        ns = ref_import.load()
        R = ns.svd_operators
        cfg, sd = cases.celeba_net("small")
        ref = ns.models.Model(cfg)
        ref.load_state_dict(sd)
        ref.eval()
        cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
        n_it = len(schedule.jump_times(20, 2, 2)) - 1
    d = cfg.data.image_size
    out = {}
    for name in ("deblur_uni", "deblur_gauss"):
        x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
        op = ref_operator(R, name, d)
        y = op.A(x_orig)
        gy = torch.Generator().manual_seed(cases.SEED + 9)
        y = y + 0.2 * torch.randn(y.shape, generator=gy)
        with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
            xs, x0s = ns.svd_ddnm.ddnm_plus_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, y, 0.2, cls_fn=None,
                                                      classes=None, config=cfg)
        out[f"{name}_x"], out[f"{name}_x0"], out[f"{name}_y"] = xs[0].numpy(), x0s[0].numpy(), y.numpy()
        # operator-level vectors in two regimes of the threshold
        g = torch.Generator().manual_seed(3)
        v, e = torch.randn(2, 3 * d * d, generator=g), torch.randn(2, 3 * d * d, generator=g)
        for tag, tn, sy in (("hi", 990, 0.4), ("mid", 500, 0.4), ("lo", 10, 0.4)):
            atn = schedule.alpha_bar(cases.betas(), tn)
            a, st = atn.sqrt(), (1 - atn).sqrt()
            out[f"{name}_lambda_{tag}"] = op.Lambda(v.clone(), a, sy, st, 0.85).numpy()
            out[f"{name}_lambda_noise_{tag}"] = op.Lambda_noise(v.clone(), a, sy, st, 0.85, e.clone()).numpy()
    np.savez_compressed(os.path.join(HERE, "ddnm_plus_deblur.npz"), **out)


def make_adm():
    """ADM UNetModel (guided_diffusion/unet.py via script_util.create_model), fp32, seeded weights with the
    zero_module'd tensors re-randomised: key list, forwards at three sizes, one sampler run."""
    ns = ref_import.load()
    R = ns.svd_operators
    torch.set_num_threads(os.cpu_count())
    out = {}
    for kind, batch in (("small", 2), ("mid", 2), ("full", 1)):
        cfg, sd = cases.adm_net(kind)
        ref = ns.script_util.create_model(**vars(cfg.model))
        if kind == "full":
            keys = [[k, list(v.shape)] for k, v in ref.state_dict().items()]
            json.dump(keys, open(os.path.join(HERE, "adm_state_dict_keys.json"), "w"))
        ref.load_state_dict(sd)
        ref.eval()
        x, t, y = cases.adm_forward_inputs(cfg, batch)
        with torch.no_grad():
            e = ref(x, t, y) if y is not None else ref(x, t)
        out[f"{kind}_eps"] = e.numpy() if kind != "full" else sub(e, 4)
        out[f"{kind}_stats"] = np.array([e.double().mean().item(), e.double().std().item(), e.double().abs().sum().item()])
        if kind == "mid":       # sampler: colorization + inpainting with time travel through the reference loop
            cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
            n_it = len(schedule.jump_times(20, 2, 2)) - 1
            d = cfg.data.image_size
            for name in ("colorization", "inpainting"):
                x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
                op = ref_operator(R, name, d)
                yy = op.A(x_orig)
                with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
                    xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, yy, cls_fn=None,
                                                         classes=None, config=cfg)
                out[f"mid_{name}_x"] = xs[0].numpy()
                out[f"mid_{name}_x0"] = x0s[0].numpy()
    np.savez_compressed(os.path.join(HERE, "adm_forward.npz"), **out)


def make_spectral():
    """The matrix-free SVD surface of the five BASELINE operators at 64 x 64 through the REAL reference classes
    (functions/svd_operators.py:9-97 and the per-class V / Vt / U / Ut / add_zeros): tests/golden/spectral.npz.
    V / Vt of the permutation-type operators are unique; for the operators built on a LAPACK SVD only the
    basis-independent products (At, A_pinv_eta, A) are compared tightly by the tests."""
    ns = ref_import.load()
    R = ns.svd_operators
    out = {}
    d = 64
    x = cases.operator_input(d, 2)
    g = torch.Generator().manual_seed(cases.SEED + 31)
    SP = lambda t: t[:, ::5].contiguous()        # noqa: E731  (every 5th entry of the big vectors)
    for name in ("sr_averagepooling", "sr_bicubic", "colorization", "inpainting", "cs_walshhadamard"):
        op = ref_operator(R, name, d)
        y = op.A(x)
        z = torch.randn(2, 3 * d * d, generator=g)
        w = torch.randn(*y.shape, generator=g)
        out[f"{name}_Vt"] = SP(op.Vt(x.clone())).numpy()
        out[f"{name}_V"] = SP(op.V(z.clone())).numpy()
        out[f"{name}_Ut"] = op.Ut(w.clone()).numpy()
        out[f"{name}_U"] = op.U(w.clone()).numpy()
        out[f"{name}_add_zeros"] = op.add_zeros(w.clone())[:, ::7].numpy()
        out[f"{name}_At"] = SP(op.At(w.clone())).numpy()
        out[f"{name}_A_pinv_eta"] = SP(op.A_pinv_eta(w.clone(), 0.3)).numpy()
        out[f"{name}_z"], out[f"{name}_w"] = z.numpy(), w.numpy()
    np.savez_compressed(os.path.join(HERE, "spectral.npz"), **{k: v.astype(np.float32) for k, v in out.items()})


def make_spectral_sr16():
    """The same surface for SuperResolution at ratio 16 (`--deg sr_averagepooling --deg_scale 16`, evaluation.sh:18: sites of
    n = 256 entries -- the engine's V / Vt take the GEMM route there), 64 x 64 image, plus Lambda / Lambda_noise at the
    noisy setting of that line (sigma_y = 0.4 after the runner's doubling): tests/golden/spectral_sr16.npz."""
    ns = ref_import.load()
    R = ns.svd_operators
    d = 64
    x = cases.operator_input(d, 2)
    g = torch.Generator().manual_seed(cases.SEED + 32)
    op = R.SuperResolution(3, d, 16, "cpu")
    y = op.A(x)
    z = torch.randn(2, 3 * d * d, generator=g)
    w = torch.randn(*y.shape, generator=g)
    e = torch.randn(2, 3 * d * d, generator=g)
    out = dict(z=z.numpy(), w=w.numpy(), e=e.numpy(), A=y.numpy(), A_pinv=op.A_pinv(w.clone()).numpy(),
               Vt=op.Vt(x.clone()).numpy(), V=op.V(z.clone()).numpy(), Ut=op.Ut(w.clone()).numpy(), U=op.U(w.clone()).numpy(),
               add_zeros=op.add_zeros(w.clone()).numpy(), At=op.At(w.clone()).numpy(),
               A_pinv_eta=op.A_pinv_eta(w.clone(), 0.3).numpy())
    for tag, (a, st) in {"early": (0.2, 0.97), "late": (0.98, 0.15)}.items():     # both branches of svd_ddnm.py:121-131
        out[f"Lambda_{tag}"] = op.Lambda(z.clone(), torch.tensor(a), 0.4, torch.tensor(st), 0.85).numpy()
        out[f"Lambda_noise_{tag}"] = op.Lambda_noise(z.clone(), torch.tensor(a), 0.4, torch.tensor(st), 0.85, e.clone()).numpy()
    np.savez_compressed(os.path.join(HERE, "spectral_sr16.npz"), **{k: v.astype(np.float32) for k, v in out.items()})


def make_general_a():
    """GeneralA (functions/svd_operators.py:173-208) of the reference on a seeded dense 64 x 192 matrix (3 x 8 x 8 image, rank
    deficient on purpose: the singular-value threshold 1e-3 of :184-185 must act): tests/golden/general_a.npz holds the
    basis-independent products A, A_pinv, At, A_pinv_eta and the thresholded singular values."""
    ns = ref_import.load()
    g = torch.Generator().manual_seed(cases.SEED + 41)
    A = torch.randn(64, 192, generator=g) / 192 ** 0.5
    A[40:] = A[:24] * 0.5 + 1e-5 * torch.randn(24, 192, generator=g)          # 24 nearly dependent rows -> tiny singular values
    op = ns.svd_operators.GeneralA(A.clone())
    x = torch.randn(4, 3, 8, 8, generator=g)
    w = torch.randn(4, 64, generator=g)
    out = dict(A_mat=A.numpy(), x=x.numpy(), w=w.numpy(), singulars=op.singulars().numpy(), A=op.A(x.clone()).numpy(),
               A_pinv=op.A_pinv(w.clone()).numpy(), At=op.At(w.clone()).numpy(),
               A_pinv_eta=op.A_pinv_eta(w.clone(), 0.3).numpy())
    np.savez_compressed(os.path.join(HERE, "general_a.npz"), **{k: v.astype(np.float32) for k, v in out.items()})


def make_full(names):
    """--full-adm / --full-c2b8: the FULL BASELINE configurations through the real reference loop
    (functions/svd_ddnm.py:19-78), operators built as guided_diffusion/diffusion.py:451-523 builds them, fp32
    models from the reference constructors (`Model`, `create_model`, `create_classifier`), `cond_fn` restated from
    diffusion.py:183-189.  One file per case: tests/golden/full_<case>.npz with the stride-4 sub-sample of x_0 and of
    the last x0|t, per-image PSNR vs x_orig, full-tensor statistics and three intermediate x0|t (captured as the
    argument of the loop's `torch.randn_like(x0_t)`).  CPU cost on 8 threads: c2b8 ~17 min, c3 ~8 min, c4 ~25 min,
    c5 ~13 min."""
    import time
    import torch.nn.functional as F
    from oracle import sampler, weights
    ns = ref_import.load()
    R = ns.svd_operators
    torch.set_num_threads(os.cpu_count())
    mask_real = torch.from_numpy(np.load(os.path.join(ref_import.REF_ROOT, "exp/inp_masks/mask.npy")))
    for name in names:
        c, cfg, sd, x_orig, x_T, tape = cases.full_case(name)
        if c["net"] == "celeba":
            ref = ns.models.Model(cfg)
        else:
            ref = ns.script_util.create_model(**vars(cfg.model))
        ref.load_state_dict(sd)
        ref.eval()
        cls_fn = None
        if c["class_cond"]:
            cc = weights.classifier_config()
            clf = ns.script_util.create_classifier(**{k: getattr(cc, k) for k in ns.script_util.classifier_defaults()})
            clf.load_state_dict(weights.classifier_state_dict(cc))
            clf.eval()

            def cls_fn(x, t, y, clf=clf, scale=cc.classifier_scale):          # diffusion.py:183-189
                with torch.enable_grad():
                    x_in = x.detach().requires_grad_(True)
                    logits = clf(x_in, t)
                    log_probs = F.log_softmax(logits, dim=-1)
                    selected = log_probs[range(len(logits)), y.view(-1)]
                    return torch.autograd.grad(selected.sum(), x_in)[0] * scale
        op = ref_operator(R, c["deg"], 256, mask_real if c["deg"] == "inpainting" else None, ratio=c.get("ratio", 4))
        y = op.A(x_orig)
        sigma_y = c.get("sigma_y", 0.0)
        if sigma_y > 0:                                  # --add_noise (diffusion.py:549-551), seeded
            y = y + sigma_y * torch.randn(y.shape, generator=torch.Generator().manual_seed(cases.SEED + 9))
        times = schedule.jump_times(c["T"], *c["travel"])
        for k in c["record"]:
            assert times[k + 1] < times[k], f"record index {k} of {name} is not a reverse step"
        inter = {}

        def on_call(k, x0_t, inter=inter, want=c["record"]):
            if k in want:
                inter[k] = x0_t.detach().clone()
        t0 = time.perf_counter()
        with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape, on_call=on_call):
            if sigma_y > 0:                              # diffusion.py:587-590
                xs, x0s = ns.svd_ddnm.ddnm_plus_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, y, sigma_y,
                                                          cls_fn=cls_fn, classes=None, config=cfg)
            else:
                xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, cases.betas(), 0.85, op, y, cls_fn=cls_fn,
                                                     classes=None, config=cfg)
        dt = time.perf_counter() - t0
        x, x0 = xs[0], x0s[0]
        out = dict(x_sub=sub(x, 4), x0_sub=sub(x0, 4), psnr=sampler.psnr(x, x_orig).numpy(),
                   stats=np.array([x.double().mean().item(), x.double().std().item(), x.double().abs().sum().item()]),
                   consistency=np.array([(op.A(x) - y).abs().max().item()]),
                   record_k=np.array(c["record"]), ref_cpu_seconds=np.array([dt]), y=(y.numpy() if sigma_y > 0 else np.zeros(0)),
                   ref_cpu_threads=np.array([torch.get_num_threads()]))
        for k, v in inter.items():
            out[f"x0_k{k}_sub"] = sub(v, 4)
            out[f"x0_k{k}_stats"] = np.array([v.double().mean().item(), v.double().std().item(),
                                              v.double().abs().sum().item()])
        np.savez_compressed(os.path.join(HERE, f"full_{name}.npz"), **out)
        print(f"full {name}: {dt:.1f} s on {torch.get_num_threads()} threads, psnr {out['psnr']}, "
              f"consistency {out['consistency']}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--full-cases", default="", help="comma list of c2b8,c3,c4,c5: run the FULL BASELINE "
                    "configurations through the reference (tests/golden/full_<case>.npz; up to 25 min of CPU each)")
    ap.add_argument("--adm-only", action="store_true", help="only (re)generate the ADM UNet goldens")
    ap.add_argument("--plus-only", action="store_true", help="only (re)generate the DDNM+ goldens")
    ap.add_argument("--plus-deblur-only", action="store_true", help="only (re)generate the DDNM+ deblurring goldens")
    ap.add_argument("--simplified-only", action="store_true", help="only (re)generate the simplified-loop goldens")
    ap.add_argument("--cs-only", action="store_true", help="only (re)generate the block-based CS goldens")
    ap.add_argument("--deblur-only", action="store_true", help="only (re)generate the deblurring goldens")
    ap.add_argument("--classifier-only", action="store_true", help="only (re)generate the classifier goldens")
    ap.add_argument("--spectral-only", action="store_true", help="only (re)generate the V / Vt / U / Ut / At goldens")
    ap.add_argument("--spectral-sr16-only", action="store_true", help="only (re)generate the ratio-16 SuperResolution goldens")
    ap.add_argument("--general-a-only", action="store_true", help="only (re)generate the GeneralA goldens")
    args = ap.parse_args()
    if args.full_cases:
        return make_full(args.full_cases.split(","))
    if args.spectral_only:
        return make_spectral()
    if args.spectral_sr16_only:
        return make_spectral_sr16()
    if args.general_a_only:
        return make_general_a()
    if args.classifier_only:
        return make_classifier()
    if args.plus_deblur_only:
        return make_plus_deblur()
    if args.simplified_only:
        return make_simplified()
    if args.cs_only:
        return make_cs()
    if args.deblur_only:
        return make_deblur()
    if args.adm_only:
        return make_adm()
    if args.plus_only:
        return make_plus()
    ns = ref_import.load()
    R = ns.svd_operators
    torch.set_num_threads(os.cpu_count())

    # 1. state-dict key/shape list of the real constructor (full celeba config)
    cfg_full, sd_full = cases.celeba_net("full")
    ref_full = ns.models.Model(cfg_full)
    keys = [[k, list(v.shape)] for k, v in ref_full.state_dict().items()]
    json.dump(keys, open(os.path.join(HERE, "celeba_state_dict_keys.json"), "w"))

    # 2. schedules
    sched = {f"jump_{T}_{l}_{r}": np.array(ns.svd_ddnm.get_schedule_jump(T, l, r))
             for (T, l, r) in [(100, 1, 1), (100, 10, 3), (100, 2, 2), (20, 2, 2)]}
    b = cases.betas()
    ab = ns.svd_ddnm.compute_alpha(b, torch.arange(-1, 1000)).reshape(-1)
    sched["alpha_bar_m1_to_999"] = ab.numpy()
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **sched)

    # 3. operators
    out = {}
    mask_real = torch.from_numpy(np.load(os.path.join(ref_import.REF_ROOT, "exp/inp_masks/mask.npy")))
    np.savez_compressed(os.path.join(HERE, "inp_mask.npz"), packed=np.packbits(mask_real.numpy().astype(np.uint8)),
                        shape=np.array(mask_real.shape))
    for d in (64, 256):
        x = cases.operator_input(d, 2)
        for name in OPS:
            mask = mask_real if (name == "inpainting" and d == 256) else None
            op = ref_operator(R, name, d, mask)
            y = op.A(x)
            p = op.A_pinv(y.clone())
            out[f"{name}_{d}_y"] = y.numpy() if d == 64 else y[:, ::31].contiguous().numpy()
            out[f"{name}_{d}_pinv"] = (p.reshape(2, 3, d, d).numpy() if d == 64
                                       else sub(p.reshape(2, 3, d, d), 8))
            out[f"{name}_{d}_ysum"] = np.array([y.double().sum().item(), y.double().abs().sum().item()])
    # bicubic kernel as built by the reference runner (diffusion.py:485-499), re-executed verbatim
    # through the oracle's restatement and compared against the SRConv it feeds
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **out)

    # 4. UNet forwards
    fw = {}
    for kind, batch in (("small", 2), ("mid", 2), ("full", 1)):
        cfg, sd = cases.celeba_net(kind)
        ref = ns.models.Model(cfg)
        ref.load_state_dict(sd)
        ref.eval()
        x, t = cases.forward_inputs(cfg, batch)
        with torch.no_grad():
            e = ref(x, t)
        fw[f"{kind}_eps"] = e.numpy() if kind != "full" else sub(e, 4)
        fw[f"{kind}_stats"] = np.array([e.double().mean().item(), e.double().std().item(),
                                        e.double().abs().sum().item()])
    np.savez_compressed(os.path.join(HERE, "celeba_forward.npz"), **fw)

    # 5. sampler, small net, every operator, with time travel
    sm = {}
    cfg, sd = cases.celeba_net("small")
    ref = ns.models.Model(cfg)
    ref.load_state_dict(sd)
    ref.eval()
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 20, 2, 2
    n_it = len(schedule.jump_times(20, 2, 2)) - 1
    d = cfg.data.image_size
    for name in OPS:
        x_orig, x_T, tape = cases.sampler_case(cfg, 2, n_it)
        op = ref_operator(R, name, d)
        y = op.A(x_orig)
        with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
            xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, b, 0.85, op, y, cls_fn=None,
                                                 classes=None, config=cfg)
        sm[f"{name}_x"] = xs[0].numpy()
        sm[f"{name}_x0"] = x0s[0].numpy()
    np.savez_compressed(os.path.join(HERE, "ddnm_small.npz"), **sm)

    # 6. BASELINE config 2 at B=1: celeba_hq Model, sr_bicubic 4x, T=100 (slow)
    if args.full:
        import time
        cfg, sd = cases.celeba_net("full")
        ref = ns.models.Model(cfg)
        ref.load_state_dict(sd)
        ref.eval()
        x_orig, x_T, tape = cases.sampler_case(cfg, 1, 100)
        op = ref_operator(R, "sr_bicubic", 256)
        y = op.A(x_orig)
        t0 = time.perf_counter()
        with ref_import.cuda_is_cpu(), ref_import.noise_tape(tape):
            xs, x0s = ns.svd_ddnm.ddnm_diffusion(x_T.clone(), ref, b, 0.85, op, y, cls_fn=None,
                                                 classes=None, config=cfg)
        dt = time.perf_counter() - t0
        x = xs[0]
        from oracle import sampler
        np.savez_compressed(os.path.join(HERE, "ddnm_full_c2.npz"), x_sub=sub(x, 4), x0_sub=sub(x0s[0], 4),
                            psnr=sampler.psnr(x, x_orig).numpy(),
                            stats=np.array([x.double().mean().item(), x.double().std().item()]),
                            ref_cpu_seconds=np.array([dt]), ref_cpu_threads=np.array([torch.get_num_threads()]))
        print("full c2: %.1f s, psnr %s" % (dt, sampler.psnr(x, x_orig)))


if __name__ == "__main__":
    main()
