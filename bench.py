#!/usr/bin/env python
"""Headline benchmark: restored images/sec @256x256, 100 DDIM steps, 4x SR (BASELINE.json).

Workload at N=1 = BASELINE config 2: celeba_hq `Model` (fp32, 113.67 M params, 498.35 GFLOP per
forward per image), `sr_bicubic` 4x (SRConv), sigma_y = 0, eta = 0.85, T_sampling = 100,
batch_size = 8 per GPU.  One bench "step" = one full pass of the hot path over one batch:
x_T on device -> 100 reverse steps (UNet forward + projection + DDIM update) -> x_0 on device.
Synthetic inputs, seeded random weights of the real architecture (no checkpoints offline).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N>1: one process per GPU, every rank restores its own batch of 8 (weak scaling), one RCCL
all_gather of the restored images at the end of each step (inside the timed region).
Prints ONE JSON line on rank 0.  Started WITHOUT torchrun, `--gpus N > 1` re-executes itself under
`torch.distributed.run --nproc-per-node N` (exit code 2 when fewer than N devices are visible); the line
carries `backend`, `ranks_seen` and the per-rank step-time spread, and at N > 1 it appends configs[2] /
configs[3] with their fixed GLOBAL batch split over the ranks (`--no-extra-workloads` skips them).

At 1 GPU the default run appends `"workloads": {"c3", "c4", "c5"}`: the per-GPU shards of BASELINE configs[2..4]
(ImageNet ADM UNet in the runner's fp16 mode), 1 warm-up + 2 timed restorations each, each with its own
`roofline` against the 2.5 PFLOP/s fp16 MFMA peak (`--no-extra-workloads` skips them; `--workload c3 --scaling
strong` times configs[2] / configs[3] with their fixed GLOBAL batch of 32 / 16 split over the ranks).
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.perf_counter()      # before `import torch`: start-up is part of what an N > 1 line reports (`startup_s`)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= vector) peak
# The 3x3 layers of the headline workload run fp32-GRADE products on the fp16 matrix pipe: one product = three
# v_mfma_f32_32x32x16_f16 products (hi*hi + hi*lo + lo*hi, ddnm_conv3x3_s16_f32).  `achieved` counts ALGORITHMIC FLOPs
# (one multiply-add per product), so the bound of that kernel is a third of the dense fp16 MFMA peak.
PEAK_SPLIT16_TFLOPS = round(2500.0 / 3.0, 1)
FLOPS_PER_FWD_PER_IMAGE = 498.35e9   # SURVEY.md section 8(d), celeba Model
T_SAMPLING = 100
BATCH_PER_GPU = 8


def precision_check(dev):
    """One seeded layer of the headline UNet (128 -> 128 @128^2, B = 2, GroupNorm affine + swish prologue, bias, residual)
    evaluated in fp64 by torch on the GPU and by the two convolution engines through the C ABI: relative L2 error of the
    split-fp16 kernel (ddnm_conv3x3_s16_f32) and of the fp32 MFMA kernel (ddnm_conv2d_f32) against fp64."""
    import torch.nn.functional as F
    from ddnm_amd import ops
    g = torch.Generator(device=dev).manual_seed(2024)
    rn = lambda *sh: torch.randn(*sh, device=dev, generator=g)  # noqa: E731
    B, C, H = 2, 128, 128
    a, w, bias = rn(B, H, H, C) * 1.5, rn(C, C, 3, 3) / (3.0 * C ** 0.5), rn(C)
    sc, sh, r = rn(B, C) * 0.3 + 1.0, rn(B, C) * 0.3, rn(B, H, H, C) * 2.0
    x = a.double() * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
    x = (x * torch.sigmoid(x)).permute(0, 3, 1, 2)
    y = F.conv2d(x, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1) + r.double()
    w32, scale = ops.pack_conv_weight(w), ops.s16_weight_scale(w)
    s16 = (ops.pack_conv_weight_s16(w, scale), scale, None)
    out = {}
    for name, ws in (("split16", s16), ("f32_mfma", None)):
        o = ops.conv2d(a, w32, C, 3, bias=bias, res=r, gn=(sc, sh), gn_silu=True, weight_s16=ws)
        out[name] = float((o.double() - y).norm() / y.norm())
    return {"layer": "3x3 128->128 @128x128, B=2, GroupNorm affine + swish, bias, residual (seeded)",
            "rel_l2_error_vs_fp64": out,
            "note": "fp64 = torch on the same GPU; both engines through the C ABI; tests/test_gpu_s16.py holds 16 more layer forms"}


def baseline_metric():
    """The headline metric's name exactly as BASELINE.json spells it."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:      # noqa: BLE001
        return "restored images/sec @256x256, 100 DDIM steps, 4x SR"


def make_config():
    import types
    ns = types.SimpleNamespace
    return ns(
        model=ns(type="simple", in_channels=3, out_ch=3, ch=128, ch_mult=[1, 1, 2, 2, 4, 4], num_res_blocks=2,
                 attn_resolutions=[16], dropout=0.0, var_type="fixedsmall", ema_rate=0.999, ema=True,
                 resamp_with_conv=True),
        data=ns(dataset="CelebA_HQ", image_size=256, channels=3, rescaled=True, logit_transform=False,
                uniform_dequantization=False, gaussian_dequantization=False),
        diffusion=ns(beta_schedule="linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000),
        time_travel=ns(T_sampling=T_SAMPLING, travel_length=1, travel_repeat=1),
        sampling=ns(batch_size=BATCH_PER_GPU))


def cpu_baseline(cfg, sd, budget_s=36.0):
    """Reported baseline (not the optimisation target): the reference path on this box's host cores -- the reference's
    own `Model` when /root/reference is importable (`kind: "reference"`, build container only), otherwise the oracle
    restatement (`kind: "port"`; bit-identical to the reference UNet on CPU, tests/test_oracle_pins.py).
    Bounded sample: B=1, as many reverse steps of the 100 as fit in `budget_s` (at least 30, all 100 when they fit),
    extrapolated; then the workload's own batch (B=8) for 3 reverse steps (`b8_value`, the like-for-like figure).  The
    intra-op thread count is calibrated first on a reduced UNet (oversubscribing a many-core host makes the ATen CPU
    kernels dramatically slower)."""
    from oracle import cases, ref_import, sampler, unet_celeba
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    mcfg, msd = cases.celeba_net("mid")
    mnet = unet_celeba.Net(msd, mcfg)

    def calibrate(batch):
        """Best intra-op thread count for forwards of this batch size (reduced UNet; ATen's CPU convolutions parallelise
        differently over batch 1 and batch 8, so each timed batch gets its own calibration)."""
        mx, mt = cases.forward_inputs(mcfg, batch)
        best = (None, 1e30)
        for nt in (8, 16, 32, 64, 128):
            if nt > avail:
                break
            torch.set_num_threads(nt)
            mnet(mx, mt)
            t0 = time.perf_counter()
            for _ in range(2):
                mnet(mx, mt)
            dt = (time.perf_counter() - t0) / 2
            if dt < best[1]:
                best = (nt, dt)
        return best[0] or min(avail, 8)
    threads = calibrate(1)
    torch.set_num_threads(threads)
    kind = "port"
    net = unet_celeba.Net(sd, cfg)
    if ref_import.available():
        try:
            ref = ref_import.load().models.Model(cfg)
            ref.load_state_dict(sd)
            ref.eval()
            net, kind = (lambda x, t, ref=ref: ref(x, t)), "reference"
        except Exception:      # noqa: BLE001
            kind = "port"
    op = cases.make_operator("sr_bicubic", 256)

    class Stop(Exception):
        pass

    def timed(batch, n_steps):
        x_orig, x_T, tape = cases.sampler_case(cfg, batch, n_steps)
        y = op.A(x_orig)

        def record(k, name, t):
            if name == "xt_next" and k == n_steps - 1:
                raise Stop
        t0 = time.perf_counter()
        try:
            sampler.ddnm_diffusion(x_T, net, cases.betas(), 0.85, op, y, tape, record=record)
        except Stop:
            pass
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    net(torch.zeros(1, 3, 256, 256), torch.tensor([990.0]))          # warm-up forward, also sizes the sample
    t_fwd = time.perf_counter() - t0
    # the SAME bounded sample at three thread counts around the calibrated one (the calibration on the reduced UNet has
    # picked 16 threads on boxes where the full-size loop then ran 2x apart: 0.0225 vs 0.045 images/s, round 5); the best
    # of the three is `value`, all three are in the line (`by_threads`)
    cands = sorted({max(1, min(avail, c)) for c in (threads // 2, threads, threads * 2)})
    n_steps = int(max(20, min(T_SAMPLING, (budget_s / len(cands)) // max(t_fwd, 1e-3))))
    by_threads = {}
    for c in cands:
        torch.set_num_threads(c)
        net(torch.zeros(1, 3, 256, 256), torch.tensor([990.0]))      # this thread count's warm-up
        by_threads[c] = timed(1, n_steps)
    threads, dt = min(by_threads.items(), key=lambda kv: kv[1])
    torch.set_num_threads(threads)
    cpu_model = "unknown"
    try:                       # the figure swings 0.038 ... 0.054 images/s between boxes: name the host CPU next to it
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.lower().startswith("model name")), "unknown")
    except OSError:
        pass
    out = {"value": 1.0 / (dt / n_steps * T_SAMPLING), "unit": "images/sec", "cores": threads, "kind": kind,
           "host_cpu": cpu_model, "logical_cpus": avail,
           "by_threads": {str(c): round(1.0 / (t / n_steps * T_SAMPLING), 5) for c, t in sorted(by_threads.items())},
           "sample": f"B=1, {n_steps} of {T_SAMPLING} reverse steps timed at each of {sorted(by_threads)} threads "
                     f"({sum(by_threads.values()):.1f} s in all; {avail} logical CPUs visible), extrapolated "
                     f"x{T_SAMPLING / n_steps:g}; `value` = the best of them ({threads} threads), `by_threads` = all; "
                     "`b8_value`: the workload's own batch of 8 for 3 reverse steps, same extrapolation"}
    try:
        threads8 = calibrate(BATCH_PER_GPU)
        torch.set_num_threads(threads8)
        net(torch.zeros(BATCH_PER_GPU, 3, 256, 256), torch.full((BATCH_PER_GPU,), 990.0))      # warm-up at this batch
        dt8 = timed(BATCH_PER_GPU, 3)
        out["b8_value"] = BATCH_PER_GPU / (dt8 / 3 * T_SAMPLING)
        out["b8_seconds_for_3_steps"] = round(dt8, 2)
        out["b8_cores"] = threads8
    except Exception as e:    # noqa: BLE001
        out["b8_value"] = None
        out["b8_error"] = repr(e)
    return out


# ------------------------------------------------------------------------------------------------- ImageNet workloads
ADM_FLOPS_PER_FWD = 2242.87e9          # SURVEY.md section 8(d), per image
ADM_CC_FLOPS_PER_FWD = 2243.9e9        # class-conditional UNet (+ label embedding)
CLS_FLOPS_ESTIMATE = 300e9             # classifier forward + input gradient per image: SURVEY's estimate, replaced by the
#                                        count of the engine's own launches (convolutions + batched GEMMs) when available
PEAK_F16_TFLOPS = 2500.0               # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
ADM_WORKLOADS = {
    "c3": dict(desc="imagenet_256.yml colorization, T_sampling=100 (BASELINE configs[2]: batch_size=32 sharded across "
                    "8 GPUs = 4 images per GPU)", batch=4, travel=(1, 1), global_batch=32, ranks=8),
    "c4": dict(desc="imagenet_256.yml inpainting, time travel l=10 r=3: 280 NFE + 180 re-noise steps (BASELINE configs[3]: "
                    "batch_size=16 on 4 GPUs = 4 images per GPU)", batch=4, travel=(10, 3), global_batch=16, ranks=4),
    "c5": dict(desc="imagenet_256_cc.yml cs_walshhadamard 0.25, class-conditional ADM + classifier guidance (class 951, "
                    "scale 1.0), batch_size=8 on 1 GPU (BASELINE configs[4])", batch=8, travel=(1, 1), global_batch=8,
               ranks=1),
}


def real_inpainting_mask():
    """The reference's exp/inp_masks/mask.npy (256 x 256, 1 = kept), committed bit-packed as tests/golden/inp_mask.npz."""
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "inp_mask.npz"))
    shape = tuple(g["shape"])
    bits = np.unpackbits(g["packed"])[: shape[0] * shape[1]]
    return torch.from_numpy(bits.reshape(shape).astype(np.int64))


def adm_workload(name, ddist, rank, world, dev, steps, warmup, strong=False, roofline=True, lib_digest=None):
    """One ImageNet workload: ADM UNet (552.81 M parameters, 2242.87 GFLOP per forward per image) in the runner's
    `use_fp16: true` mode = fp16 activations + fp16 MFMA operands, fp32 accumulation (ddnm_amd/guided_diffusion/unet.py).
    weak scaling (default): every rank restores the per-GPU shard of the BASELINE config; `strong`: the config's GLOBAL
    batch is split over the ranks present.  Returns the result dict (rank 0 prints it)."""
    import types
    from ddnm_amd import ops
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion, get_schedule_jump
    from ddnm_amd.functions.svd_operators import Colorization, Inpainting, WalshHadamardCS
    from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier, make_cond_fn
    from ddnm_amd.guided_diffusion.diffusion import get_beta_schedule
    from ddnm_amd.guided_diffusion.unet import create_model
    W = ADM_WORKLOADS[name]
    ns = types.SimpleNamespace
    travel = W["travel"]
    cfg = ns(diffusion=ns(num_diffusion_timesteps=1000), data=ns(image_size=256, channels=3),
             time_travel=ns(T_sampling=T_SAMPLING, travel_length=travel[0], travel_repeat=travel[1]))
    model = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8",
                         num_head_channels=64, learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
                         use_fp16=True, class_cond=(name == "c5"))
    model.device = dev
    model.load_state_dict(model.random_state_dict(1234))
    model.convert_to_fp16()
    cls_fn = None
    cls_flops_per_image = None
    if name == "c5":
        kw = classifier_defaults()
        kw["image_size"] = 256
        clf = create_classifier(**kw)
        clf.device = dev
        gsd = torch.Generator().manual_seed(4321)
        clf.load_state_dict({k: (torch.randn(v, generator=gsd) * (1.0 / max(1, int(torch.tensor(v[1:]).prod()))) ** 0.5
                                 if len(v) > 1 else (1.0 + 0.1 * torch.randn(v, generator=gsd) if k.endswith("weight")
                                                     else 0.05 * torch.randn(v, generator=gsd)))
                             for k, v in clf.state_dict_shapes().items()})
        clf.convert_to_fp16()                  # imagenet_256_cc.yml: classifier_use_fp16 true
        cls_fn = make_cond_fn(clf, 1.0)
        try:        # FLOPs of one guidance evaluation, counted from the launches of one instrumented call
            ctimer = ops.KernelTimer()
            ops.set_kernel_timer(ctimer)
            xx = torch.randn(2, 3, 256, 256, device=dev)
            cls_fn(xx, torch.full((2,), 500.0, device=dev), torch.full((2,), 951, dtype=torch.long, device=dev))
            ops.set_kernel_timer(None)
            torch.cuda.synchronize()
            cls_flops_per_image = sum(f for _, f, _, _ in ctimer.records) / 2.0
        except Exception:      # noqa: BLE001
            ops.set_kernel_timer(None)
    betas = torch.from_numpy(get_beta_schedule("linear", beta_start=1e-4, beta_end=0.02,
                                               num_diffusion_timesteps=1000)).float().to(dev)
    if strong:
        lo, hi = ddist.shard_range(W["global_batch"], rank, world)
        B, n_total = hi - lo, W["global_batch"]
    else:
        B, n_total = W["batch"], W["batch"] * world
    g = torch.Generator().manual_seed(1234 + rank)
    x_orig = (torch.rand(max(B, 1), 3, 256, 256, generator=g) * 2 - 1).to(dev)
    if name == "c3":
        op = Colorization(256, dev)
    elif name == "c5":
        op = WalshHadamardCS(3, 256, 4, torch.randperm(256 * 256, generator=g), dev)
    else:
        mask = real_inpainting_mask().reshape(-1)        # exp/inp_masks/mask.npy of the reference (bit-packed copy)
        r = torch.nonzero(mask == 0).long().reshape(-1) * 3
        op = Inpainting(3, 256, torch.cat([r, r + 1, r + 2], 0), dev)
    y = op.A(x_orig)
    times = get_schedule_jump(T_SAMPLING, *travel)
    nfe = sum(1 for a, b in zip(times[:-1], times[1:]) if b < a)

    passes = [0]

    def one_pass():
        # x_T and the per-iteration noise from the engine's own counter-based generator (in-kernel Philox draw: no ATen
        # op in the timed loop); seed = pass number, counter = GLOBAL image index, so the noise of an image does not
        # depend on the rank count
        passes[0] += 1
        ph = ops.PhiloxNoise(seed=0x5EED0000 + passes[0], image_base=(lo if strong else rank * B))
        x_T = ph.tensor(ops.PhiloxNoise.XT_ITER, x_orig)
        xs, _ = ddnm_diffusion(x_T, model, betas, 0.85, op, y, cls_fn=cls_fn, classes=None, config=cfg, return_cpu=False,
                               noise=ph)
        return ddist.gather_images(xs[0][:B], n_total=n_total)

    for _ in range(warmup):
        out = one_pass()
    ddist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_pass()
    torch.cuda.synchronize()
    ddist.barrier()
    dt_rank = time.perf_counter() - t0
    dt, dt_min = ddist.reduce_scalar(dt_rank, dev, "max"), ddist.reduce_scalar(dt_rank, dev, "min")
    value = steps * n_total / dt
    per_step = ADM_FLOPS_PER_FWD
    if name == "c5":
        per_step = ADM_CC_FLOPS_PER_FWD + (cls_flops_per_image or CLS_FLOPS_ESTIMATE)
    tfl = value * nfe * per_step / 1e12 / world
    res = {"value": round(value, 4), "unit": "images/sec", "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(dt / steps * 1e3, 2), "ms_per_step_rank_min": round(dt_min / steps * 1e3, 2),
           "scaling": "strong" if strong else "weak",
           "dtype": "f16 (activations and MFMA operands; f32 accumulate, GroupNorm statistics, softmax)",
           "data": "synthetic",
           "config": {"workload": W["desc"], "global_batch": n_total, "per_gpu_batch": B, "nfe_per_image": nfe,
                      # BASELINE quotes c3 on 8 GPUs and c4 on 4: only the strong line at THAT rank count reproduces the
                      # configuration; at another N it is the same global batch on other shards (labelled)
                      "baseline_ranks": W["ranks"],
                      "reproduces_baseline_config": bool(world == W["ranks"] and (strong or W["ranks"] == 1)),
                      "variant": (None if world == W["ranks"] or not strong else
                                  f"BASELINE global batch {W['global_batch']} on {world} ranks instead of {W['ranks']}")},
           "whole_loop_tflops_per_gpu": round(tfl, 1), "whole_loop_frac": round(tfl / PEAK_F16_TFLOPS, 4),
           "finite": bool(torch.isfinite(out).all())}
    if name == "c5":
        res["classifier_gflop_per_image_step"] = round((cls_flops_per_image or CLS_FLOPS_ESTIMATE) / 1e9, 1)
        res["classifier_flops_source"] = ("counted from the launches of one guidance evaluation (convolutions, batched "
                                          "GEMMs; GroupNorm / softmax / pooling not counted)" if cls_flops_per_image
                                          else "SURVEY estimate")
        res["guidance_stream"] = "second HIP stream, one reverse step ahead (DDNM_CLS_OVERLAP=0: serial)"
    if roofline and rank == 0 and world == 1 and name == "c3" and not strong:
        try:
            def restore_small(b):
                ddnm_diffusion(torch.randn(b, 3, 256, 256, device=dev), model, betas, 0.85, op, y[:b], cls_fn=None,
                               classes=None, config=cfg, return_cpu=False)
            res["latency"] = latency_probe(model, restore_small, lambda b: (torch.randn(b, 3, 256, 256, device=dev),
                                                                             torch.full((b,), 500.0, device=dev)))
        except Exception as e:    # noqa: BLE001
            res["latency"] = {"error": repr(e)}
    if roofline and rank == 0 and B > 0:
        try:
            timer = ops.KernelTimer()
            ops.set_kernel_timer(timer)
            x_T = torch.randn(B, 3, 256, 256, device=dev)
            t = torch.full((B,), 500.0, device=dev)
            cls = torch.full((B,), 951, dtype=torch.long, device=dev)
            for _ in range(2):
                model(x_T, t, cls) if name == "c5" else model(x_T, t)
            ops.set_kernel_timer(None)
            summ = timer.summary()
            kname, r = max(summ.items(), key=lambda kv: kv[1]["ms"])
            total_ms = sum(v["ms"] for v in summ.values())
            achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
            # algorithmic HBM bytes of the same launches (fp16 tensors: input [+ shortcut input] + weights + output
            # [+ residual], every operand once) -- what `traffic` (PMC) is to be compared with
            alg = []
            for (variant, _, _, _), (b_, ho, wo, cin, cout, k, stride, ups, skipc, _, has_res) in zip(timer.records, timer.shapes):
                if variant == kname:
                    pix_in = b_ * ho * wo // (4 if ups else 1)
                    alg.append(2.0 * (pix_in * cin + b_ * ho * wo * skipc + cout * (k * k * cin + skipc)
                                      + b_ * ho * wo * cout * (2 if has_res else 1)))
            # HBM traffic / MFMA-busy / rocprofv3 launch time of the same kernels from the PMC passes over an ADM forward
            # at B=4 (tools/adm_fwd.py), bound to the loaded binary by its source digest like the c2 figures
            traffic = mfma_busy = frac_rocprof = None
            pmc_note = "no profiles/*_adm_pmc_conv16.json carries the loaded library's source digest"
            hit = None
            if lib_digest:          # the PMC passes are taken at B = 4 (c3 / c4) and at B = 8 (c5): pick this batch's file
                hit = pmc_for_loaded_binary(lib_digest, f"_adm_pmc_conv16_b{B}.json") or \
                    (pmc_for_loaded_binary(lib_digest, "_adm_pmc_conv16.json") if B == 4 else None)
            if hit is not None:
                fname, pj = hit
                traffic, mfma_busy = pj.get("hbm_bytes_per_launch"), pj.get("mfma_busy_frac_weighted")
                if pj.get("rocprof_avg_launch_us"):
                    frac_rocprof = round(r["flops"] / r["launches"] / (pj["rocprof_avg_launch_us"] * 1e-6) / 1e12
                                         / PEAK_F16_TFLOPS, 4)
                pmc_note = f"profiles/{fname}: PMC passes over 2 ADM forwards at B={B} ({pj.get('kernel')}), same source digest"
            res["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_F16_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(achieved / PEAK_F16_TFLOPS, 4), "traffic": traffic,
                               "mfma_busy_pmc": mfma_busy, "frac_rocprof": frac_rocprof, "traffic_note": pmc_note,
                               "traffic_algorithmic": round(sum(alg) / max(1, len(alg)), 1),
                               "mfma_work_tflops": round(achieved, 1),
                               "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                               "avg_flops_per_launch": r["flops"] / r["launches"],
                               "share_of_conv_time": round(r["ms"] / total_ms, 4),
                               "note": "HIP events around every launch of the kernel during 2 UNet forwards at this batch"}
        except Exception as e:    # noqa: BLE001
            ops.set_kernel_timer(None)
            res["roofline"] = {"error": repr(e)}
    del model
    torch.cuda.empty_cache()
    return res


def latency_probe(model, restore, fwd_args, batches=(1, 2)):
    """Small-batch operating point (the reference's shipped `sampling.batch_size: 1`): one full restoration at B = 1 / 2,
    eager and with the forward replayed from a hipGraph (what `Diffusion` switches on for B <= 2), plus the host time to
    ENQUEUE one eager forward against the GPU time of that forward.  `restore(B)` runs one restoration of B images and
    returns after enqueueing; `fwd_args(B)` gives the arguments of one forward."""
    res = {}
    for B in batches:
        ent = {}
        for mode in ("eager", "graph"):
            model.auto_graphs(2 if mode == "graph" else 0)
            restore(B)                                   # warm-up (captures the graph)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            restore(B)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ent[mode] = {"images_per_s": round(B / dt, 4), "ms_per_restoration": round(dt * 1e3, 1)}
        model.auto_graphs(0)
        a = fwd_args(B)
        for _ in range(2):
            model(*a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(*a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        model.auto_graphs(2)
        model(*a)
        e0.record()
        for _ in range(5):
            model(*a)
        e1.record()
        torch.cuda.synchronize()
        model.auto_graphs(0)
        ent["forward"] = {"host_enqueue_ms_eager": round((t1 - t0) * 1e3, 3), "until_done_ms_eager": round((t2 - t0) * 1e3, 3),
                          "gpu_ms_graph_replay": round(e0.elapsed_time(e1) / 5, 3)}
        ent["graph_speedup"] = round(ent["graph"]["images_per_s"] / ent["eager"]["images_per_s"], 3)
        res[f"b{B}"] = ent
    res["note"] = ("one restoration each (T = 100), 1 warm-up; `graph` = the forward replayed from a captured hipGraph "
                   "(ddnm_amd/graph.py), the runner's default for batch sizes <= 2 (DDNM_GRAPH_MAX_BATCH)")
    return res


def pmc_for_loaded_binary(lib_digest, suffix="_pmc_dominant_kernel.json"):
    """HBM traffic / MFMA-busy of the dominant kernel come from separate rocprofv3 --pmc passes (bench.py cannot run
    under the profiler itself); tools/pmc_summary.py stamps them with the digest of the sources the profiled binary was
    built from.  They are reported ONLY when that digest is the loaded library's -- a kernel edit invalidates them."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if f.endswith(suffix):
            try:
                pj = json.load(open(os.path.join(pdir, f)))
            except Exception:      # noqa: BLE001
                continue
            if pj.get("source_digest") == lib_digest:
                best = (f, pj)
    return best


def self_launch(args):
    """`python bench.py --gpus N` without torchrun: re-exec under torch.distributed.run, one rank per GPU (the reference
    takes every visible GPU from one plain command through nn.DataParallel, guided_diffusion/diffusion.py:140,164,180).
    Fails loudly when fewer than N devices are visible -- unless DDNM_DIST_BACKEND=gloo, the 1-GPU test mode in which
    the ranks share one device."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("DDNM_DIST_BACKEND") != "gloo":
        sys.stderr.write(f"[bench] --gpus {args.gpus} but only {ndev} GPU(s) are visible: refusing to measure fewer "
                         f"devices than asked for\n")
        sys.exit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, headline): celeba_hq sr_bicubic 4x B=8/GPU, fp32.  c3 = imagenet_256 colorization, "
                         "c4 = imagenet_256 inpainting with time travel l=10 r=3, c5 = imagenet_256_cc cs_walshhadamard "
                         "0.25 + classifier guidance (ADM UNet in the runner's use_fp16 mode)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="c3 / c4 only: strong = the BASELINE config's global batch (32 / 16) split over the ranks")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default c2 run at 1 GPU: do not append the short c3 / c4 / c5 measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-side-path", action="store_true",
                    help="skip the one-pass measurement of the all-fp32-MFMA engine (kernel-trace profiles of the headline path)")
    args = ap.parse_args()

    from ddnm_amd import _lib
    from ddnm_amd import dist as ddist
    from ddnm_amd import ops
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from ddnm_amd.functions.svd_operators import build_operator
    from ddnm_amd.guided_diffusion.diffusion import get_beta_schedule
    from ddnm_amd.guided_diffusion.models import Model

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                         # does not return
    rank, local_rank, world = ddist.init()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE\n")
    if world > torch.cuda.device_count() and ddist.backend_name() != "gloo":
        sys.stderr.write(f"[bench] {world} ranks but {torch.cuda.device_count()} visible GPU(s)\n")
        sys.exit(2)
    dev = torch.device("cuda", torch.cuda.current_device())
    lib_digest = _lib.lib().ddnm_build_digest().decode()
    ranks_seen = int(round(ddist.reduce_scalar(1.0, dev, "sum")))        # an all_reduce of ones over the real group
    dist_info = {"backend": {"nccl": "nccl (RCCL)"}.get(ddist.backend_name(), ddist.backend_name()),
                 "ranks_seen": ranks_seen}

    cfg = make_config()
    if args.workload != "c2":
        res = adm_workload(args.workload, ddist, rank, world, dev, args.steps, args.warmup,
                           strong=(args.scaling == "strong"), roofline=not args.no_roofline, lib_digest=lib_digest)
        line = {"metric": "restored images/sec @256x256, 100 DDIM steps", "higher_is_better": True, "vs_baseline": None,
                "library_digest": lib_digest[:16]}
        line.update(res)
        line.update(dist_info)
        if rank == 0:
            print(json.dumps(line), flush=True)
        ddist.barrier()
        ddist.shutdown()
        return
    model = Model(cfg, device=dev)
    sd = model.random_state_dict(seed=1234)          # identical replica on every rank, no broadcast
    model.load_state_dict(sd)
    betas = torch.from_numpy(get_beta_schedule("linear", beta_start=1e-4, beta_end=0.02,
                                               num_diffusion_timesteps=1000)).float().to(dev)
    op = build_operator("sr_bicubic", 4, cfg, dev)
    B = BATCH_PER_GPU
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x_orig = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    y = op.A(x_orig)
    torch.cuda.manual_seed(1234 + rank)

    torch.cuda.synchronize()
    # process start -> ready to sample (interpreter + torch import, library load, process group, weight packing, operator
    # set-up), the slowest rank: what a first N-GPU run pays before its steady state (reported, outside the timed region)
    startup_s = ddist.reduce_scalar(time.perf_counter() - T_PROCESS_START, dev, "max")
    passes = [0]

    def one_pass(m=None, seed=None):
        # x_T and the per-iteration noise from the engine's own counter-based generator (in-kernel Philox draw: no ATen
        # op in the timed loop); counter = GLOBAL image index, so an image's noise does not depend on the rank count
        passes[0] += 1
        ph = ops.PhiloxNoise(seed=(0x5EED0000 + passes[0]) if seed is None else seed, image_base=rank * B)
        x_T = ph.tensor(ops.PhiloxNoise.XT_ITER, x_orig)
        xs, _ = ddnm_diffusion(x_T, model if m is None else m, betas, 0.85, op, y, cls_fn=None, classes=None, config=cfg,
                               return_cpu=False, noise=ph)
        return ddist.gather_images(xs[0])            # the path's single collective

    for _ in range(args.warmup):
        out = one_pass()
    ddist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    ddist.barrier()
    dt_rank = time.perf_counter() - t0
    dt, dt_min = ddist.reduce_scalar(dt_rank, dev, "max"), ddist.reduce_scalar(dt_rank, dev, "min")
    assert out.shape[0] == B * world and bool(torch.isfinite(out).all())
    # data-consistency spot check of the last pass (this rank's slice): A x_0 = y
    mine = out[rank * B:(rank + 1) * B]
    resid = (op.A(mine) - y).abs().max().item()

    # the path's single collective on its own: the all_gather of the restored images (inside every timed step above),
    # timed over 5 repetitions after the timed region, slowest rank
    ddist.barrier()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    for _ in range(5):
        ddist.gather_images(mine.contiguous())
    torch.cuda.synchronize()
    gather_ms = ddist.reduce_scalar((time.perf_counter() - tg) / 5 * 1e3, dev, "max")
    value = args.steps * B * world / dt
    line = {
        "metric": baseline_metric(), "value": round(value, 4),
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "ms_per_step_rank_min": round(dt_min / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "backend": dist_info["backend"], "ranks_seen": ranks_seen,
        "startup_s": round(startup_s, 2), "gather_ms": round(gather_ms, 3),
        "timing_note": "startup_s = process start -> ready to sample (imports, library load, process group, weight packing), "
                       "slowest rank, NOT in the timed region; gather_ms = the single all_gather of one step's restored images "
                       "(part of every timed step), slowest rank; ms_per_step = steady state incl. that gather",
        "vs_baseline": None,
        "dtype": ("f32 (3x3 convolutions: fp32 operands carried as hi + lo fp16 halves, three fp16 MFMA products per "
                  "product, fp32 accumulate -- operand error 2^-22, measured closer to fp64 than the fp32 MFMA kernel; "
                  "everything else fp32)") if getattr(model, "split16", False) else "f32",
        "data": "synthetic",
        "noise": "in-kernel Philox4x32-10 + Box-Muller (x_T and every iteration; no ATen RNG in the timed loop)",
        "config": {"workload": "celeba_hq.yml SVD sr_bicubic 4x, sigma_y=0, eta=0.85, T_sampling=100, "
                               "batch_size=8 per GPU (BASELINE configs[1])",
                   "global_batch": B * world, "image": "3x256x256", "parallelism": f"dp{world} (image sharding)"},
        "consistency_max_abs": resid, "library_digest": lib_digest[:16],
    }

    if rank == 0 and not args.no_roofline:
        try:                      # the headline line must be printed even if the instrumented pass fails
            # dominant kernel: the 128x128-tile implicit-GEMM convolution.  One more, instrumented pass of
            # the same workload: HIP events around every convolution launch on the launch stream.
            timer = ops.KernelTimer()
            ops.set_kernel_timer(timer)
            x_T = torch.randn(B, 3, 256, 256, device=dev)
            t = torch.full((B,), 500.0, device=dev)
            for _ in range(3):
                model(x_T, t)
            ops.set_kernel_timer(None)
            summ = timer.summary()
            dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
            name, r = dom
            # algorithmic HBM bytes of the same launches: every operand once (input [+ shortcut input] + weights +
            # output [+ residual]), fp32 -- what `traffic` (PMC) is to be compared with
            alg = []
            for (variant, _, _, _), (b_, ho, wo, cin, cout, k, stride, ups, skipc, _, has_res) in zip(timer.records, timer.shapes):
                if variant == name:
                    pix_in = b_ * ho * wo * stride * stride // (4 if ups else 1)
                    alg.append(4.0 * (pix_in * cin + b_ * ho * wo * skipc + cout * (k * k * cin + skipc)
                                      + b_ * ho * wo * cout * (2 if has_res else 1)))
            total_ms = sum(v["ms"] for v in summ.values())
            achieved = r["flops"] / (r["ms"] * 1e-3) / 1e12
            if name.startswith("conv3x3_halo_s16") or name.startswith("conv3x3_s16_persist"):
                peak = PEAK_SPLIT16_TFLOPS
                peak_note = ("dense fp16 MFMA peak 2500 TFLOP/s / 3: one fp32-grade product = three fp16 MFMA products; "
                             "`achieved` counts algorithmic FLOPs (x3 = %.0f TFLOP/s of MFMA work)" % (3 * achieved))
            else:
                peak, peak_note = PEAK_F32_TFLOPS, "fp32 MFMA peak"
            traffic, mfma_busy, frac_rocprof, pmc_note = None, None, None, \
                "no profiles/*_pmc_dominant_kernel.json carries the loaded library's source digest: traffic / MFMA-busy " \
                "are not reported for this binary"
            hit = pmc_for_loaded_binary(lib_digest)
            if hit is not None:
                fname, pj = hit
                traffic, mfma_busy = pj.get("hbm_bytes_per_launch"), pj.get("mfma_busy_frac_weighted")
                if pj.get("rocprof_avg_launch_us"):
                    frac_rocprof = round(r["flops"] / r["launches"] / (pj["rocprof_avg_launch_us"] * 1e-6) / 1e12
                                         / peak, 4)
                pmc_note = f"HBM bytes per launch = FETCH_SIZE x2 + WRITE_SIZE, MFMA-busy and the rocprofv3 average " \
                           f"launch time from profiles/{fname} (same source digest as the loaded library)"
            line["roofline"] = {
                "kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "peak_note": peak_note,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "frac_rocprof": frac_rocprof, "traffic_note": pmc_note,
                "traffic_algorithmic": round(sum(alg) / max(1, len(alg)), 1),
                "mfma_busy_pmc": mfma_busy,
                "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                "avg_flops_per_launch": r["flops"] / r["launches"],
                "share_of_conv_time": round(r["ms"] / total_ms, 4),
                # MFMA work actually issued by the dominant kernel: 3 fp16 products per algorithmic product in the split
                # form (compare THIS with the 2500 TFLOP/s dense fp16 peak; `achieved` with `peak`)
                "mfma_work_tflops": round(achieved * (3 if (name.startswith("conv3x3_halo_s16") or name.startswith("conv3x3_s16_persist")) else 1), 1),
                "whole_loop_tflops": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world, 2),
                # the loop holds split-fp16, fp32-MFMA and vector kernels, so its algorithmic rate is quoted against both
                # fixed peaks by name: the split bound 2500 / 3 and the fp32 MFMA peak 157.3 of rounds 1-2
                "whole_loop_frac": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world / peak, 4),
                "whole_loop_frac_of_split16_bound_833": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world
                                                              / PEAK_SPLIT16_TFLOPS, 4),
                "whole_loop_frac_of_f32_mfma_peak_157": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world
                                                              / PEAK_F32_TFLOPS, 4),
            }
        except Exception as e:    # noqa: BLE001
            ops.set_kernel_timer(None)
            line["roofline"] = {"error": repr(e)}
    if rank == 0 and not args.no_roofline:
        # hbm_frac = bytes / (t x 6.29 TB/s) of every HBM-bound kernel of the path (sampler steps,
        # FWHT, re-noise, finalize, GroupNorm backward, small-Cout convolution, FiLM projection), and the counter-level
        # stall attribution of the dominant kernels -- from separate rocprofv3 passes (tools/hbm_kernels.py,
        # tools/pmc_stalls.py), reported only when the profile carries the loaded library's source digest
        hit = pmc_for_loaded_binary(lib_digest, "_hbm_kernels.json")
        if hit is not None:
            line["roofline_hbm"] = {"source": f"profiles/{hit[0]}", "peak_TBps": hit[1].get("hbm_achievable_TBps"),
                                    "note": hit[1].get("passes"),
                                    "kernels": [{"kernel": k["kernel"], "bytes_algorithmic": k["bytes_algorithmic"],
                                                 "bytes_pmc": k["bytes_pmc"], "us": k["us"],
                                                 "frac_of_6.29TB/s": k["frac_of_6.29TBps"], "frac_pmc": k["frac_pmc"]}
                                                for k in hit[1].get("kernels", [])]}
        else:
            line["roofline_hbm"] = {"note": "no profiles/*_hbm_kernels.json carries the loaded library's source digest"}
        st = pmc_for_loaded_binary(lib_digest, "_pmc_stalls_headline.json")
        if st is not None and "roofline" in line and isinstance(line["roofline"], dict):
            line["roofline"]["stall_attribution"] = dict(st[1].get("all_launches", {}), source=f"profiles/{st[0]}",
                                                         units=st[1].get("units"))
    if rank == 0 and getattr(model, "split16", False) and not args.no_roofline:
        try:
            line["precision_check"] = precision_check(dev)
        except Exception as e:    # noqa: BLE001
            line["precision_check"] = {"error": repr(e)}
    if world == 1 and not args.no_roofline and not args.no_side_path:
        try:
            def restore_small(b):
                ddnm_diffusion(torch.randn(b, 3, 256, 256, device=dev), model, betas, 0.85, op, y[:b], cls_fn=None,
                               classes=None, config=cfg, return_cpu=False)
            line["latency"] = latency_probe(model, restore_small, lambda b: (torch.randn(b, 3, 256, 256, device=dev),
                                                                              torch.full((b,), 500.0, device=dev)))
        except Exception as e:    # noqa: BLE001
            line["latency"] = {"error": repr(e)}
    if world == 1 and getattr(model, "split16", False) and not args.no_roofline and not args.no_side_path:
        # the same restoration (same noise) on the all-fp32-MFMA engine: its speed, and how far the two results are apart
        try:
            model32 = Model(cfg, device=dev, split16=False)
            model32.load_state_dict(sd)
            one_pass(model32)                                   # warm-up
            out_s = one_pass(seed=4321)                         # same seed: same x_T and noise for both engines
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out_f = one_pass(model32, seed=4321)
            torch.cuda.synchronize()
            dt32 = time.perf_counter() - t1

            line["f32_mfma_path"] = {
                "value": round(B / dt32, 4), "unit": "images/sec",
                "note": "same workload with every convolution on v_mfma_f32_32x32x2_f32 (DDNM_CONV_F32=mfma32), 1 pass; "
                        "the differences are between the two engines' restorations of the same inputs and noise",
                "restoration_rel_l2_diff": float((out_s - out_f).double().norm() / out_f.double().norm()),
                "restoration_max_abs_diff": float((out_s - out_f).abs().max()),
                "restoration_max_abs": float(out_f.abs().max()),
            }
            del model32, out_s, out_f
        except Exception as e:    # noqa: BLE001
            line["f32_mfma_path"] = {"error": repr(e)}
    if world > 1:
        ddist.barrier()
    if not args.no_extra_workloads:
        del model
        torch.cuda.empty_cache()
        line["workloads"] = {}
        done = None
        if world > 1:
            # insurance for the headline line of a multi-rank run: if a collective of the appended workloads hangs, rank 0
            # still prints the line (workloads marked unmeasured) and every rank leaves
            import threading
            done = threading.Event()

            def watchdog(limit=900.0):
                if not done.wait(limit + (0 if rank == 0 else 30)):
                    if rank == 0:
                        line["workloads"] = {"error": f"appended workloads did not finish within {limit:.0f} s: unmeasured"}
                        print(json.dumps(line), flush=True)
                    os._exit(0 if rank == 0 else 1)
            threading.Thread(target=watchdog, daemon=True).start()
        if world == 1:
            # BASELINE configs[2..4] on this GPU (their per-GPU shards), short: 1 warm-up + 2 timed restorations each
            plan = [("c3", False, 2), ("c4", False, 2), ("c5", False, 2)]
        else:
            # N > 1: configs[2] / configs[3] with their fixed GLOBAL batch (32 / 16) split over the ranks present
            # (strong scaling: the driver's per-N lines give the curve), 1 warm-up + 1 timed restoration each
            plan = [("c3", True, 1), ("c4", True, 1)]
        for wname, strong, nsteps in plan:
            try:                       # every rank takes part (collectives inside); a failure must not cost the line
                line["workloads"][wname] = adm_workload(wname, ddist, rank, world, dev, nsteps, 1, strong=strong,
                                                        roofline=not args.no_roofline, lib_digest=lib_digest)
            except Exception as e:    # noqa: BLE001
                line["workloads"][wname] = {"error": repr(e)}
        if done is not None:
            done.set()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, sd)
            except Exception as e:    # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    ddist.barrier()          # (no-ops without a process group; a launcher-started world of one runs them on RCCL)
    ddist.shutdown()


if __name__ == "__main__":
    main()
