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
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= vector) peak
FLOPS_PER_FWD_PER_IMAGE = 498.35e9   # SURVEY.md section 8(d), celeba Model
T_SAMPLING = 100
BATCH_PER_GPU = 8


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


def cpu_baseline(cfg, sd, budget_s=25.0):
    """Reported baseline (not the optimisation target): the oracle restatement of the reference path
    (bit-identical to the reference UNet on CPU, tests/test_oracle_pins.py) on this box's host cores.
    Bounded sample: B=1, as many reverse steps of the 100 as fit in `budget_s`, extrapolated.
    The intra-op thread count is calibrated first on a reduced UNet (oversubscribing a many-core
    host makes the ATen CPU kernels dramatically slower)."""
    from oracle import cases, sampler, unet_celeba
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    mcfg, msd = cases.celeba_net("mid")
    mnet = unet_celeba.Net(msd, mcfg)
    mx, mt = cases.forward_inputs(mcfg, 1)
    best = (None, 1e30)
    for nt in (8, 16, 32, 64, 128):
        if nt > avail:
            break
        torch.set_num_threads(nt)
        mnet(mx, mt)
        t0 = time.perf_counter()
        for _ in range(3):
            mnet(mx, mt)
        dt = (time.perf_counter() - t0) / 3
        if dt < best[1]:
            best = (nt, dt)
    threads = best[0] or min(avail, 8)
    torch.set_num_threads(threads)

    x_orig, x_T, tape = cases.sampler_case(cfg, 1, T_SAMPLING)
    op = cases.make_operator("sr_bicubic", 256)
    y = op.A(x_orig)
    net = unet_celeba.Net(sd, cfg)
    t0 = time.perf_counter()
    net(x_T, torch.tensor([990.0]))          # warm-up forward, also sizes the sample
    t_fwd = time.perf_counter() - t0
    n_steps = int(max(1, min(10, budget_s // max(t_fwd, 1e-3))))

    class Stop(Exception):
        pass

    def record(k, name, t):
        if name == "xt_next" and k == n_steps - 1:
            raise Stop

    t0 = time.perf_counter()
    try:
        sampler.ddnm_diffusion(x_T, net, cases.betas(), 0.85, op, y, tape, record=record)
    except Stop:
        pass
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt / n_steps * T_SAMPLING), "unit": "images/sec", "cores": threads,
            "kind": "port", "sample": f"B=1, {n_steps} of {T_SAMPLING} reverse steps timed ({dt:.1f} s) on {threads} "
                                      f"threads ({avail} logical CPUs visible), extrapolated x{T_SAMPLING / n_steps:g}"}


def bench_adm(args, ddist, rank, world, dev):
    """Informational ImageNet workloads (BASELINE configs[2], configs[3] per-GPU shards): ADM UNet
    (552.81 M params, 2242.87 GFLOP per forward per image), fp16-operand MFMA torso."""
    import types
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion, get_schedule_jump
    from ddnm_amd.functions.svd_operators import Colorization, Inpainting, WalshHadamardCS
    from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier, make_cond_fn
    from ddnm_amd.guided_diffusion.diffusion import get_beta_schedule
    from ddnm_amd.guided_diffusion.unet import create_model
    ns = types.SimpleNamespace
    travel = (10, 3) if args.workload == "c4" else (1, 1)
    cfg = ns(diffusion=ns(num_diffusion_timesteps=1000), data=ns(image_size=256, channels=3),
             time_travel=ns(T_sampling=T_SAMPLING, travel_length=travel[0], travel_repeat=travel[1]))
    model = create_model(image_size=256, num_channels=256, num_res_blocks=2, attention_resolutions="32,16,8",
                         num_head_channels=64, learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
                         use_fp16=True, class_cond=(args.workload == "c5"))
    model.device = dev
    model.load_state_dict(model.random_state_dict(1234))
    model.convert_to_fp16()
    cls_fn = None
    if args.workload == "c5":
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
    betas = torch.from_numpy(get_beta_schedule("linear", beta_start=1e-4, beta_end=0.02,
                                               num_diffusion_timesteps=1000)).float().to(dev)
    B = 8 if args.workload == "c5" else 4
    g = torch.Generator().manual_seed(1234 + rank)
    x_orig = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).to(dev)
    if args.workload == "c3":
        op = Colorization(256, dev)
    elif args.workload == "c5":
        op = WalshHadamardCS(3, 256, 4, torch.randperm(256 * 256, generator=g), dev)
    else:
        mask = (torch.rand(256, 256, generator=g) > 0.26).long().reshape(-1)        # 74 % kept, like exp/inp_masks/mask.npy
        r = torch.nonzero(mask == 0).long().reshape(-1) * 3
        op = Inpainting(3, 256, torch.cat([r, r + 1, r + 2], 0), dev)
    y = op.A(x_orig)
    times = get_schedule_jump(T_SAMPLING, *travel)
    nfe = sum(1 for a, b in zip(times[:-1], times[1:]) if b < a)

    def one_pass():
        x_T = torch.randn(B, 3, 256, 256, device=dev)
        xs, _ = ddnm_diffusion(x_T, model, betas, 0.85, op, y, cls_fn=cls_fn, classes=None, config=cfg)
        return ddist.gather_images(xs[0])

    for _ in range(args.warmup):
        out = one_pass()
    ddist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    torch.cuda.synchronize()
    ddist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = tmax.item()
    value = args.steps * B * world / dt
    tfl = value * nfe * (2243.9e9 + 300e9 if args.workload == "c5" else 2242.87e9) / 1e12 / world
    line = {"metric": "restored images/sec @256x256, 100 DDIM steps", "value": round(value, 4), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16 operands / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": {"c3": "imagenet_256.yml colorization, T_sampling=100, batch 4 per GPU (BASELINE configs[2] shard)",
                                    "c4": "imagenet_256.yml inpainting, time-travel l=10 r=3 (280 NFE + 180 re-noise), "
                                          "batch 4 per GPU (BASELINE configs[3] shard)",
                                    "c5": "imagenet_256_cc.yml cs_walshhadamard ratio 0.25, class-conditional ADM + classifier "
                                          "guidance (class 951, scale 1.0; classifier fwd + input-gradient on fp16 MFMA operands), batch 8 "
                                          "on 1 GPU (BASELINE configs[4])"}[args.workload],
                       "global_batch": B * world, "nfe_per_image": nfe},
            "whole_loop_tflops_per_gpu": round(tfl, 1), "finite": bool(torch.isfinite(out).all())}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        ddist.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, headline): celeba_hq sr_bicubic 4x B=8/GPU, fp32.  Informational extras: "
                         "c3 = imagenet_256 colorization B=4/GPU, c4 = imagenet_256 inpainting with time travel "
                         "l=10 r=3 B=4/GPU, c5 = imagenet_256_cc cs_walshhadamard 0.25 + classifier guidance "
                         "B=8 (ADM UNet, fp16-operand torso like the reference's use_fp16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from ddnm_amd import dist as ddist
    from ddnm_amd import ops
    from ddnm_amd.functions.svd_ddnm import ddnm_diffusion
    from ddnm_amd.functions.svd_operators import build_operator
    from ddnm_amd.guided_diffusion.diffusion import get_beta_schedule
    from ddnm_amd.guided_diffusion.models import Model

    rank, local_rank, world = ddist.init()
    if world != args.gpus:
        if rank == 0:
            sys.stderr.write(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE\n")
    dev = torch.device("cuda", torch.cuda.current_device())

    cfg = make_config()
    if args.workload != "c2":
        return bench_adm(args, ddist, rank, world, dev)
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

    def one_pass():
        x_T = torch.randn(B, 3, 256, 256, device=dev)
        xs, _ = ddnm_diffusion(x_T, model, betas, 0.85, op, y, cls_fn=None, classes=None, config=cfg)
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
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = tmax.item()
    assert out.shape[0] == B * world and bool(torch.isfinite(out).all())
    # data-consistency spot check of the last pass (this rank's slice): A x_0 = y
    mine = out[rank * B:(rank + 1) * B]
    resid = (op.A(mine) - y).abs().max().item()

    value = args.steps * B * world / dt
    line = {
        "metric": baseline_metric(), "value": round(value, 4),
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "celeba_hq.yml SVD sr_bicubic 4x, sigma_y=0, eta=0.85, T_sampling=100, "
                               "batch_size=8 per GPU (BASELINE configs[1])",
                   "global_batch": B * world, "image": "3x256x256", "parallelism": f"dp{world} (image sharding)"},
        "consistency_max_abs": resid,
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
            # HBM traffic / MFMA-busy of the same kernel come from separate rocprofv3 --pmc passes (bench.py cannot
            # run under the profiler itself); tools/pmc_summary.py writes them to profiles/.
            traffic, mfma_busy = None, None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_dominant_kernel.json")
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                traffic, mfma_busy = pj.get("hbm_bytes_per_launch"), pj.get("mfma_busy_frac_weighted")
            line["roofline"] = {
                "kernel": name, "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_TFLOPS, 4), "traffic": traffic,
                "traffic_note": "HBM bytes per launch, FETCH_SIZE x2 + WRITE_SIZE from profiles/r01_pmc_dominant_kernel.json",
                "traffic_algorithmic": round(sum(alg) / max(1, len(alg)), 1),
                "mfma_busy_pmc": mfma_busy,
                "launches": r["launches"], "avg_launch_us": round(r["ms"] * 1e3 / r["launches"], 2),
                "avg_flops_per_launch": r["flops"] / r["launches"],
                "share_of_conv_time": round(r["ms"] / total_ms, 4),
                "whole_loop_tflops": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world, 2),
                "whole_loop_frac": round(value * T_SAMPLING * FLOPS_PER_FWD_PER_IMAGE / 1e12 / world / PEAK_F32_TFLOPS, 4),
            }
        except Exception as e:    # noqa: BLE001
            ops.set_kernel_timer(None)
            line["roofline"] = {"error": repr(e)}
    if world > 1:
        ddist.barrier()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, sd)
            except Exception as e:    # noqa: BLE001
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        ddist.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
