"""The drop-in shell on the GPU: reference command lines through main.py / Diffusion, and the
simplified (--simplified) loop against its oracle restatement."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import rel

pytestmark = pytest.mark.gpu


def _reduced_yaml(tmp_path, T=5, batch=2):
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "celeba_hq.yml")))
    cfg["time_travel"]["T_sampling"] = T
    cfg["sampling"]["batch_size"] = batch
    cfg["data"]["image_size"] = 64
    cfg["model"]["ch_mult"] = [1, 1, 2]
    os.makedirs(tmp_path / "configs", exist_ok=True)
    with open(tmp_path / "configs" / "mini.yml", "w") as f:
        yaml.safe_dump(cfg, f)


@pytest.mark.parametrize("deg,scale", [("sr_bicubic", "4"), ("colorization", "0"), ("cs_walshhadamard", "0.25"),
                                       ("cs_blockbased", "0.25")])
def test_main_cli_end_to_end(hip, tmp_path, monkeypatch, deg, scale, capsys):
    """`python main.py --ni --config ... --deg ... -i ...` (evaluation.sh style) on synthetic images with
    seeded random weights: images are written, PSNR is reported, exit code 0."""
    import main
    _reduced_yaml(tmp_path)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDNM_RANDOM_WEIGHTS", "1")
    rc = main.main(["--ni", "--config", "mini.yml", "--path_y", "synthetic:4", "--eta", "0.85", "--deg", deg,
                    "--deg_scale", scale, "--sigma_y", "0.", "-i", "out_" + deg])
    assert rc == 0
    out = capsys.readouterr().out
    assert "Total Average PSNR" in out and "Number of samples: 4" in out, out
    folder = tmp_path / "exp" / "image_samples" / ("out_" + deg)
    pngs = sorted(p.name for p in folder.glob("*.png"))
    assert pngs == ["0_0.png", "1_0.png", "2_0.png", "3_0.png"]
    assert len(list((folder / "Apy").glob("*.png"))) == 8


def test_main_cli_reports_errors_but_returns_zero(hip, tmp_path, monkeypatch, caplog):
    import main
    _reduced_yaml(tmp_path)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDNM_RANDOM_WEIGHTS", "1")
    rc = main.main(["--ni", "--config", "mini.yml", "--path_y", "synthetic:1", "--deg", "deblur_nope", "-i", "bad"])
    assert rc == 0          # reference main.py:164-170 swallows the ValueError("degradation type not supported")


@pytest.mark.parametrize("deg", ["sr_averagepooling", "colorization", "inpainting", "denoising"])
@pytest.mark.parametrize("sigma_y", [0.0, 0.3])
def test_simplified_loop_vs_oracle(hip, deg, sigma_y, golden_dir, tmp_path, monkeypatch):
    """guided_diffusion/diffusion.py:333-397 (Eq. 19 lambda_t / gamma_t, sigma_t = sqrt(1 - abar'^2))."""
    from ddnm_amd.guided_diffusion.diffusion import simplified_loop
    from ddnm_amd.guided_diffusion.models import Model
    from ddnm_amd.functions import svd_operators as E
    from oracle import cases, sampler, unet_celeba, schedule
    cfg, sd = cases.celeba_net("small")
    cfg.time_travel.T_sampling, cfg.time_travel.travel_length, cfg.time_travel.travel_repeat = 10, 2, 2
    n_it = len(schedule.jump_times(10, 2, 2)) - 1
    x_orig, x_T, tape = cases.sampler_case(cfg, 1, n_it)
    d = cfg.data.image_size
    mask = cases.random_mask(d)
    if deg == "sr_averagepooling":
        A = torch.nn.AdaptiveAvgPool2d((d // 4, d // 4))
        Ap = lambda z: sampler.mean_upsample(z, 4)                         # noqa: E731
        op = E.SuperResolution(3, d, 4, "cuda")
    elif deg == "colorization":
        A = lambda z: (z[:, 0] / 3 + z[:, 1] / 3 + z[:, 2] / 3)[:, None].repeat(1, 3, 1, 1)     # noqa: E731  color2gray
        Ap = lambda z: torch.stack([z[:, 0] * (1 / 3) / (3 * (1 / 3) ** 2)] * 3, 1)            # noqa: E731  gray2color
        op = E.Colorization(d, "cuda", weights=(1 / 3, 1 / 3, 1 / 3))
    elif deg == "inpainting":
        A = Ap = lambda z: z * mask                                                              # noqa: E731
        r = torch.nonzero(mask.reshape(-1) == 0).long().reshape(-1) * 3
        op = E.Inpainting(3, d, torch.cat([r, r + 1, r + 2], 0), "cuda")
    else:
        A = Ap = lambda z: z                                                                     # noqa: E731
        op = E.Denoising(3, d, "cuda")
    y_img = A(x_orig)
    ref, _ = sampler.simplified_ddnm(x_T.clone(), unet_celeba.Net(sd, cfg), cases.betas(), 0.85, A, Ap, y_img, sigma_y,
                                     tape, T_sampling=10, travel_length=2, travel_repeat=2)
    model = Model(cfg)
    model.load_state_dict(sd)
    y_eng = op.A(x_orig.cuda())          # same measurement in the engine operator's own layout
    got = simplified_loop(x_T.cuda(), model, cases.betas().cuda(), 0.85, op, y_eng, sigma_y, cfg,
                          noise=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    assert rel(got, ref) < 2e-4


def test_main_cli_class_conditional_with_classifier_guidance(hip, tmp_path, monkeypatch, capsys):
    """imagenet_256_cc-style run (reduced to 64 px): class-conditional ADM net (fp16 torso), noisy classifier,
    cond_fn gradient, cs_walshhadamard -- the whole BASELINE-config-5 plumbing through main.py."""
    import yaml
    import main
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "imagenet_256_cc.yml")))
    cfg["data"]["image_size"] = 64
    cfg["model"]["image_size"] = 64
    cfg["model"]["num_channels"] = 128
    cfg["model"]["attention_resolutions"] = "16,8"
    cfg["classifier"]["image_size"] = 64
    cfg["classifier"]["classifier_depth"] = 1
    cfg["classifier"]["classifier_attention_resolutions"] = "16,8"
    cfg["time_travel"]["T_sampling"] = 3
    cfg["sampling"]["batch_size"] = 2
    os.makedirs(tmp_path / "configs", exist_ok=True)
    with open(tmp_path / "configs" / "mini_cc.yml", "w") as f:
        yaml.safe_dump(cfg, f)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDNM_RANDOM_WEIGHTS", "1")
    rc = main.main(["--ni", "--config", "mini_cc.yml", "--path_y", "synthetic:2", "--eta", "0.85", "--deg",
                    "cs_walshhadamard", "--deg_scale", "0.25", "--sigma_y", "0.", "-i", "cc"])
    assert rc == 0
    out = capsys.readouterr().out
    assert "Total Average PSNR" in out and "Number of samples: 2" in out, out


def _run_cli(tmp_path, nproc, folder, port):
    """`main.py` on synthetic images: nproc = 1 in-process style launch via torchrun too, so both runs share one code path."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DDNM_RANDOM_WEIGHTS="1", DDNM_DIST_BACKEND="gloo", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "main.py"), "--ni", "--config", "mini.yml",
           "--path_y", "synthetic:5", "--eta", "0.85", "--deg", "sr_averagepooling", "--deg_scale", "4", "--sigma_y", "0.",
           "-i", folder]
    r = subprocess.run(cmd, cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_two_ranks_shard_every_batch_and_match_one_rank(hip, tmp_path):
    """The product CLI under torchrun with 2 ranks (both on this one GPU, gloo backend): every batch of 2 images
    (and the last, ragged batch of 1) is split by image index, ONE gather per batch, rank 0 writes the PNGs -- and the
    files are those of the 1-rank run (an image's noise is a function of (seed, batch, image index) only -- in-kernel Philox
    draws -- so sharding does not change it)."""
    import socket
    from PIL import Image
    _reduced_yaml(tmp_path)

    def port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        return p
    out1 = _run_cli(tmp_path, 1, "one", port())
    out2 = _run_cli(tmp_path, 2, "two", port())
    # three ranks, loader batches of two: a batch SMALLER than the rank count is dealt whole (round-robin), every rank
    # writes its own images and the PSNR is reduced -- the shipped configs' batch_size 1 on a multi-GPU node
    out3 = _run_cli(tmp_path, 3, "three", port())
    assert "Number of samples: 5" in out3 and out3.count("Total Average PSNR") == 1
    d1, d3 = tmp_path / "exp" / "image_samples" / "one", tmp_path / "exp" / "image_samples" / "three"
    assert sorted(p.name for p in d3.glob("*.png")) == [f"{i}_0.png" for i in range(5)]
    assert len(list((d3 / "Apy").glob("*.png"))) == 10
    for i in range(5):          # whole batches: the same launch shapes as the 1-rank run, the same noise -> the same bytes
        assert (d1 / f"{i}_0.png").read_bytes() == (d3 / f"{i}_0.png").read_bytes(), i
    assert "Number of samples: 5" in out1 and "Number of samples: 5" in out2
    assert out1.count("Total Average PSNR") == 1 and out2.count("Total Average PSNR") == 1     # rank 0 only
    psnr = lambda o: float(o.split("Total Average PSNR:")[1].split()[0])                        # noqa: E731
    assert abs(psnr(out1) - psnr(out2)) <= 0.01
    d1, d2 = tmp_path / "exp" / "image_samples" / "one", tmp_path / "exp" / "image_samples" / "two"
    names = sorted(p.name for p in d1.glob("*.png"))
    assert names == sorted(p.name for p in d2.glob("*.png")) == [f"{i}_0.png" for i in range(5)]
    for n in names:
        a = np.asarray(Image.open(d1 / n), dtype=np.int16)
        b = np.asarray(Image.open(d2 / n), dtype=np.int16)
        # equal shapes run bit-identical kernels; batch-size dependent split-K plans may move a value by 1e-6, i.e. at
        # most an isolated 8-bit rounding flip
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3, n


def test_bench_self_launches_ranks_from_a_plain_command(hip):
    """`python bench.py --gpus 2` with no torchrun around it re-executes itself under torch.distributed.run (the
    reference scales from one plain command too: nn.DataParallel, guided_diffusion/diffusion.py:140,164,180).  On this
    1-GPU lease the two ranks share the device through the gloo backend: the printed line must say so -- n_gpus 2,
    ranks_seen 2 (an all_reduce of ones), both ranks' step times.  No scaling number can be measured here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DDNM_DIST_BACKEND="gloo", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-extra-workloads", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"
    assert line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert 0 < line["ms_per_step_rank_min"] <= line["ms_per_step"]
    assert line["value"] > 0 and line["consistency_max_abs"] < 1e-3


def test_main_self_launches_ranks_from_a_plain_command(hip, tmp_path):
    """`python main.py ...` with more than one GPU to use (DDNM_GPUS=2 here; every visible GPU by default) re-executes
    itself as one rank per GPU -- the reference scales from the plain command too (nn.DataParallel,
    guided_diffusion/diffusion.py:140,164,180).  The two ranks share this box's one GPU (gloo); the PNGs are those of the
    single-process run."""
    import subprocess
    import sys
    from PIL import Image
    _reduced_yaml(tmp_path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for n, folder in ((1, "solo"), (2, "duo")):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(DDNM_RANDOM_WEIGHTS="1", DDNM_DIST_BACKEND="gloo", PYTHONPATH=root, DDNM_GPUS=str(n))
        r = subprocess.run([sys.executable, os.path.join(root, "main.py"), "--ni", "--config", "mini.yml", "--path_y",
                            "synthetic:3", "--eta", "0.85", "--deg", "colorization", "--sigma_y", "0.", "-i", folder],
                           cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert r.stdout.count("Total Average PSNR") == 1 and "Number of samples: 3" in r.stdout
        outs[n] = r.stdout
    d1, d2 = tmp_path / "exp" / "image_samples" / "solo", tmp_path / "exp" / "image_samples" / "duo"
    names = sorted(p.name for p in d1.glob("*.png"))
    assert names == sorted(p.name for p in d2.glob("*.png")) and len(names) == 3
    for n in names:
        a = np.asarray(Image.open(d1 / n), dtype=np.int16)
        b = np.asarray(Image.open(d2 / n), dtype=np.int16)
        assert np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-3, n


def test_bench_eight_ranks_on_one_gpu(hip):
    """The driver's first 8-GPU run must not be the first execution of the 8-rank path: `python bench.py --gpus 8` from a
    plain command with the 8 ranks sharing this box's one MI355X (DDNM_DIST_BACKEND=gloo).  Exactly one JSON line,
    n_gpus 8, ranks_seen 8 (an all_reduce of ones over the real group), the weak-scaling headline (8 images per rank),
    the appended strong lines of BASELINE configs[2] (32 images -> 4 per rank = the BASELINE shard) and configs[3]
    (16 images -> 2 per rank, labelled as a variant: BASELINE quotes it on 4 GPUs), all finite; the whole command stays
    far inside the driver's 1800 s and the device memory of 8 replicated models fits the one GPU."""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DDNM_DIST_BACKEND="gloo", PYTHONPATH=root)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1700)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["backend"] == "gloo"
    # the first real N > 1 record explains itself: start-up, the single collective and the steady state are separate figures
    assert line["startup_s"] > 0 and line["gather_ms"] > 0 and line["gather_ms"] < line["ms_per_step"]
    assert line["config"]["global_batch"] == 64 and line["scaling"] == "weak" and line["value"] > 0
    w = line["workloads"]
    assert w["c3"]["scaling"] == "strong" and w["c3"]["config"]["global_batch"] == 32 and w["c3"]["config"]["per_gpu_batch"] == 4
    assert w["c3"]["config"]["reproduces_baseline_config"] is True and w["c3"]["finite"]
    assert w["c4"]["config"]["global_batch"] == 16 and w["c4"]["config"]["per_gpu_batch"] == 2 and w["c4"]["finite"]
    assert w["c4"]["config"]["reproduces_baseline_config"] is False and "4" in w["c4"]["config"]["variant"]
    print(f"[8 ranks on one GPU] wall {wall:.0f} s; weak {line['value']:.2f} img/s, c3 {w['c3']['value']:.2f}, c4 {w['c4']['value']:.2f}")
    assert wall < 1500


def test_bench_four_ranks_reproduce_baseline_config3(hip):
    """`--gpus 4 --workload c4 --scaling strong` = BASELINE configs[3] as quoted: 16 images on 4 GPUs (4 per rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DDNM_DIST_BACKEND="gloo", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--workload", "c4", "--scaling", "strong",
                        "--steps", "1", "--warmup", "0", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 4 and line["ranks_seen"] == 4 and line["config"]["global_batch"] == 16
    assert line["config"]["per_gpu_batch"] == 4 and line["config"]["reproduces_baseline_config"] is True
    assert line["config"]["variant"] is None and line["finite"] and line["value"] > 0
