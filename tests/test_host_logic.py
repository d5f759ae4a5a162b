"""CPU tests of the host-side logic that needs no kernel launch."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, schedule, weights


def test_engine_schedule_matches_oracle():
    from ddnm_amd.functions.svd_ddnm import _AlphaTable, compute_alpha, get_schedule_jump
    for args in [(100, 1, 1), (100, 10, 3), (100, 2, 2), (20, 2, 2), (10, 1, 1)]:
        assert get_schedule_jump(*args) == schedule.jump_times(*args)
    betas = cases.betas()
    tab = _AlphaTable(betas)
    for t in (-1, 0, 10, 500, 990, 999):
        assert tab(t).item() == schedule.alpha_bar(betas, t).item()
        assert compute_alpha(betas, torch.tensor([t])).item() == tab(t).item()


def test_engine_model_plan_consumes_reference_state_dict(golden_dir):
    from ddnm_amd.guided_diffusion.models import Model
    keys = json.load(open(f"{golden_dir}/celeba_state_dict_keys.json"))
    m = Model(weights.celeba_config(), device="cpu")
    assert [[k, list(v)] for k, v in m.state_dict_shapes().items()] == keys
    a, b = m.random_state_dict(1234), weights.celeba_state_dict(weights.celeba_config(), 1234)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert m.temb_total == sum(rb.cout for rb in m.res_blocks) and len(m.res_blocks) == 32


def test_conv_weight_packing_layout():
    from ddnm_amd import ops
    w = torch.arange(5 * 3 * 3 * 3, dtype=torch.float32).reshape(5, 3, 3, 3)
    p = ops.pack_conv_weight(w, cin_pad=32)
    assert p.shape == (128, 9, 32)
    assert p[4, 2 * 3 + 1, 2].item() == w[4, 2, 2, 1].item()       # (o, ky, kx, i)
    assert p[5:].abs().sum().item() == 0 and p[:, :, 3:].abs().sum().item() == 0


def test_step_scalars_follow_reference_arithmetic():
    from ddnm_amd import ops
    betas = cases.betas()
    at, atn = schedule.alpha_bar(betas, 500), schedule.alpha_bar(betas, 490)
    s = ops.step_scalars(at, atn, 0.85)
    assert s.sqrt_at == float(at.sqrt()) and s.sqrt_1m_at == float((1 - at).sqrt())
    assert s.c1 == float((1 - atn).sqrt() * 0.85)
    assert s.c2 == float((1 - atn).sqrt() * ((1 - 0.85 ** 2) ** 0.5))
    last = ops.step_scalars(schedule.alpha_bar(betas, 0), schedule.alpha_bar(betas, -1), 0.85)
    assert last.sqrt_at_next == 1.0 and last.c1 == 0.0 and last.c2 == 0.0     # final step is deterministic


def test_cli_parser_matches_reference_flags(tmp_path, monkeypatch):
    import main
    monkeypatch.chdir(tmp_path)
    argv = ["--ni", "--config", "celeba_hq.yml", "--path_y", "celeba_hq", "--eta", "0.85", "--deg", "sr_bicubic",
            "--deg_scale", "4", "--sigma_y", "0.", "-i", "celeba_sr_bc_4"]          # evaluation.sh:7
    args, config = main.parse_args_and_config(argv)
    assert args.deg == "sr_bicubic" and args.deg_scale == 4.0 and args.eta == 0.85 and args.seed == 1234
    assert args.image_folder == os.path.join("exp", "image_samples", "celeba_sr_bc_4")
    assert os.path.isdir(args.image_folder)
    assert config.model.type == "simple" and config.model.ch_mult == [1, 1, 2, 2, 4, 4]
    assert config.time_travel.T_sampling == 100 and config.sampling.batch_size == 1
    flags = {n for names, _ in main.FLAGS for n in names}
    assert {"--config", "--seed", "--exp", "--deg", "--path_y", "--sigma_y", "--eta", "--simplified", "-i",
            "--image_folder", "--deg_scale", "--verbose", "--ni", "--subset_start", "--subset_end", "-n",
            "--noise_type", "--add_noise"} == flags


def test_main_swallows_exceptions_like_reference(tmp_path, monkeypatch):
    """main() logs the traceback and still returns 0 (reference main.py:164-170); without a GPU the
    runner raises (no CPU fallback), which must not change the exit code."""
    import main
    monkeypatch.chdir(tmp_path)
    rc = main.main(["--ni", "--config", "celeba_hq.yml", "--path_y", "synthetic:1", "--deg", "denoising", "-i", "t"])
    assert rc == 0


def test_product_path_never_imports_oracle():
    """The oracle is test infrastructure: nothing under ddnm_amd/, main.py may import it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bad = []
    for dirpath, _, files in os.walk(os.path.join(root, "ddnm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if "import oracle" in src or "from oracle" in src or "/root/reference" in src:
                    bad.append(f)
    src = open(os.path.join(root, "main.py")).read()
    assert "oracle" not in src and not bad, bad


def test_operator_factory_rejects_unknown_deg():
    from ddnm_amd.functions.svd_operators import build_operator
    cfg = weights.celeba_config()
    with pytest.raises(ValueError):
        build_operator("deblur_nope", 0, cfg, "cpu")


def test_bicubic_kernel_matches_oracle():
    from ddnm_amd.functions.svd_operators import bicubic_kernel
    from oracle import operators as O
    assert torch.equal(bicubic_kernel(4), O.bicubic_kernel(4))
    assert abs(bicubic_kernel(4).sum().item() - 1.0) < 1e-6 and bicubic_kernel(4).numel() == 16


def test_data_transform_and_beta_schedule():
    from ddnm_amd.guided_diffusion.diffusion import data_transform, get_beta_schedule
    cfg = weights.celeba_config()
    x = torch.rand(2, 3, 4, 4)
    assert torch.equal(data_transform(cfg, x), 2 * x - 1.0)
    b = get_beta_schedule("linear", beta_start=1e-4, beta_end=0.02, num_diffusion_timesteps=1000)
    assert b.dtype == np.float64 and torch.equal(torch.from_numpy(b).float(), cases.betas())


def test_imagenet_subset_transform_is_center_crop_long_edge_then_bilinear(tmp_path):
    """`subset_1k: true` reads images like datasets/imagenet_subset.py::ImageDataset(normalize=False): CenterCropLongEdge
    (torchvision center_crop to min(w, h), offsets int(round((dim - s) / 2.0))) + Resize(S) bilinear + ToTensor -- NOT the
    guided-diffusion center_crop_arr (box halving + bicubic), which belongs to the out_of_dist folders."""
    import numpy as np
    import torch
    from PIL import Image
    from ddnm_amd.guided_diffusion.diffusion import ImageFolder, ImageList, center_crop_long_edge
    rng = np.random.RandomState(0)
    w, h = 301, 200                                     # odd margin: exercises the rounding of the crop offset
    arr = rng.randint(0, 256, size=(h, w, 3), dtype=np.uint8)
    os.makedirs(tmp_path / "imgs" / "c0")
    Image.fromarray(arr).save(tmp_path / "imgs" / "c0" / "a.png")
    (tmp_path / "list.txt").write_text("c0/a.png 7\nc0/a.png\n")
    ds = ImageList(str(tmp_path / "imgs"), str(tmp_path / "list.txt"), 64)
    x, label = ds[0]
    assert label == 7 and ds[1][1] == -1 and x.shape == (3, 64, 64)
    left = int(round((w - h) / 2.0))
    crop = Image.fromarray(arr[:, left:left + h])
    want = torch.from_numpy(np.asarray(crop.resize((64, 64), Image.BILINEAR))).permute(2, 0, 1).float() / 255
    assert torch.equal(x, want)
    assert center_crop_long_edge(Image.fromarray(arr)).size == (h, h)
    # the out_of_dist folders keep the guided-diffusion transform and differ from it
    ood = ImageFolder(str(tmp_path / "imgs"), 64, transform="center_crop_arr")[0][0]
    assert ood.shape == (3, 64, 64) and not torch.equal(ood, x)


def test_lsun_checkpoint_names_follow_ckpt_util(tmp_path, monkeypatch):
    import types
    from ddnm_amd.guided_diffusion.diffusion import simple_checkpoint_path
    monkeypatch.delenv("XDG_CACHE_HOME", raising=False)
    ns = types.SimpleNamespace
    cfg = lambda ds, cat="": ns(data=ns(dataset=ds, category=cat))      # noqa: E731
    assert simple_checkpoint_path(cfg("CelebA_HQ"), "exp") == "exp/logs/celeba/celeba_hq.ckpt"
    church = simple_checkpoint_path(cfg("LSUN", "church_outdoor"), "exp")
    assert church == "exp/logs/diffusion_models_converted/ema_diffusion_lsun_church_model/model-4432000.ckpt"
    assert simple_checkpoint_path(cfg("LSUN", "cat"), "exp").endswith("ema_diffusion_lsun_cat_model/model-1761000.ckpt")
    assert simple_checkpoint_path(cfg("LSUN", "bedroom"), "exp").endswith("ema_diffusion_lsun_bedroom_model/model-2388000.ckpt")
    with pytest.raises(ValueError):
        simple_checkpoint_path(cfg("CIFAR10"), "exp")


def test_every_reference_config_has_a_counterpart():
    import yaml
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = ["bedroom.yml", "cat.yml", "celeba_hq.yml", "church.yml", "imagenet_256.yml", "imagenet_256_cc.yml", "oldphoto.yml"]
    for n in names:
        cfg = yaml.safe_load(open(os.path.join(root, "configs", n)))
        assert {"model", "diffusion", "data", "sampling", "time_travel"} <= set(cfg), n
    ref_dir = "/root/reference/configs"
    if os.path.isdir(ref_dir):
        assert sorted(os.listdir(ref_dir)) == names
        for n in names:
            assert yaml.safe_load(open(os.path.join(root, "configs", n))) == yaml.safe_load(open(os.path.join(ref_dir, n))), n


def test_respaced_betas_keep_only_integer_timesteps():
    """hq_demo respace.py:96-101 keeps `i in use_timesteps` for integer i: fractional members (np.linspace oversampling)
    are dropped, never truncated into duplicates (beta = 0)."""
    import numpy as np
    from ddnm_amd.hq_demo.respace import respaced_betas
    betas = np.linspace(1e-4, 0.02, 1000)
    new, keep = respaced_betas(betas, {0.0, 10.0, 10.5, 10.9, 20.0, 999.0})
    assert keep == [0, 10, 20, 999] and np.all(new > 0)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` where fewer devices exist must fail loudly (exit 2), never measure one GPU and print
    a line (a directly invoked scaling command produced an n_gpus = 1 line)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DDNM_DIST_BACKEND")}
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "refusing to measure fewer" in r.stderr and "{" not in r.stdout


def test_step_tables_hold_every_reverse_step_time():
    """The per-run timestep table of the sampling loop (one host-to-device copy instead of a fill kernel per step):
    one row per distinct reverse-step time, in the reference's float form `torch.ones(n) * i` (svd_ddnm.py:40)."""
    from ddnm_amd.functions.svd_ddnm import _step_tables, get_schedule_jump
    times = get_schedule_jump(20, 2, 2)
    t_of, cls = _step_tables(times, 50, 3, "cpu", True)
    seen = set()
    for a, c in zip(times[:-1], times[1:]):
        if c < a:
            t = t_of(a * 50)
            assert t.shape == (3,) and t.dtype == torch.float32 and bool((t == float(a * 50)).all())
            seen.add(a * 50)
    assert len(seen) == 20 and cls.tolist() == [951, 951, 951] and cls.dtype == torch.long
    assert _step_tables(times, 50, 3, "cpu", False)[1] is None
