"""`hq_demo` (arbitrary-size restoration with the mask-shift trick): oracle
restatement pinned to goldens generated from the reference's own hq_demo package (CPU), HIP engine vs both (GPU)."""
import numpy as np
import pytest
import torch

from oracle import cases, classifier as ocls, hq_cases, hq_demo as H, unet_adm, weights
from tests.helpers import rel


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(f"{golden_dir}/hq_demo.npz")


@pytest.mark.parametrize("key", list(hq_cases.SCHEDULES))
def test_schedule_jump_golden(golden, key):
    assert H.schedule_jump(**hq_cases.SCHEDULES[key]) == golden[f"schedule_{key}"].tolist()


@pytest.mark.parametrize("key", list(hq_cases.RESPACINGS))
def test_space_timesteps_golden(golden, key):
    steps, resp = hq_cases.RESPACINGS[key]
    assert H.space_timesteps(steps, resp) == golden[f"space_{key}"].tolist()


def test_posterior_tables_golden(golden):
    p = H.Posterior(1000, "8")
    assert p.timestep_map == golden["timestep_map"].tolist()
    for mine, name in ((p.betas, "betas_respaced"), (p.variance, "posterior_variance"), (p.coef1, "posterior_mean_coef1"),
                       (p.coef2, "posterior_mean_coef2")):
        assert np.allclose(mine, golden[name], rtol=1e-13, atol=0)


def test_tile_plan_geometry():
    assert H.tile_plan(256, 256) == [(0, 0, 0, 0)]
    assert H.tile_plan(256, 384) == [(0, 0, 0, 0), (0, 128, 128, 0)]
    # 288 x 320: both directions end with an irregular shift that re-uses 256 - (size % 128) restored pixels
    assert H.tile_plan(288, 320) == [(0, 0, 0, 0), (0, 64, 192, 0), (32, 0, 0, 224), (32, 64, 192, 224)]
    with pytest.raises(ValueError):
        H.tile_plan(128, 512)


def _oracle_models(case):
    mcfg = hq_cases.model_config(case)
    net = unet_adm.Net(weights.adm_state_dict(mcfg, cases.SEED), mcfg)
    model = lambda x, t, y: net(x, t.float(), y)                                         # noqa: E731
    cond_fn = None
    if case["classifier"]:
        cc = hq_cases.classifier_config(case)
        csd = weights.classifier_state_dict(cc)
        cond_fn = lambda x, t, y: ocls.cond_fn(csd, cc, x, t.float(), y)                 # noqa: E731
    return model, cond_fn


@pytest.mark.parametrize("name", list(hq_cases.CASES_ALL))
def test_oracle_hq_demo_golden(golden, name):
    case = hq_cases.CASES_ALL[name]
    gt, x_init, tape = hq_cases.inputs(case)
    model, cond_fn = _oracle_models(case)
    face = bool(case.get("face"))
    final, _, _ = H.restore(model, gt, case["deg"], case["scale"], case["sigma_y"], case["resize_y"], x_init, tape,
                            classes=None if face else torch.full((1,), case["class"], dtype=torch.long), cond_fn=cond_fn,
                            mask=hq_cases.keep_mask(case) if face else None,
                            timestep_respacing=case["timestep_respacing"], schedule=case["schedule"])
    ref = torch.from_numpy(golden[f"{name}_final"])
    assert rel(final[:, :, ::4, ::4], ref) < 2e-5
    st = golden[f"{name}_final_stats"]
    assert abs(final.double().abs().sum().item() - st[2]) < 2e-5 * st[2]


# ------------------------------------------------------------------------------------------------ engine, host side
@pytest.mark.parametrize("key", list(hq_cases.SCHEDULES))
def test_engine_schedule_jump(golden, key):
    from ddnm_amd.hq_demo import get_schedule_jump
    assert get_schedule_jump(**hq_cases.SCHEDULES[key]) == golden[f"schedule_{key}"].tolist()


def test_engine_respacing_and_tables(golden):
    from ddnm_amd.hq_demo import create_model_and_diffusion, space_timesteps, tile_plan
    from ddnm_amd.hq_demo.script_util import create_gaussian_diffusion
    for key, (steps, resp) in hq_cases.RESPACINGS.items():
        assert sorted(space_timesteps(steps, resp)) == golden[f"space_{key}"].tolist()
    assert len(space_timesteps(1000, "ddim50")) == 50
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing="8")
    assert d.timestep_map == golden["timestep_map"].tolist()
    for mine, name in ((d.betas, "betas_respaced"), (d.posterior_variance, "posterior_variance"),
                       (d.posterior_mean_coef1, "posterior_mean_coef1"), (d.posterior_mean_coef2, "posterior_mean_coef2")):
        assert np.allclose(mine, golden[name], rtol=1e-13, atol=0)
    for hw in ((256, 256), (256, 384), (288, 320), (640, 512), (300, 700)):
        assert tile_plan(*hw) == H.tile_plan(*hw)
    assert callable(create_model_and_diffusion)


def test_engine_conf_and_cli_flags(tmp_path):
    import importlib.util
    import os
    from ddnm_amd.hq_demo.conf import Default_Conf, yamlread
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    conf = Default_Conf()
    conf.update(yamlread(os.path.join(root, "hq_demo", "confs", "inet256.yml")))
    assert conf.name == "inet256" and conf.no_such_key is None and conf.timestep_respacing == "100"
    assert conf.schedule_jump_params == dict(t_T=100, n_sample=1, jump_length=10, jump_n_sample=3)
    spec = importlib.util.spec_from_file_location("hq_main", os.path.join(root, "hq_demo", "main.py"))
    hq_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hq_main)
    a = hq_main.parse(["--resize_y", "--deg", "sr_averagepooling", "--scale", "2", "--class", "130", "-i", "flamingo"])
    assert a["resize_y"] and a["scale"] == 2 and a["class"] == 130 and a["save_path"] == "flamingo" and a["sigma_y"] == 0.0


# ------------------------------------------------------------------------------------------------ engine, GPU
def _engine_models(case, fp16=False):
    from ddnm_amd.guided_diffusion.classifier import create_classifier
    from ddnm_amd.guided_diffusion.unet import create_model
    mcfg = hq_cases.model_config(case)
    m = create_model(**vars(mcfg.model))
    m.load_state_dict(weights.adm_state_dict(mcfg, cases.SEED))
    if fp16:
        m.convert_to_fp16()
    model_fn = lambda x, t, y=None, **kw: m(x, t, y)                                     # noqa: E731
    cond_fn = None
    if case["classifier"]:
        cc = hq_cases.classifier_config(case)
        keys = ("image_size", "classifier_use_fp16", "classifier_width", "classifier_depth",
                "classifier_attention_resolutions", "classifier_use_scale_shift_norm", "classifier_resblock_updown",
                "classifier_pool")
        c = create_classifier(**{k: getattr(cc, k) for k in keys})
        c.load_state_dict(weights.classifier_state_dict(cc))
        cond_fn = lambda x, t, y=None, **kw: c.log_prob_grad(x, t.float(), y)            # noqa: E731
    return model_fn, cond_fn


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(hq_cases.CASES))
def test_engine_hq_demo_vs_reference_golden(hip, golden, name, tmp_path, monkeypatch):
    from ddnm_amd.hq_demo.script_util import create_gaussian_diffusion
    monkeypatch.chdir(tmp_path)
    case = hq_cases.CASES[name]
    gt, x_init, tape = hq_cases.inputs(case)
    model_fn, cond_fn = _engine_models(case)
    conf = hq_cases.conf_dict(case)
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing=case["timestep_respacing"],
                                          conf=conf)
    kw = {"gt": gt.cuda(), "scale": case["scale"], "deg": case["deg"], "resize_y": case["resize_y"],
          "sigma_y": case["sigma_y"], "save_path": name, "y": torch.full((1,), case["class"], dtype=torch.long).cuda()}
    res = diffusion.p_sample_loop(model_fn, (1, 3, 256, 256), noise=x_init.cuda(), clip_denoised=True, model_kwargs=kw,
                                  cond_fn=cond_fn, device="cuda", progress=False, return_all=True, conf=conf,
                                  noise_tape=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    final = res["sample"].cpu()
    ref = torch.from_numpy(golden[f"{name}_final"])
    assert rel(final[:, :, ::4, ::4], ref) < 3e-4
    st = golden[f"{name}_final_stats"]
    assert abs(final.double().abs().sum().item() - st[2]) < 1e-4 * st[2]
    assert rel(res["x0_t"].cpu()[:, :, ::8, ::8], torch.from_numpy(golden[f"{name}_last_sample"])) < 3e-4
    # the reference's result files
    for sub in ("y", "Apy", "final"):
        assert (tmp_path / "results" / name / sub / "00000.png").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(hq_cases.FACE_CASES))
def test_engine_hq_demo_face_degradations(hip, golden, name, tmp_path, monkeypatch):
    """face256-only degradations of hq_demo (inpainting, mask_color_sr) with the loader's keep-mask."""
    from ddnm_amd.guided_diffusion.unet import create_model
    from ddnm_amd.hq_demo.script_util import create_gaussian_diffusion
    monkeypatch.chdir(tmp_path)
    case = hq_cases.FACE_CASES[name]
    gt, x_init, tape = hq_cases.inputs(case)
    mcfg = hq_cases.model_config(case)
    m = create_model(**vars(mcfg.model))
    m.load_state_dict(weights.adm_state_dict(mcfg, cases.SEED))
    conf = hq_cases.conf_dict(case)
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing=case["timestep_respacing"],
                                          conf=conf)
    kw = {"gt": gt.cuda(), "scale": case["scale"], "deg": case["deg"], "resize_y": False, "sigma_y": 0.0,
          "save_path": None, "gt_keep_mask": hq_cases.keep_mask(case).cuda(),
          "y": torch.full((1,), case["class"], dtype=torch.long).cuda()}
    res = diffusion.p_sample_loop(lambda x, t, y=None, **k: m(x, t), (1, 3, 256, 256), noise=x_init.cuda(),
                                  model_kwargs=kw, device="cuda", progress=False, return_all=True, conf=conf,
                                  noise_tape=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    assert rel(res["sample"].cpu()[:, :, ::4, ::4], torch.from_numpy(golden[f"{name}_final"])) < 3e-4


@pytest.mark.gpu
def test_engine_hq_demo_vs_oracle_with_trace(hip, tmp_path, monkeypatch):
    """Step-level parity on the 2x2 ragged-tile case: every x0_hat and x_{t-1} of the oracle."""
    from ddnm_amd.hq_demo.script_util import create_gaussian_diffusion
    monkeypatch.chdir(tmp_path)
    case = hq_cases.CASES["sr4_tiles"]
    gt, x_init, tape = hq_cases.inputs(case)
    model, _ = _oracle_models(case)
    trace = []
    final_o, y_o, apy_o = H.restore(model, gt, case["deg"], case["scale"], case["sigma_y"], case["resize_y"], x_init, tape,
                                    classes=torch.full((1,), case["class"], dtype=torch.long),
                                    timestep_respacing=case["timestep_respacing"], schedule=case["schedule"], trace=trace)
    model_fn, _ = _engine_models(case)
    conf = hq_cases.conf_dict(case)
    diffusion = create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing=case["timestep_respacing"],
                                          conf=conf)
    kw = {"gt": gt.cuda(), "scale": case["scale"], "deg": case["deg"], "resize_y": True, "sigma_y": 0.0, "save_path": None,
          "y": torch.full((1,), case["class"], dtype=torch.long).cuda()}
    res = diffusion.p_sample_loop(model_fn, (1, 3, 256, 256), noise=x_init.cuda(), model_kwargs=kw, device="cuda",
                                  progress=False, return_all=True, conf=conf, noise_tape=[n.cuda() for n in tape])
    torch.cuda.synchronize()
    assert rel(res["y"].cpu(), y_o) < 1e-6 and rel(res["Apy"].cpu(), apy_o) < 1e-6
    assert rel(res["sample"].cpu(), final_o) < 3e-4
    assert not (tmp_path / "results").exists()          # save_path None: nothing written


@pytest.mark.gpu
def test_hq_demo_cli_end_to_end(hip, tmp_path, monkeypatch):
    """`python hq_demo/main.py --resize_y --config ... --path_y img.png --class 950 --deg sr_averagepooling --scale 4 -i x`
    (hq_demo/evaluation.sh) with a reduced YAML and seeded random weights: full-size PNGs are written."""
    import importlib.util
    import os
    import yaml
    from PIL import Image
    from ddnm_amd.hq_demo.conf import Default_Conf, yamlread
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yamlread(os.path.join(root, "hq_demo", "confs", "inet256.yml"))
    cfg.update(num_channels=128, channel_mult="1,1,2,2", attention_resolutions="32", num_res_blocks=1,
               timestep_respacing="4", schedule_jump_params=dict(t_T=4, n_sample=1, jump_length=2, jump_n_sample=2),
               classifier_scale=0.0, show_progress=False)
    os.makedirs(tmp_path / "confs")
    with open(tmp_path / "confs" / "mini.yml", "w") as f:
        yaml.safe_dump(cfg, f)
    rng = np.random.RandomState(0)
    Image.fromarray(rng.randint(0, 255, size=(80, 96, 3), dtype=np.uint8)).save(tmp_path / "lowres.png")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDNM_RANDOM_WEIGHTS", "1")
    spec = importlib.util.spec_from_file_location("hq_main", os.path.join(root, "hq_demo", "main.py"))
    hq_main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hq_main)
    args = hq_main.parse(["--resize_y", "--config", "confs/mini.yml", "--path_y", "lowres.png", "--class", "950", "--deg",
                          "sr_averagepooling", "--scale", "4", "-i", "demo_out"])
    conf = Default_Conf()
    conf.update(yamlread(args["config"]))
    res = hq_main.main(conf, args)
    torch.cuda.synchronize()
    assert res["sample"].shape == (1, 3, 320, 384) and bool(torch.isfinite(res["sample"]).all())
    final = Image.open(tmp_path / "results" / "demo_out" / "final" / "00000.png")
    assert final.size == (384, 320)
    assert Image.open(tmp_path / "results" / "demo_out" / "y" / "00000.png").size == (96, 80)
    # consistency of the noise-free result: A(x) == y on every tile interior written last (bottom-right tile)
    from ddnm_amd.hq_demo.gaussian_diffusion import avg_pool
    y_back = avg_pool(res["sample"][:, :, 64:, 128:].contiguous(), 4)
    assert rel(y_back.cpu(), res["y"][:, :, 16:, 32:].cpu()) < 1e-4
