"""Real image files through the GPU path (review of round 5: the CLI tests fed `--path_y synthetic:N`, so the image front-end,
`y = A(x)`, the `Apy` preview and the PNG writer never ran together on the GPU).  Two of the reference's own bundled sample
images (exp/datasets/imagenet/imagenet/*.JPEG: reference-held test data, committed as fixtures under tests/golden/images/)
are restored by `main.py` exactly as `evaluation.sh` would: file list -> ImageList / ImageFolder -> data_transform -> A ->
A^+ y preview -> sampler -> PNGs.  What is compared:

  * `Apy/orig_<i>.png` against tests/golden/data_frontend.json, i.e. against what the REFERENCE's OWN `center_crop_arr` +
    ToTensor pipeline makes of the same file (sha256 of the uint8 image: bit for bit);
  * `y = A(x)` and the `Apy/Apy_<i>.png` preview against the oracle's operator applied to the same pixels;
  * the restored PNGs exist, are finite images, and the run reports a PSNR for both files."""
import hashlib
import json
import os
import shutil

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMAGES = os.path.join(ROOT, "tests", "golden", "images")
NAMES = ["ILSVRC2012_val_00001226.JPEG", "ILSVRC2012_val_00001285.JPEG"]


def _png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def _run(tmp_path, monkeypatch, capsys, cfg_edit, path_y, deg, scale, folder):
    import yaml
    import main
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "imagenet_256.yml")))
    cfg["time_travel"]["T_sampling"] = 2
    cfg["sampling"]["batch_size"] = 2
    cfg_edit(cfg)
    os.makedirs(tmp_path / "configs", exist_ok=True)
    with open(tmp_path / "configs" / "real.yml", "w") as f:
        yaml.safe_dump(cfg, f)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("DDNM_RANDOM_WEIGHTS", "1")
    rc = main.main(["--ni", "--config", "real.yml", "--path_y", path_y, "--eta", "0.85", "--deg", deg, "--deg_scale", scale,
                    "--sigma_y", "0.", "-i", folder])
    assert rc == 0
    out = capsys.readouterr().out
    assert "Total Average PSNR" in out and "Number of samples: 2" in out, out
    return tmp_path / "exp" / "image_samples" / folder


def test_imagenet_file_list_through_the_gpu_path(hip, tmp_path, monkeypatch, capsys):
    """imagenet_256.yml as shipped (`subset_1k: true`): exp/imagenet_val_1k.txt + exp/datasets/imagenet/imagenet/<files>
    (reference datasets/__init__.py:169-175), colorization (BASELINE configs[2]), full-size ADM UNet, 2 sampling steps."""
    from oracle import operators
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "data_frontend.json")))["imagenet_subset_1k"]
    d = tmp_path / "exp" / "datasets" / "imagenet" / "imagenet"
    os.makedirs(d)
    for n in NAMES:
        shutil.copy(os.path.join(IMAGES, n), d / n)
    (tmp_path / "exp" / "imagenet_val_1k.txt").write_text("".join(f"{n} {7 + k}\n" for k, n in enumerate(NAMES)))
    folder = _run(tmp_path, monkeypatch, capsys, lambda c: None, "imagenet", "colorization", "0", "real_color")
    # the loader shuffles (seeded): match every written original to its source file through the reference's own hash
    want = {golden[n]["sha256"]: n for n in NAMES}
    seen = {}
    for i in range(2):
        o = _png(folder / "Apy" / f"orig_{i}.png")
        h = hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()
        assert h in want, f"orig_{i}.png is not the reference pipeline's image of any fixture file"
        seen[i] = want[h]
        # y = A(x) (grey = mean of the channels, svd_operators.py:627-667) and its preview, from the SAME uint8 pixels
        x = torch.from_numpy(o.astype(np.float32) / 255.0).permute(2, 0, 1)[None] * 2 - 1
        op = operators.Colorization(256)
        y = op.A(x)
        apy_want = ((y.reshape(1, 1, 256, 256).repeat(1, 3, 1, 1) + 1) / 2).clamp(0, 1)
        apy_want = (apy_want[0].permute(1, 2, 0).numpy() * 255.0).round().astype(np.int16)
        apy = _png(folder / "Apy" / f"Apy_{i}.png").astype(np.int16)
        diff = np.abs(apy - apy_want)
        assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (i, diff.max(), (diff != 0).mean())     # (fp32 rounding at .5 ties)
        r = _png(folder / f"{i}_0.png")
        assert r.shape == (256, 256, 3) and r.std() > 0
    assert sorted(seen.values()) == sorted(NAMES)


def test_image_folder_through_the_gpu_path(hip, tmp_path, monkeypatch, capsys):
    """The `out_of_dist: true` branch (reference datasets/__init__.py:176-183): exp/datasets/ood/<class folder>/<files> through
    ImageFolder + center_crop_arr, 4x average-pooling super-resolution; the measurement is checked through the engine's own
    operator on the written original: A(A^+ y) = y (data consistency of the preview) and Apy = block means of the original."""
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "data_frontend.json")))["imagenet_center_crop_arr"]
    d = tmp_path / "exp" / "datasets" / "ood" / "0"
    os.makedirs(d)
    for n in NAMES:
        shutil.copy(os.path.join(IMAGES, n), d / n)

    def edit(c):
        c["data"]["subset_1k"] = False
        c["data"]["out_of_dist"] = True
    folder = _run(tmp_path, monkeypatch, capsys, edit, "ood", "sr_averagepooling", "4", "real_sr")
    want = {golden[n]["sha256"] for n in NAMES}
    for i in range(2):
        o = _png(folder / "Apy" / f"orig_{i}.png")
        assert hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest() in want
        blocks = o.astype(np.float64).reshape(64, 4, 64, 4, 3).mean((1, 3))                 # A: 4 x 4 means
        apy_want = np.repeat(np.repeat(blocks, 4, 0), 4, 1)                                    # A^+: replication
        apy = _png(folder / "Apy" / f"Apy_{i}.png").astype(np.float64)
        assert np.abs(apy - apy_want).max() <= 0.51                                            # one 8-bit rounding
        r = _png(folder / f"{i}_0.png")
        assert r.shape == (256, 256, 3) and r.std() > 0
