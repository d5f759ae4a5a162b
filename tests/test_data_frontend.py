"""Image front-end against tests/golden/data_frontend.json -- what the REFERENCE's dataset code
(its own `center_crop_arr` and `CenterCropLongEdge`, imported from /root/reference; torchvision's Resize / ToTensor restated
as the PIL / numpy calls they are for PIL images, tests/golden/make_golden_data.py) makes of the 8 + 8 images under
exp/datasets.  The images live in the reference tree only, so the comparison runs where /root/reference exists (the build
container); everywhere the fixture itself is checked for shape."""
import hashlib
import json
import os

import numpy as np
import pytest

REF = "/root/reference/exp/datasets"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "data_frontend.json")))


def test_fixture_is_complete(golden):
    assert {k: len(v) for k, v in golden.items()} == {"celeba_hq": 8, "imagenet_center_crop_arr": 8, "imagenet_subset_1k": 8}
    for grp in golden.values():
        for ent in grp.values():
            assert ent["shape"] == [256, 256, 3] and len(ent["sub8"]) == 8 * 8 * 3 and len(ent["sha256"]) == 64


def _check(x, ent, name):
    arr = (x.permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)       # ToTensor^-1: the uint8 HWC image
    assert list(arr.shape) == ent["shape"], name
    assert arr[16::32, 16::32].reshape(-1).tolist() == ent["sub8"], name
    assert hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest() == ent["sha256"], name


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's sample images are not on this machine")
def test_engine_front_end_reproduces_the_reference_pipeline(golden, tmp_path):
    from ddnm_amd.guided_diffusion.diffusion import ImageFolder, ImageList
    ds = ImageFolder(os.path.join(REF, "celeba_hq"), 256, transform="resize")            # CelebA_HQ / FFHQ branch
    assert len(ds) == 8
    for i in range(8):
        x, cls = ds[i]
        assert cls == 0 and x.shape == (3, 256, 256) and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
        _check(x, golden["celeba_hq"][os.path.basename(ds.items[i][0])], ds.items[i][0])
    ds = ImageFolder(os.path.join(REF, "imagenet"), 256, transform="center_crop_arr")    # ImageNet / out_of_dist branch
    for i in range(len(ds)):
        x, _ = ds[i]
        _check(x, golden["imagenet_center_crop_arr"][os.path.basename(ds.items[i][0])], ds.items[i][0])
    names = sorted(golden["imagenet_subset_1k"])
    lst = tmp_path / "val.txt"
    lst.write_text("".join(f"{n} {k}\n" for k, n in enumerate(names)))
    ds = ImageList(os.path.join(REF, "imagenet", "imagenet"), str(lst), 256)              # subset_1k branch
    for i, n in enumerate(names):
        x, label = ds[i]
        assert label == i
        _check(x, golden["imagenet_subset_1k"][n], n)
