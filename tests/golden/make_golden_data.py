#!/usr/bin/env python
"""Fixture of the image front-end: what the REFERENCE's dataset code makes of the 8 + 8 images it
ships under exp/datasets.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_data.py

  * `center_crop_arr` (datasets/__init__.py:29-44) and `CenterCropLongEdge` (datasets/imagenet_subset.py:5-23) are the
    reference's OWN functions, imported from /root/reference under the torchvision stubs of oracle/ref_import.py;
    CenterCropLongEdge calls torchvision.transforms.functional.center_crop, which is absent here: the stub installed below
    restates it for PIL images (crop box int(round((dim - s) / 2.0)), exactly torchvision's arithmetic);
  * `transforms.Resize` + `transforms.ToTensor` (datasets/__init__.py:138-150, imagenet_subset.py:60-71) are torchvision
    code that is NOT available offline.  On PIL images torchvision's Resize IS `PIL.Image.resize(size[::-1], BILINEAR)` and
    ToTensor IS uint8 HWC -> float CHW / 255, so the fixture applies exactly those two PIL / numpy calls.  This part of the
    row stays "restated, not reference-executed" (DESIGN.md).
Stored per image: sha256 of the uint8 HWC array and its 8 x 8 sub-sample (every 32nd pixel) -- 3 KB in all."""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

REF = "/root/reference"


def digest(arr):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    return {"sha256": hashlib.sha256(arr.tobytes()).hexdigest(), "shape": list(arr.shape),
            "sub8": arr[16::32, 16::32].reshape(-1).tolist()}


def main():
    ref_import.load()
    import torchvision.transforms.functional as TF      # the stub module

    def center_crop(img, size):           # torchvision.transforms.functional.center_crop for a PIL image, square output
        s = size if isinstance(size, int) else size[0]
        w, h = img.size
        left, top = int(round((w - s) / 2.0)), int(round((h - s) / 2.0))
        return img.crop((left, top, left + s, top + s))
    TF.center_crop = center_crop
    # /root/reference/datasets/{__init__,imagenet_subset}.py by path (a `datasets` wheel in site-packages shadows the name)
    import importlib.util

    def by_path(name, path):
        spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    import types
    for name, attr in (("datasets.celeba", "CelebA"), ("datasets.lsun", "LSUN")):     # torchvision dataset classes: not used here
        m = types.ModuleType(name)
        setattr(m, attr, object)
        sys.modules[name] = m
    ref_datasets = by_path("datasets", os.path.join(REF, "datasets", "__init__.py"))
    ref_subset = by_path("datasets.imagenet_subset", os.path.join(REF, "datasets", "imagenet_subset.py"))
    out = {"celeba_hq": {}, "imagenet_center_crop_arr": {}, "imagenet_subset_1k": {}}
    d = os.path.join(REF, "exp/datasets/celeba_hq/face")
    for f in sorted(os.listdir(d)):
        img = Image.open(os.path.join(d, f)).convert("RGB")
        out["celeba_hq"][f] = digest(np.asarray(img.resize((256, 256), Image.BILINEAR)))     # Resize([256, 256]) + ToTensor * 255
    d = os.path.join(REF, "exp/datasets/imagenet/imagenet")
    crop = ref_subset.CenterCropLongEdge()
    for f in sorted(os.listdir(d)):
        img = Image.open(os.path.join(d, f)).convert("RGB")
        out["imagenet_center_crop_arr"][f] = digest(ref_datasets.center_crop_arr(img, 256))      # reference code
        sq = crop(img)                                                                           # reference code
        # transforms.Resize(256) on the square crop (the short side is already the side) + ToTensor * 255
        out["imagenet_subset_1k"][f] = digest(np.asarray(sq.resize((256, 256), Image.BILINEAR)))
        out["imagenet_subset_1k"][f]["square"] = list(sq.size)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "data_frontend.json"), "w"), indent=0)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
