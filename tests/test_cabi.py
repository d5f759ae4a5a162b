"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/ddnm_hip.h declares; the ctypes prototypes cover exactly that set; struct layouts
match the header.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ddnm_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddnm_[A-Za-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from ddnm_amd import _lib, build
    build.build()                      # hipcc cross-compiles for gfx950 without a GPU
    return _lib.lib()


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ddnm_hip.h but not exported"


def test_prototypes_match_header():
    from ddnm_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()


def test_version_and_error_strings(lib):
    from ddnm_amd import _lib
    assert lib.ddnm_version() == _lib.ABI_VERSION == 7
    assert b"shape" in lib.ddnm_error_string(-2)
    assert b"bad argument" in lib.ddnm_error_string(-1)
    assert lib.ddnm_error_string(0) == b"success"


def test_struct_layouts_match_header():
    from ddnm_amd._lib import ConvDesc, GemmDesc, StepScalars
    # 9 pointers + 16 int32 + pointer + int64 + 2 int32 + pointer + 3 pointers + 2 int32 + float + int32 (8-byte aligned)
    # + the operand-bound pointer of ABI 5
    assert ctypes.sizeof(ConvDesc) == 9 * 8 + 16 * 4 + 8 + 8 + 8 + 8 + 3 * 8 + 8 + 8 + 8
    assert ConvDesc.amax_in.offset == ctypes.sizeof(ConvDesc) - 8
    assert ConvDesc.workspace.offset == 9 * 8 + 16 * 4
    assert ctypes.sizeof(GemmDesc) == 4 * 8 + 10 * 4 + 8 * 8 + 2 * 4 + 2 * 4
    assert ctypes.sizeof(StepScalars) == 24 + 6 * 4          # + the in-kernel noise fields of ABI 6
    src = open(HEADER).read()
    conv = src[src.index("typedef struct ddnm_conv_desc"):src.index("} ddnm_conv_desc;")]
    fields = re.findall(r"\b(?:const\s+)?(?:float|int32_t|int64_t)\s*\*?\s*([A-Za-z0-9_, ]+);", conv)
    names = [n.strip() for grp in fields for n in grp.split(",")]
    assert names == [f[0] for f in ConvDesc._fields_]


def test_conv16_struct_layout_matches_header():
    from ddnm_amd._lib import Conv16Desc
    src = open(HEADER).read()
    body = src[src.index("typedef struct ddnm_conv16_desc"):src.index("} ddnm_conv16_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:const\s+)?(?:void|float|int32_t|int64_t)\s*\*?\s*([A-Za-z0-9_, \*]+);", body)
    names = [n.strip().lstrip("*") for grp in fields for n in grp.split(",")]
    assert names == [f[0] for f in Conv16Desc._fields_]
    assert ctypes.sizeof(Conv16Desc) == 10 * 8 + 8 + 10 * 4 + 3 * 8 + 4 * 4 + 5 * 8 + 4 * 4


def test_binary_identifies_itself(lib):
    """The binary carries the digest of the sources it was built from and the sizes of its descriptor structs; the
    loader refuses a binary that disagrees with the sources next to it (stale .so ran silently)."""
    from ddnm_amd import _lib, build
    from ddnm_amd._lib import Conv16Desc, ConvDesc, GemmDesc, StepScalars
    assert lib.ddnm_build_digest().decode() == build._digest()
    for i, st in enumerate((ConvDesc, GemmDesc, Conv16Desc, StepScalars)):
        assert lib.ddnm_sizeof(i) == ctypes.sizeof(st)
    assert lib.ddnm_sizeof(99) == -1


def test_stale_binary_is_refused(monkeypatch):
    from ddnm_amd import _lib, build
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("DDNM_NO_AUTOBUILD", "1")
    monkeypatch.setattr(build, "_digest", lambda: "0" * 64)          # pretend the sources changed
    with pytest.raises(_lib.DDNMHipError, match="different sources"):
        _lib.lib()


def test_argument_validation_without_gpu(lib):
    """Entry points reject bad descriptors before touching the device."""
    from ddnm_amd._lib import ConvDesc, GemmDesc
    d = ConvDesc()
    assert lib.ddnm_conv2d_f32(ctypes.byref(d), None) == -1
    g = GemmDesc()
    assert lib.ddnm_bgemm_f32(ctypes.byref(g), None) == -1
    assert lib.ddnm_gn_nchunk(256 * 256, 128) == 128
    assert lib.ddnm_gn_nchunk(64, 1024) == 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from ddnm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("DDNM_NO_AUTOBUILD", "1")
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DDNMHipError):
        _lib.lib()


def test_conv16_plans_without_a_gpu(lib):
    """The planning entry points are host code: which launches the fp16 convolution accepts, whether they slice K
    (workspace) and how many GroupNorm partial tiles they emit -- the ADM layer shapes of BASELINE configs 3-5."""
    from ddnm_amd._lib import Conv16Desc

    def plan(B, H, Cin, Cout, k, **kw):
        d = Conv16Desc()
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize = B, H, H, Cin, Cout, k
        for key, v in kw.items():
            setattr(d, key, v)
        ok = lib.ddnm_conv16_supported(ctypes.byref(d)) == 1
        return ok, (lib.ddnm_conv16_workspace_floats(ctypes.byref(d)) if ok else None), \
            (lib.ddnm_conv16_stats_tiles(ctypes.byref(d)) if ok else None)

    assert plan(4, 256, 256, 256, 3) == (True, 0, 256)              # 1024 tiles of 256 pixels, no slicing
    assert plan(4, 64, 512, 512, 3) == (True, 0, 32)                # 256 tiles of 128 pixels
    ok, ws, tiles = plan(4, 16, 1024, 1024, 3)                      # 32 tiles -> 8 slices of fp32 slabs
    assert ok and ws == 8 * 4 * 16 * 16 * 1024 and tiles > 0
    assert plan(4, 16, 1024, 1024, 1) == (True, 0, 4)               # 1x1 with K <= 1024: never sliced; 64-pixel tiles
    ok, ws, _ = plan(4, 8, 9216, 1024, 1)                           # the im2col'ed 8x8 level: K = 9216 is sliced
    assert ok and ws > 0
    assert plan(4, 8, 1024, 1024, 3)[0] is False                    # 8x8 images have no 3x3 pixel tile (im2col route)
    assert plan(4, 256, 256, 6, 3, out_nchw_f32=1)[:2] == (True, 0)    # the fp32 NCHW output convolution
    assert plan(4, 256, 200, 256, 3)[0] is False                    # Cin must be a multiple of 64


def test_conv_entry_points_refuse_tensors_beyond_32bit_offsets(lib):
    """The convolution kernels address outputs with 32-bit element offsets and operands through 32-bit buffer descriptors:
    a launch whose tensors exceed that must be refused (DDNM_E_SHAPE), not wrap around (checked before any launch, so
    this runs without a GPU)."""
    from ddnm_amd._lib import ConvDesc
    for fn in (lib.ddnm_conv2d_f32, lib.ddnm_conv3x3_s16_f32, lib.ddnm_conv3x3_f16_f32, lib.ddnm_conv_gather_s16_f32,
               lib.ddnm_conv1x1_f16_f32):
        d = ConvDesc()
        d.src0 = d.weight = d.out = d.gn_scale = d.gn_shift = 4096         # non-null dummies: never dereferenced
        d.B, d.Hin, d.Win, d.C0, d.Cout, d.ksize, d.stride, d.pad, d.Ho, d.Wo = 256, 256, 256, 128, 128, 3, 1, 1, 256, 256
        d.acc_scale = 1.0
        assert fn(ctypes.byref(d), None) == -2, fn.__name__                # 2^31 output elements exactly: refused


# ---- INTEGRATION.md section 2 is executable: the worked ctypes stub a maintainer would copy --------------------------------
def _integration_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def test_integration_md_stub_setup_block_runs_against_the_built_library(lib):
    """The first python block of INTEGRATION.md (struct mirror, version / size assertions, prototypes) is EXECUTED here
    against the library the build produced: a document that falls behind include/ddnm_hip.h (round 5: a 24-byte
    ddnm_step_scalars against the 48-byte ABI-6 struct) fails this test."""
    blocks = _integration_blocks()
    assert len(blocks) >= 2 and "class StepScalars" in blocks[0] and "ddnm_step_sr_avgpool_f32(" in blocks[1]
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)                         # the block loads "ddnm_amd/libddnm_hip.so" relative to the repository root
    try:
        exec(compile(blocks[0], "INTEGRATION.md[block 0]", "exec"), ns)
    finally:
        os.chdir(cwd)
    doc = ns["StepScalars"]
    from ddnm_amd._lib import PROTOTYPES, StepScalars
    assert ctypes.sizeof(doc) == ctypes.sizeof(StepScalars) == lib.ddnm_sizeof(3) == 48
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in StepScalars._fields_]
    # the prototype the document declares is the one the loader uses (the struct pointer aside: two mirror classes)
    restype, argtypes = PROTOTYPES["ddnm_step_sr_avgpool_f32"]
    fn = ns["lib"].ddnm_step_sr_avgpool_f32
    assert fn.restype is restype and len(fn.argtypes) == len(argtypes)
    assert [a for a in fn.argtypes if a is not ctypes.POINTER(doc)] == [a for a in argtypes if a is not ctypes.POINTER(StepScalars)]


@pytest.mark.gpu
def test_integration_md_stub_call_block_runs_on_the_gpu():
    """Both blocks executed as written, on device tensors; the result must be the oracle's step (svd_ddnm.py:57-65 with
    the average-pooling operator)."""
    import torch
    from oracle import operators
    blocks = _integration_blocks()
    torch.manual_seed(0)
    n, eta = 2, 0.85
    at, at_next = torch.tensor(0.37), torch.tensor(0.52)
    xt, et = torch.randn(n, 3, 256, 256, device="cuda"), torch.randn(n, 3, 256, 256, device="cuda")
    op = operators.SuperResolution(3, 256, 4)
    y = op.A(torch.rand(n, 3, 256, 256) * 2 - 1).cuda().contiguous()
    x0_t, xt_next = torch.empty_like(xt), torch.empty_like(xt)
    ns = dict(torch=torch, at=at, at_next=at_next, eta=eta, xt=xt, et=et, y=y, x0_t=x0_t, xt_next=xt_next, n=n)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        torch.manual_seed(7)
        exec(compile(blocks[0] + "\n" + blocks[1], "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    noise = ns["noise"]
    x0 = (xt.cpu() - et.cpu() * (1 - at).sqrt()) / at.sqrt()
    x0h = x0 - op.A_pinv(op.A(x0) - y.cpu().reshape(n, -1)).reshape(x0.shape)
    c1, c2 = (1 - at_next).sqrt() * eta, (1 - at_next).sqrt() * (1 - eta ** 2) ** 0.5
    want = at_next.sqrt() * x0h + c1 * noise.cpu() + c2 * et.cpu()
    assert ((x0_t.cpu() - x0).norm() / x0.norm()).item() < 1e-6
    assert ((xt_next.cpu() - want).norm() / want.norm()).item() < 2e-6
