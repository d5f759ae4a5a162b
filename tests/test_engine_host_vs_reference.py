"""Host-side set-up of the ENGINE operators (the matrices / tables their constructors compute before any kernel
runs) against the behaviour of the reference's operator classes, evaluated on CPU with plain torch: a second,
oracle-independent pin of the constructors.  Needs /root/reference (build container only); no GPU, no HIP library."""
import numpy as np
import pytest
import torch

from oracle import cases, ref_import
from tests.helpers import rel

needs_ref = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present on this box")
pytestmark = needs_ref


def _ref_and_engine():
    from ddnm_amd.functions import svd_operators as E
    return ref_import.load().svd_operators, E


def _planes(v, c, d):
    return v.reshape(v.shape[0], c, d, d)


@pytest.mark.parametrize("factor,d", [(4, 64), (2, 32), (8, 64)])
def test_srconv_matrices(factor, d):
    R, E = _ref_and_engine()
    k = E.bicubic_kernel(factor)
    ref = R.SRConv(k, 3, d, "cpu", stride=factor)
    eng = E.SRConv(k, 3, d, "cpu", stride=factor)
    x = cases.operator_input(d, 2)
    y_ref = ref.A(x)
    y_eng = (eng.Ae @ _planes(x, 3, d) @ eng.Ae.T).reshape(2, -1)            # what _A's two GEMMs compute
    assert rel(y_eng, y_ref) < 2e-6
    p_ref = ref.A_pinv(y_ref.clone())
    m = d // factor
    p_eng = (eng.Pe @ _planes(y_ref, 3, m) @ eng.Pe.T).reshape(2, -1)
    assert rel(p_eng, p_ref) < 2e-5
    assert torch.equal(eng.singulars().cpu(), ref.singulars())


@pytest.mark.parametrize("name", ["deblur_uni", "deblur_gauss", "deblur_aniso"])
def test_deblurring_tables(name):
    R, E = _ref_and_engine()
    d = 32
    cfg = cases.weights.celeba_config(resolution=d)
    eng = E.build_operator(name, 0, cfg, "cpu")
    if name == "deblur_uni":
        ref = R.Deblurring(torch.Tensor([1 / 9] * 9), 3, d, "cpu")
    elif name == "deblur_gauss":
        kk = E.gaussian_taps(10, 2)
        ref = R.Deblurring(kk / kk.sum(), 3, d, "cpu")
    else:
        k2, k1 = E.gaussian_taps(20, 4), E.gaussian_taps(1, 4)
        ref = R.Deblurring2D(k1 / k1.sum(), k2 / k2.sum(), 3, d, "cpu")
    x = cases.operator_input(d, 2)
    y_ref = ref.A(x)
    X = _planes(x, 3, d)
    spec = (eng.V1t @ X @ eng.V2t.T) * eng.G.reshape(1, 3, d, d)               # V1^T X V2, gains
    y_eng = (eng.U1 @ spec @ eng.U2.T).reshape(2, -1)
    assert rel(y_eng, y_ref) < 1e-5
    Y = _planes(y_ref, 3, d)
    p_eng = (eng.V1 @ ((eng.U1t @ Y @ eng.U2t.T) * eng.Ginv.reshape(1, 3, d, d)) @ eng.V2.T).reshape(2, -1)
    assert rel(p_eng, ref.A_pinv(y_ref.clone())) < 1e-4
    assert torch.equal(eng.singulars().cpu(), ref.singulars())


def test_cs_matrix():
    R, E = _ref_and_engine()
    from oracle import operators as O
    gauss = O.gauss_matrix(cases.SEED + 21)
    torch.manual_seed(cases.SEED + 21)
    ref = R.CS(3, 64, 0.25, "cpu")
    eng = E.CS(3, 64, 0.25, "cpu", gauss=gauss)
    x = cases.operator_input(64, 2)
    patches = x.reshape(2, 3, 2, 32, 2, 32).permute(0, 1, 2, 4, 3, 5).reshape(2 * 3 * 4, 1024)      # ddnm_patchify_f32
    y_eng = (patches @ eng.M.T).reshape(2, -1)
    assert rel(y_eng, ref.A(x)) < 2e-6
    assert torch.equal(eng.singulars().cpu(), ref.singulars())


@pytest.mark.parametrize("seed", [0, 1])
def test_inpainting_rank_table(seed):
    R, E = _ref_and_engine()
    d = 16
    g = torch.Generator().manual_seed(seed)
    mask = (torch.rand(d, d, generator=g) > 0.3).long()
    r = torch.nonzero(mask.reshape(-1) == 0).long().reshape(-1) * 3
    missing = torch.cat([r, r + 1, r + 2], dim=0)
    ref = R.Inpainting(3, d, missing, "cpu")
    eng = E.Inpainting(3, d, missing, "cpu")
    x = torch.randn(2, 3, d, d, generator=g)
    y_ref = ref.A(x)
    # what ddnm_op_inpaint_A_f32 does with the table: y[b, 3*rank[p] + c] = x[b, c, p] for kept pixels
    rank = eng.rank.long()
    y_eng = torch.zeros(2, 3 * eng.n_kept)
    kept = torch.nonzero(rank >= 0).reshape(-1)
    for c in range(3):
        y_eng[:, 3 * rank[kept] + c] = x.reshape(2, 3, -1)[:, c, kept]
    assert torch.equal(y_eng, y_ref)
    assert torch.equal(eng.kept_mask.cpu().reshape(d, d), mask.float())


def test_walsh_hadamard_mask():
    R, E = _ref_and_engine()
    d = 16
    perm = cases.wh_perm(d)
    ref = R.WalshHadamardCS(3, d, 4, perm, "cpu")
    eng = E.WalshHadamardCS(3, d, 4, perm, "cpu")
    x = cases.operator_input(d, 2)
    # A^+ A x through the engine's spectral mask: fwht, keep the measured (channel, frequency) entries, fwht
    def fwht2(v):                                             # orthonormal separable WHT of [B,C,d,d]
        H = torch.tensor([[1.0]])
        while H.shape[0] < d:
            H = torch.cat([torch.cat([H, H], 1), torch.cat([H, -H], 1)], 0)
        H = H / d ** 0.5
        return H @ v @ H.T
    spec = fwht2(_planes(x, 3, d)) * eng.mask.reshape(1, 3, d, d)
    got = fwht2(spec).reshape(2, -1)
    want = ref.A_pinv(ref.A(x)).reshape(2, -1)
    assert rel(got, want) < 1e-5
    assert eng.n_keep == ref.singulars().shape[0]
