"""Restated degradation operators in direct (matrix-free) math form
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Each class restates what the reference's SVD-factored class computes through
`A_functions.A` / `A_functions.A_pinv` (functions/svd_operators.py:52-58,68-80):
    A(x)      = U . (Sigma * V^T x[:, :n])
    A_pinv(y) = V . pad(Sigma^+ * U^T y)
Inputs are [B, ...]; outputs are [B, D] contiguous fp32, exactly like the
reference.  Pinned against the imported reference classes in
tests/test_oracle_pins.py and through tests/golden/operators_*.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F


class Denoising:
    """functions/svd_operators.py:442-462 -- identity."""

    def __init__(self, channels, img_dim):
        self.channels, self.img_dim = channels, img_dim

    def A(self, x):
        return x.reshape(x.shape[0], -1).clone()

    def A_pinv(self, y):
        return y.reshape(y.shape[0], -1).clone()


class SuperResolution:
    """functions/svd_operators.py:479-533: per r x r patch A = [1/r^2 ... 1/r^2]
    => A = average pooling, A^+ = patch replication (sigma = 1/r)."""

    def __init__(self, channels, img_dim, ratio):
        assert img_dim % ratio == 0
        self.channels, self.img_dim, self.ratio = channels, img_dim, ratio

    def A(self, x):
        x = x.reshape(x.shape[0], self.channels, self.img_dim, self.img_dim)
        return F.avg_pool2d(x, self.ratio).reshape(x.shape[0], -1)

    def A_pinv(self, y):
        s = self.img_dim // self.ratio
        y = y.reshape(y.shape[0], self.channels, s, s)
        up = y.repeat_interleave(self.ratio, dim=2).repeat_interleave(self.ratio, dim=3)
        return up.reshape(y.shape[0], -1)


class Colorization:
    """functions/svd_operators.py:627-667: per pixel y = w . rgb with
    w = (0.3333, 0.3334, 0.3333); A^+ y = w / |w|^2 * y."""

    W = (0.3333, 0.3334, 0.3333)

    def __init__(self, img_dim):
        self.channels, self.img_dim = 3, img_dim
        w = torch.tensor(self.W, dtype=torch.float32)
        self.w = w
        self.wp = w / (w * w).sum()

    def A(self, x):
        x = x.reshape(x.shape[0], 3, -1)
        return (x * self.w[None, :, None]).sum(1)

    def A_pinv(self, y):
        y = y.reshape(y.shape[0], 1, -1)
        return (y * self.wp[None, :, None]).reshape(y.shape[0], -1)


class Inpainting:
    """functions/svd_operators.py:324-359 with the index construction of
    guided_diffusion/diffusion.py:463-471: `missing` indexes the HWC-interleaved
    image; y keeps the remaining entries in ascending HWC index order."""

    def __init__(self, channels, img_dim, missing_indices):
        self.channels, self.img_dim = channels, img_dim
        n = channels * img_dim ** 2
        keep = torch.ones(n, dtype=torch.bool)
        keep[missing_indices.long()] = False
        self.kept = torch.nonzero(keep).reshape(-1)

    @staticmethod
    def missing_from_mask(mask2d):
        """diffusion.py:465-470 (mask==0 -> missing; r, g, b triplets concatenated)."""
        m = torch.as_tensor(mask2d).reshape(-1)
        r = torch.nonzero(m == 0).long().reshape(-1) * 3
        return torch.cat([r, r + 1, r + 2], dim=0)

    def A(self, x):
        b = x.shape[0]
        hwc = x.reshape(b, self.channels, -1).permute(0, 2, 1).reshape(b, -1)
        return hwc[:, self.kept].contiguous()

    def A_pinv(self, y):
        b = y.shape[0]
        hwc = torch.zeros(b, self.channels * self.img_dim ** 2, dtype=y.dtype)
        hwc[:, self.kept] = y.reshape(b, -1)
        return hwc.reshape(b, -1, self.channels).permute(0, 2, 1).reshape(b, -1)


def _fwht_last(a):
    """Unnormalised natural-order Walsh-Hadamard transform over the last axis of [B, C, N]."""
    b, c, n = a.shape
    h = 1
    while h < n:
        a = a.reshape(b, c, n // (2 * h), 2, h)
        lo, hi = a[..., 0, :], a[..., 1, :]
        a = torch.stack([lo + hi, lo - hi], dim=-2).reshape(b, c, n)
        h *= 2
    return a


class WalshHadamardCS:
    """functions/svd_operators.py:211-251: orthonormal FWHT over the 65536 pixels of
    each channel (16 butterfly stages, /img_dim), entries permuted by `perm`,
    laid out (k, c)-interleaved, first C*N/ratio kept."""

    def __init__(self, channels, img_dim, ratio, perm):
        self.channels, self.img_dim, self.ratio = channels, img_dim, ratio
        self.perm = perm.long()
        self.n_keep = channels * img_dim ** 2 // ratio

    def fwht(self, v):
        a = v.reshape(v.shape[0], self.channels, self.img_dim ** 2)
        return _fwht_last(a) / self.img_dim

    def A(self, x):
        b = x.shape[0]
        coef = self.fwht(x)[:, :, self.perm].permute(0, 2, 1).reshape(b, -1)
        return coef[:, :self.n_keep].contiguous()

    def A_pinv(self, y):
        b = y.shape[0]
        full = torch.zeros(b, self.channels * self.img_dim ** 2, dtype=y.dtype)
        full[:, :self.n_keep] = y.reshape(b, -1)
        tmp = torch.zeros(b, self.channels, self.img_dim ** 2, dtype=y.dtype)
        tmp[:, :, self.perm] = full.reshape(b, -1, self.channels).permute(0, 2, 1)
        return self.fwht(tmp).reshape(b, -1)


def bicubic_kernel(factor):
    """guided_diffusion/diffusion.py:485-499 (a = -0.5, 4*factor taps, normalised twice)."""
    def cubic(x, a=-0.5):
        ax = abs(x)
        if ax <= 1:
            return (a + 2) * ax ** 3 - (a + 3) * ax ** 2 + 1
        if 1 < ax < 2:
            return a * ax ** 3 - 5 * a * ax ** 2 + 8 * a * ax - 4 * a
        return 0.0
    k = np.zeros(factor * 4)
    for i in range(factor * 4):
        k[i] = cubic((1 / factor) * (i - np.floor(factor * 4 / 2) + 0.5))
    k = k / np.sum(k)
    k = torch.from_numpy(k).float()
    return k / k.sum()


def srconv_matrix(kernel, img_dim, stride):
    """1-D strided convolution matrix with reflective padding,
    functions/svd_operators.py:862-875."""
    small = img_dim // stride
    A = torch.zeros(small, img_dim)
    half = kernel.shape[0] // 2
    for i in range(stride // 2, img_dim + stride // 2, stride):
        for j in range(i - half, i + half):
            je = j
            if je < 0:
                je = -je - 1
            if je >= img_dim:
                je = (img_dim - 1) - (je - img_dim)
            A[i // stride, je] += kernel[j - i + half]
    return A


class SRConv:
    """functions/svd_operators.py:851-931: separable strided blur.  With
    A_small = U S V^T (singular values < 3e-2 zeroed, :878-879):
        A x   = Ae X Ae^T,  Ae = U S V[:, :m]^T          (per channel image X)
        A^+ y = Pe Y Pe^T,  Pe = V[:, :m] S^+ U^T
    y is NCHW-flat [B, C*m*m]."""

    ZERO = 3e-2

    def __init__(self, kernel, channels, img_dim, stride=1):
        self.channels, self.img_dim, self.ratio = channels, img_dim, stride
        self.small = img_dim // stride
        A_small = srconv_matrix(kernel.float().cpu(), img_dim, stride)
        U, S, V = torch.svd(A_small, some=False)
        S = S.clone()
        S[S < self.ZERO] = 0
        Sp = torch.where(S > 0, 1.0 / S, torch.zeros_like(S))
        m = self.small
        self.Ae = (U * S[None, :]) @ V[:, :m].T          # [m, img_dim]
        self.Pe = (V[:, :m] * Sp[None, :]) @ U.T         # [img_dim, m]
        self.singulars_small = S

    def A(self, x):
        b = x.shape[0]
        X = x.reshape(b * self.channels, self.img_dim, self.img_dim)
        return (self.Ae @ X @ self.Ae.T).reshape(b, -1)

    def A_pinv(self, y):
        b = y.shape[0]
        Y = y.reshape(b * self.channels, self.small, self.small)
        return (self.Pe @ Y @ self.Pe.T).reshape(b, -1)


# =================================================================================================
# DDNM+ (sigma_y > 0): Lambda / Lambda_noise restated (functions/svd_operators.py, per class)
#
#   lambda_i = s_i*sigma_t*sqrt(1-eta^2)/(a*sigma_y)  where sigma_t < a*sigma_y/s_i, else 1   (s_i = 0 -> 1)
#   (d1, d2)_i = (sigma_t*eta, 0)                              where sigma_t < a*sigma_y/s_i
#              = (sqrt(sigma_t^2 - a^2 sigma_y^2 / s_i^2), 0)  where sigma_t > a*sigma_y/s_i
#              = (sigma_t*eta, sigma_t*sqrt(1-eta^2))          where s_i = 0 (and in the measure-zero tie)
#   Lambda(v)          = V diag(lambda) V^T v
#   Lambda_noise(v, e) = V (d1 * v~ + d2 * e~)   with v~, e~ the RAW patch / needle / permuted entries of v, e
#                        (the reference does not apply V^T to them: svd_operators.py:573-623,697-736,281-320)
# =================================================================================================
def _coef(s, a, sigma_y, sigma_t, eta):
    """Per-singular-value (lambda, d1, d2) as python floats; `s` a float singular value (0 = null space)."""
    a, sigma_y, sigma_t = float(a), float(sigma_y), float(sigma_t)
    c1, c2 = sigma_t * eta, sigma_t * (1 - eta ** 2) ** 0.5
    if s == 0 or a == 0 or sigma_y == 0:
        return 1.0, c1, c2
    thr = a * sigma_y / s
    if sigma_t < thr:
        return s * sigma_t * (1 - eta ** 2) ** 0.5 / a / sigma_y, c1, 0.0
    if sigma_t > thr:
        return 1.0, (sigma_t ** 2 - a ** 2 * sigma_y ** 2 / s ** 2) ** 0.5, 0.0
    return 1.0, c1, c2


def _lambda_denoising(self, vec, a, sigma_y, sigma_t, eta):          # svd_operators.py:464-469
    if sigma_t < a * sigma_y:
        return vec * float(sigma_t * (1 - eta ** 2) ** 0.5 / a / sigma_y)
    return vec


def _lambda_noise_denoising(self, vec, a, sigma_y, sigma_t, eta, epsilon):   # :471-476 (epsilon unused)
    if sigma_t >= a * sigma_y:
        return vec * float((sigma_t ** 2 - a ** 2 * sigma_y ** 2) ** 0.5)
    return vec * sigma_t * eta


Denoising.Lambda, Denoising.Lambda_noise = _lambda_denoising, _lambda_noise_denoising


def _small_svd(row):
    """V of the 1 x n measurement row (torch.svd on CPU, some=False) -- svd_operators.py:487,633."""
    U, S, V = torch.svd(torch.tensor([row], dtype=torch.float32), some=False)
    return float(S[0]), V


def _sr_patches(self, v):
    b = v.shape[0]
    r = self.ratio
    p = v.reshape(b, self.channels, self.img_dim, self.img_dim).unfold(2, r, r).unfold(3, r, r)
    return p.contiguous().reshape(b, self.channels, -1, r * r)


def _sr_unpatch(self, p, b):
    r, yd = self.ratio, self.img_dim // self.ratio
    p = p.reshape(b, self.channels, yd, yd, r, r).permute(0, 1, 2, 4, 3, 5).contiguous()
    return p.reshape(b, self.channels * self.img_dim ** 2)


def _sr_lambda(self, vec, a, sigma_y, sigma_t, eta):                  # :535-571
    s, V = _small_svd([1 / self.ratio ** 2] * self.ratio ** 2)
    lam = torch.ones(self.ratio ** 2)
    lam[0] = _coef(s, a, sigma_y, sigma_t, eta)[0]
    p = _sr_patches(self, vec)
    p = (p @ V) * lam                         # V^T applied to every patch (row-vector form), then lambda
    return _sr_unpatch(self, p @ V.T, vec.shape[0])


def _sr_lambda_noise(self, vec, a, sigma_y, sigma_t, eta, epsilon):   # :573-623
    s, V = _small_svd([1 / self.ratio ** 2] * self.ratio ** 2)
    n = self.ratio ** 2
    _, d1m, d2m = _coef(s, a, sigma_y, sigma_t, eta)
    _, d1n, d2n = _coef(0.0, a, sigma_y, sigma_t, eta)
    d1, d2 = torch.full((n,), d1n), torch.full((n,), d2n)
    d1[0], d2[0] = d1m, d2m
    pv, pe = _sr_patches(self, vec) * d1, _sr_patches(self, epsilon) * d2
    return _sr_unpatch(self, pv @ V.T, vec.shape[0]) + _sr_unpatch(self, pe @ V.T, vec.shape[0])


SuperResolution.Lambda, SuperResolution.Lambda_noise = _sr_lambda, _sr_lambda_noise


def _color_lambda(self, vec, a, sigma_y, sigma_t, eta):               # :669-695
    s, V = _small_svd(list(self.W))
    lam = torch.ones(3)
    lam[0] = _coef(s, a, sigma_y, sigma_t, eta)[0]
    needles = vec.reshape(vec.shape[0], 3, -1).permute(0, 2, 1)      # B, HW, 3
    out = ((needles @ V) * lam) @ V.T
    return out.permute(0, 2, 1).reshape(vec.shape[0], -1)


def _color_lambda_noise(self, vec, a, sigma_y, sigma_t, eta, epsilon):   # :697-736
    s, V = _small_svd(list(self.W))
    _, d1m, d2m = _coef(s, a, sigma_y, sigma_t, eta)
    _, d1n, d2n = _coef(0.0, a, sigma_y, sigma_t, eta)
    d1, d2 = torch.tensor([d1m, d1n, d1n]), torch.tensor([d2m, d2n, d2n])
    nv = vec.reshape(vec.shape[0], 3, -1).permute(0, 2, 1) * d1
    ne = epsilon.reshape(vec.shape[0], 3, -1).permute(0, 2, 1) * d2
    out = nv @ V.T + ne @ V.T
    return out.permute(0, 2, 1).reshape(vec.shape[0], -1)


Colorization.Lambda, Colorization.Lambda_noise = _color_lambda, _color_lambda_noise


def _inp_mask(self, b, like):
    m = torch.zeros(self.channels * self.img_dim ** 2, dtype=torch.bool)
    m[self.kept] = True
    return m.reshape(-1, self.channels).permute(1, 0).reshape(1, -1).expand(b, -1)      # CHW-flat mask of kept entries


def _inp_lambda(self, vec, a, sigma_y, sigma_t, eta):                 # :361-387
    lam = _coef(1.0, a, sigma_y, sigma_t, eta)[0]
    v = vec.reshape(vec.shape[0], -1)
    return torch.where(_inp_mask(self, v.shape[0], v), v * lam, v)


def _inp_lambda_noise(self, vec, a, sigma_y, sigma_t, eta, epsilon):  # :389-439
    _, d1m, d2m = _coef(1.0, a, sigma_y, sigma_t, eta)
    _, d1n, d2n = _coef(0.0, a, sigma_y, sigma_t, eta)
    v, e = vec.reshape(vec.shape[0], -1), epsilon.reshape(vec.shape[0], -1)
    return torch.where(_inp_mask(self, v.shape[0], v), v * d1m + e * d2m, v * d1n + e * d2n)


Inpainting.Lambda, Inpainting.Lambda_noise = _inp_lambda, _inp_lambda_noise


def _wh_measured(self):
    """[C][N] bool: spectral entry (c, q) is measured iff (invperm[q]*C + c) < n_keep."""
    k = torch.arange(self.img_dim ** 2)
    m = torch.zeros(self.channels, self.img_dim ** 2, dtype=torch.bool)
    for c in range(self.channels):
        m[c, self.perm] = (k * self.channels + c) < self.n_keep
    return m


def _wh_lambda(self, vec, a, sigma_y, sigma_t, eta):                  # :253-279
    lam = _coef(1.0, a, sigma_y, sigma_t, eta)[0]
    m = _wh_measured(self)
    coef = self.fwht(vec)
    coef = torch.where(m[None], coef * lam, coef)
    return self.fwht(coef).reshape(vec.shape[0], -1)


def _wh_lambda_noise(self, vec, a, sigma_y, sigma_t, eta, epsilon):   # :281-320 (no forward transform of vec / epsilon)
    _, d1m, d2m = _coef(1.0, a, sigma_y, sigma_t, eta)
    _, d1n, d2n = _coef(0.0, a, sigma_y, sigma_t, eta)
    m = _wh_measured(self)[None]
    v = vec.reshape(vec.shape[0], self.channels, -1)
    e = epsilon.reshape(vec.shape[0], self.channels, -1)
    tv = torch.where(m, v * d1m, v * d1n)
    te = torch.where(m, e * d2m, e * d2n)
    return (self.fwht(tv) + self.fwht(te)).reshape(vec.shape[0], -1)


WalshHadamardCS.Lambda, WalshHadamardCS.Lambda_noise = _wh_lambda, _wh_lambda_noise


# =================================================================================================
# Separable blur operators (functions/svd_operators.py:934-1091 Deblurring, :1094-1165 Deblurring2D)
# =================================================================================================
def blur_matrix(kernel, img_dim):
    """1-D convolution matrix with zero boundary, svd_operators.py:947-951."""
    A = torch.zeros(img_dim, img_dim)
    half = kernel.shape[0] // 2
    for i in range(img_dim):
        for j in range(i - half, i + half):
            if j < 0 or j >= img_dim:
                continue
            A[i, j] = kernel[j - i + half]
    return A


class Deblurring2D:
    """A = U diag(g) V^T with U = U1 (x) U2, V = V1 (x) V2 (row/column blur), singular values s1_i*s2_j
    (1-D values below 3e-2 zeroed), sorted descending.  REFERENCE QUIRK reproduced: `singulars()` tiles the
    sorted values 3x (`repeat(1, 3)`, :1012,1154) while the spectral vector is (position, channel)-interleaved,
    so entry (k, c) is scaled by s_sorted[(3k + c) mod N^2]; `A_pinv` uses the same tiling (:1014-1023)."""

    ZERO = 3e-2

    def __init__(self, kernel1, kernel2, channels, img_dim):
        self.channels, self.img_dim = channels, img_dim
        n = img_dim
        U1, S1, V1 = torch.svd(blur_matrix(kernel1.float().cpu(), n), some=False)
        U2, S2, V2 = torch.svd(blur_matrix(kernel2.float().cpu(), n), some=False)
        S1, S2 = S1.clone(), S2.clone()
        S1[S1 < self.ZERO] = 0
        S2[S2 < self.ZERO] = 0
        big = torch.matmul(S1.reshape(n, 1), S2.reshape(1, n)).reshape(n * n)
        s_sorted, perm = big.sort(descending=True)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n * n)
        idx = (3 * inv[None, :] + torch.arange(channels)[:, None]) % (n * n)          # [C][N^2]
        self.G = s_sorted[idx]                                                            # gain of spectral entry (c, q)
        self.Ginv = torch.where(self.G > 0, 1.0 / self.G, torch.zeros_like(self.G))
        self.U1, self.V1, self.U2, self.V2 = U1, V1, U2, V2
        self._singulars = s_sorted

    def _planes(self, v):
        return v.reshape(v.shape[0], self.channels, self.img_dim, self.img_dim)

    def A(self, x):
        b = x.shape[0]
        T = self.V1.T @ self._planes(x) @ self.V2
        T = T * self.G.reshape(1, self.channels, self.img_dim, self.img_dim)
        return (self.U1 @ T @ self.U2.T).reshape(b, -1)

    def A_pinv(self, y):
        b = y.shape[0]
        T = self.U1.T @ self._planes(y) @ self.U2
        T = T * self.Ginv.reshape(1, self.channels, self.img_dim, self.img_dim)
        return (self.V1 @ T @ self.V2.T).reshape(b, -1)


class Deblurring(Deblurring2D):
    def __init__(self, kernel, channels, img_dim):
        super().__init__(kernel, kernel, channels, img_dim)
        _, S, _ = torch.svd(blur_matrix(kernel.float().cpu(), img_dim), some=False)
        # `_singulars_orig` (:959,963): un-thresholded products, carried through the same permutation as the sorted
        # values -- in spectral-plane order the permutation drops out
        self.S_orig = torch.matmul(S.reshape(-1, 1), S.reshape(1, -1))

    def _weights(self, a, sigma_y, sigma_t, eta):
        """lambda, d1, d2 per spectral entry, the tensor expressions of :1021-1031 and :1049-1072."""
        s = self.S_orig
        inv = torch.where(s == 0, torch.zeros_like(s), 1.0 / s)
        lam = torch.ones_like(s)
        d1 = torch.ones_like(s) * sigma_t * eta
        d2 = torch.ones_like(s) * sigma_t * (1 - eta ** 2) ** 0.5
        if a != 0 and sigma_y != 0:
            c = (sigma_t < a * sigma_y * inv) * 1.0
            lam = lam * (1 - c) + c * (s * sigma_t * (1 - eta ** 2) ** 0.5 / a / sigma_y)
            d1 = d1 * (1 - c) + c * sigma_t * eta
            d2 = d2 * (1 - c)
            c = (sigma_t > a * sigma_y * inv) * 1.0
            d1 = d1 * (1 - c) + torch.sqrt(c * (sigma_t ** 2 - a ** 2 * sigma_y ** 2 * inv ** 2))
            d2 = d2 * (1 - c)
            c = (s == 0) * 1.0
            d1 = d1 * (1 - c) + c * sigma_t * eta
            d2 = d2 * (1 - c) + c * sigma_t * (1 - eta ** 2) ** 0.5
        return lam, d1, d2

    def Lambda(self, vec, a, sigma_y, sigma_t, eta):                       # :1016-1040
        lam = self._weights(a, sigma_y, sigma_t, eta)[0]
        T = self.V1.T @ self._planes(vec) @ self.V1
        return (self.V1 @ (T * lam) @ self.V1.T).reshape(vec.shape[0], -1)

    def Lambda_noise(self, vec, a, sigma_y, sigma_t, eta, epsilon):        # :1042-1091: vec / epsilon enter V raw
        _, d1, d2 = self._weights(a, sigma_y, sigma_t, eta)
        M = self._planes(vec) * d1 + self._planes(epsilon) * d2
        return (self.V1 @ M @ self.V1.T).reshape(vec.shape[0], -1)


def gaussian_taps(sigma, radius):
    """diffusion.py:507-520: exp(-0.5 (x/sigma)^2) evaluated in fp32, x = -radius..radius."""
    return torch.tensor([float(torch.exp(torch.Tensor([-0.5 * (x / sigma) ** 2]))) for x in range(-radius, radius + 1)])


# =================================================================================================
# Block-based compressed sensing (functions/svd_operators.py:101-159)
# =================================================================================================
def gauss_matrix(seed):
    """The 1024x1024 Gaussian matrix of svd_operators.py:107 as drawn after torch.manual_seed(seed)."""
    return torch.randn(32 ** 2, 32 ** 2, generator=torch.Generator().manual_seed(seed))


class CS:
    """Every 32x32 patch p of every channel is expanded in the right-singular basis of one Gaussian matrix,
    c = Vt_small p (:134-138); the spectral vector lists the first `cs` coefficients of all (channel, patch)
    pairs, then the rest (:140-145).  All singular values are 1 (:110), so A keeps the first block and
    A^+ = V(pad(.)) puts it back (:52-58,68-80)."""

    def __init__(self, channels, img_dim, ratio, gauss):
        self.channels, self.img_dim, self.n = channels, img_dim, img_dim // 32
        _, _, V = torch.svd(gauss.float(), some=False)                       # :108
        self.cs = int(32 * 32 * ratio)                                       # :110-111
        self.Vt_cs = V.T[:self.cs].contiguous()

    def A(self, x):
        b, c, n = x.shape[0], self.channels, self.n
        patches = x.reshape(b, c, n, 32, n, 32).permute(0, 1, 2, 4, 3, 5).reshape(b, c, n * n, 1024)
        return (patches @ self.Vt_cs.T).reshape(b, -1)

    def A_pinv(self, y):
        b, c, n = y.shape[0], self.channels, self.n
        patches = y.reshape(b, c, n * n, self.cs) @ self.Vt_cs
        return patches.reshape(b, c, n, n, 32, 32).permute(0, 1, 2, 4, 3, 5).reshape(b, -1)


# =================================================================================================
# Composed degradation of the simplified path (guided_diffusion/diffusion.py:260-290)
# =================================================================================================
def color2gray(x):
    """guided_diffusion/diffusion.py:33-36."""
    g = x[:, 0] * (1 / 3) + x[:, 1] * (1 / 3) + x[:, 2] * (1 / 3)
    return g[:, None].repeat(1, 3, 1, 1)


def gray2color(x):
    """guided_diffusion/diffusion.py:38-42."""
    coef = 1 / 3
    base = 3 * coef ** 2
    return torch.stack([x[:, 0] * coef / base] * 3, 1)


def mask_color_sr(mask, scale, img_dim):
    """(A, Ap) of `--deg mask_color_sr` / `diy`: A = pool(gray(mask * z)), Ap = mask * color(upsample(z))."""
    pool = torch.nn.AdaptiveAvgPool2d((img_dim // scale, img_dim // scale))

    def up(z):
        return z.repeat_interleave(scale, dim=2).repeat_interleave(scale, dim=3)      # MeanUpsample, :27-31

    return (lambda z: pool(color2gray(z * mask))), (lambda z: gray2color(up(z)) * mask)
