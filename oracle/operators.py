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
