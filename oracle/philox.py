"""TEST INFRASTRUCTURE (oracle): numpy restatement of the engine's counter-based noise (ddnm_amd/csrc/philox.h).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123) -- the
generator family behind `torch.randn` on GPUs, which is what the reference's `torch.randn_like(x)` (functions/svd_ddnm.py:65,74)
calls -- followed by Box-Muller.  Pinned by the Random123 known-answer vectors (tests/test_philox.py).  Only tests import it.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: (k0, k1) -> uint32 array [..., 4]."""
    c = np.array(counter, dtype=np.uint32, copy=True)
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[..., 0].astype(np.uint64)
            p1 = M1 * c[..., 2].astype(np.uint64)
            h0, l0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            h1, l1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = np.stack([h1 ^ c[..., 1] ^ k0, l1, h0 ^ c[..., 3] ^ k1, l0], axis=-1)
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c


def normal4(seed_lo, seed_hi, elem4, iteration, image):
    """The four N(0, 1) values of counter (elem4, iteration, image, 0): float32 arithmetic like the kernel."""
    elem4 = np.asarray(elem4, dtype=np.uint32)
    ctr = np.stack([elem4, np.full_like(elem4, iteration), np.full_like(elem4, image), np.zeros_like(elem4)], -1)
    r = philox4x32_10(ctr, (seed_lo, seed_hi))
    s = np.float32(2.3283064365386963e-10)
    u = (r.astype(np.float32) + np.float32(0.5)) * s
    a0, a2 = np.minimum(u[..., 0], np.float32(0.99999994)), np.minimum(u[..., 2], np.float32(0.99999994))
    r0, r1 = np.sqrt(np.float32(-2.0) * np.log(a0)), np.sqrt(np.float32(-2.0) * np.log(a2))
    t0, t1 = np.float32(6.283185307179586) * u[..., 1], np.float32(6.283185307179586) * u[..., 3]
    return np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], -1).astype(np.float32)


def randn(seed, B, chw, iteration, image_base=0):
    """[B][chw] tensor as ddnm_randn_philox_f32 / the in-kernel draw produce it."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    return np.stack([normal4(lo, hi, np.arange(chw // 4), iteration, image_base + b).reshape(-1) for b in range(B)], 0)
