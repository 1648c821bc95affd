"""CPU restatement of the integer-core contraction's arithmetic -- TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module; the product path (``csrc/sgp_suffstats_i8.hip``) never does.  It restates, in numpy
integers, what ``kfu_digits_kernel`` and ``i8_syrk_tile_kernel`` compute, so that the digit arithmetic has an oracle of its
own next to the fp64 one (``vfe_oracle.suffstats``):

    q        = rint(K' 2^54)                                        K' in [0, 1] (a kernel profile without its amplitude)
    a_p      = bytes of (q + C) ^ C as int8, C = 0x80 per byte      balanced digits: q = sum_p a_p 256^p, a_p in [-128, 127]
    Phi_IJ   = 2^-108 sum_{p + r >= 6} 256^(p + r) sum_n a_p[n, I] a_r[n, J]
             = 2^-108 sum_n q[n, I] q[n, J]  -  (the 21 dropped digit pairs, < 6 x 2^-54 per product)

There is no reference code to follow: the reference's contraction is ``torch.matmul`` inside gpytorch's
``InducingPointKernel`` (models/sgpr.py:37), a plain fp64 GEMM.  ``tests/test_i8_oracle.py`` pins this restatement against exact
Python-integer arithmetic and against ``vfe_oracle.suffstats``; ``tests/test_int8_contraction.py`` holds the HIP path against both.
"""
import numpy as np

NP_PLANES = 7
QBITS = 54          # fixed-point scale of q (round 3: 53; seven balanced digits hold 2^55, K' <= 1 + ulp needs 2^54 + 2)
C_BIAS = 0x0080808080808080
SPLIT_ROWS = 16384  # rows a split may span: 7 pairs x 2^14 x 16384 < 2^31


def quantise(K):
    """q = rint(K 2^54) as int64, the way kfu_digits_kernel forms it: hi = rint(K 2^22) and the signed remainder
    r = rint(K 2^54 - hi 2^32), each read off the mantissa of a magic-constant sum; q = (hi - bit 32 of r's field) 2^32 + low 32 bits."""
    K = np.asarray(K, dtype=np.float64)
    th = K * 2.0 ** (QBITS - 32) + 2.0 ** 52  # one rounding (the product is exact): an fma on the device
    hf = th - 2.0 ** 52
    tl = (K * 2.0 ** QBITS - hf * 2.0 ** 32) + 1.5 * 2.0 ** 52   # the difference is exact
    tb = tl.view(np.uint64)
    q_hi = (th.view(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64) - ((tb >> np.uint64(32)) & np.uint64(1)).astype(np.int64)
    return (q_hi << 32) | (tb & np.uint64(0xFFFFFFFF)).astype(np.int64)


def digits(q):
    """Seven balanced digit planes a[p] (int8) of q >= 0: the bytes of (q + C) ^ C."""
    qq = (np.asarray(q, dtype=np.int64).astype(np.uint64) + np.uint64(C_BIAS)) ^ np.uint64(C_BIAS)
    assert not np.any(qq >> np.uint64(56)), "q does not fit seven digits"
    return [((qq >> np.uint64(8 * p)) & np.uint64(0xFF)).astype(np.uint8).view(np.int8) for p in range(NP_PLANES)]


def phi_from_digits(a):
    """Phi (M x M, without sf2^2) from the digit planes a[p][N, M]: the 28 pairs p + r >= 6, int32 group sums per split of
    at most SPLIT_ROWS rows (checked), folded to fp64 lowest significance first as the kernel does."""
    N, M = a[0].shape
    Phi = np.zeros((M, M))
    for n0 in range(0, N, SPLIT_ROWS):
        blk = [x[n0:n0 + SPLIT_ROWS].astype(np.int64) for x in a]
        v = np.zeros((M, M))
        for g in range(NP_PLANES):                      # group g = p + r - 6
            acc = np.zeros((M, M), dtype=np.int64)
            for p in range(g, NP_PLANES):               # p + r = g + 6, r = g + 6 - p <= 6
                acc += blk[p].T @ blk[g + 6 - p]
            assert np.abs(acc).max(initial=0) < 2 ** 31, "int32 group sum would overflow"
            v = acc.astype(np.float64) * 2.0 ** (8 * g + 48 - 2 * QBITS) + v
        Phi += v
    return Phi


def phi_exact(q):
    """sum_n q[n, I] q[n, J] / 2^108 in exact integer arithmetic (Python ints), one rounding at the end."""
    q = np.asarray(q, dtype=object)
    S = q.T.dot(q)
    return np.array([[int(S[i, j]) / 2.0 ** (2 * QBITS) for j in range(S.shape[1])] for i in range(S.shape[0])])
