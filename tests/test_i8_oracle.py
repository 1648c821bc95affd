"""The integer-core contraction's arithmetic, restated on the CPU (oracle/i8_digits_oracle.py): digits reconstruct q exactly, the
magic-constant conversion is rint (ties included), the 28-pair sum equals the exact integer sum up to the stated truncation, the
int32 group sums cannot overflow at the split length, and Phi agrees with the fp64 oracle's K_uf K_fu."""
import math
import os
import sys
from fractions import Fraction

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import i8_digits_oracle as D  # noqa: E402
from oracle import vfe_oracle as O  # noqa: E402


def test_conversion_is_round_to_nearest_including_ties():
    rng = np.random.default_rng(0)
    K = np.concatenate([rng.random(20000), rng.random(20000) * 2.0 ** -rng.integers(0, 60, 20000),
                        [0.0, 1.0, 0.5, 2.0 ** -22, 3 * 2.0 ** -22, 5 * 2.0 ** -22, 1.0 + 2.0 ** -52, 2.0 ** -53, 2.0 ** -54, 3 * 2.0 ** -54,
                         2.0 ** -22 + 2.0 ** -53, 0.75 + 2.0 ** -22, 1 - 2.0 ** -53]])
    q = D.quantise(K)
    for x, got in zip(K.tolist(), q.tolist()):
        exact = Fraction(x) * 2 ** D.QBITS
        assert abs(Fraction(got) - exact) <= Fraction(1, 2), (x, got)          # nearest integer (either neighbour at a tie)
    assert int(D.quantise(np.array([1.0]))[0]) == 2 ** D.QBITS and int(D.quantise(np.array([0.0]))[0]) == 0
    assert int(D.quantise(np.array([np.nan]))[0]) == 0                          # a NaN kernel value has no digits (it travels through b)


def test_digits_reconstruct_q_and_are_balanced():
    rng = np.random.default_rng(1)
    q = np.concatenate([rng.integers(0, 2 ** D.QBITS, 50000), [0, 1, 2 ** 53, 2 ** 53 + 1, 2 ** 54, 2 ** 54 + 2, 127, 128, 255, 256, 2 ** 48 - 1,
                                                                2 ** 48]]).astype(np.int64)
    a = D.digits(q)
    rec = sum(a[p].astype(object) * (256 ** p) for p in range(D.NP_PLANES))
    assert all(int(r) == int(v) for r, v in zip(rec, q))
    assert all(x.dtype == np.int8 for x in a) and int(a[6].max()) <= 65 and int(a[6].min()) >= 0


def test_pair_sum_against_exact_integers_and_fp64_oracle():
    g = torch.Generator().manual_seed(3)
    N, M, d = 600, 7, 3
    X = torch.randn(N, d, dtype=torch.float64, generator=g)
    y = torch.randn(N, dtype=torch.float64, generator=g)
    Z = torch.randn(M, d, dtype=torch.float64, generator=g)
    Z[0] = X[0]
    ls = torch.tensor([1.2, 0.8, 1.5], dtype=torch.float64)
    st = O.suffstats(X, y, Z, ls, 1.0, 0)
    K = torch.exp(-0.5 * torch.cdist(X / ls, Z / ls) ** 2).numpy()
    q = D.quantise(K)
    Phi = D.phi_from_digits(D.digits(q))
    exact = D.phi_exact(q)
    assert np.abs(Phi - exact).max() < 6 * 2.0 ** -54 * math.sqrt(N) * 4          # the dropped pairs, zero-mean
    assert np.abs(Phi - exact).max() < 5e-16 * np.abs(exact).max()
    assert np.abs(Phi - st.Phi.numpy()).max() < 1e-13 * np.abs(st.Phi.numpy()).max()  # cdist vs the oracle's own distances


def test_group_sums_stay_inside_int32_at_the_split_length():
    # the adversarial planes: every digit at its largest magnitude, all products of one sign
    a = [np.full((D.SPLIT_ROWS, 2), -128, dtype=np.int8) for _ in range(6)] + [np.full((D.SPLIT_ROWS, 2), 65, dtype=np.int8)]
    D.phi_from_digits(a)  # asserts |group sum| < 2^31 inside
    worst = 5 * 128 * 128 * D.SPLIT_ROWS + 2 * 65 * 128 * D.SPLIT_ROWS
    assert worst < 2 ** 31 and 7 * 128 * 128 * D.SPLIT_ROWS < 2 ** 31
