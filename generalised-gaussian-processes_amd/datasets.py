"""Readers for the on-disk formats the reference's experiments consume.  The files themselves
are not shipped (no network); tests write small stand-ins in the same formats.

* ``mauna.txt`` -- whitespace table ``year co2`` with -99.99 marking missing months
  (experiments/co2_bayesian_sgpr_hmc.py:23-52): ``load_co2_dataset`` reproduces its normalisation and split.
* ``elevators.mat`` -- MATLAB file with one ``data`` matrix, last column = target (utils/dataset.py:257-261), followed by
  the reference's shuffle / split / standardisation (utils/dataset.py:38-41,49-71).
"""
from __future__ import annotations

import numpy as np

CO2_SPLIT_INDEX = {1990: 394, 1995: 454, 2000: 514, 2005: 574, 2010: 634}
BASE_SEED = 0


def read_mauna_txt(path):
    """(year, co2) arrays with the -99.99 rows dropped."""
    rows = []
    with open(path) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) < 2 or parts[0].startswith("#"):
                continue
            yr, co2 = float(parts[0]), float(parts[1])
            if co2 == -99.99:
                continue
            rows.append((yr, co2))
    a = np.asarray(rows, dtype=np.float64)
    return a[:, 0], a[:, 1]


def load_co2_dataset(path, year_split=2010):
    """y = (co2 - co2[0]) / std(co2), t = year - year[0]; train = first CO2_SPLIT_INDEX[year_split] months, test = the
    next 60.  Returns (y_train, t_train[:, None], y_test, t_test[:, None], std_co2)."""
    year, co2 = read_mauna_txt(path)
    std = float(np.std(co2))
    y = (co2 - co2[0]) / std
    t = year - year[0]
    sep = CO2_SPLIT_INDEX[year_split]
    return y[:sep], t[:sep, None], y[sep:sep + 60], t[sep:sep + 60, None], std


def normalize(X):
    """utils/dataset.py:38-41."""
    mean = np.average(X, 0)[None, :]
    std = 1e-6 + np.std(X, 0)[None, :]
    return (X - mean) / std, mean, std


def read_elevators_mat(path):
    from scipy.io import loadmat
    data = np.asarray(loadmat(path)["data"], dtype=np.float64)
    return data[:, :-1], data[:, -1].reshape(-1, 1)


def split_dataset(X, Y, split=0, prop=0.9, base_seed=BASE_SEED):
    """The reference's protocol: standardise X and Y over the whole set, shuffle with seed base_seed + split, first
    ``prop`` of the rows train (utils/dataset.py:49-71).  Returns X_train, y_train, X_test, y_test (float64)."""
    Xn, _, _ = normalize(np.asarray(X, dtype=np.float64))
    Yn, _, _ = normalize(np.asarray(Y, dtype=np.float64).reshape(len(X), -1))
    ind = np.arange(len(Xn))
    rng = np.random.RandomState(base_seed + split)
    rng.shuffle(ind)
    n = int(len(Xn) * prop)
    return Xn[ind[:n]], Yn[ind[:n]].ravel(), Xn[ind[n:]], Yn[ind[n:]].ravel()
