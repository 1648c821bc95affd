"""Accuracy metrics with the reference's definitions (utils/metrics.py:38-67)."""
from __future__ import annotations

import math

import numpy as np
import torch


def _std(Y_std):
    return float(torch.as_tensor(Y_std).reshape(-1)[0])


def rmse(Y_pred_mean, Y_test, Y_std):
    """utils/metrics.py:38-40."""
    d = Y_pred_mean.detach().to("cpu", torch.float64) - Y_test.detach().to("cpu", torch.float64)
    return _std(Y_std) * torch.sqrt(torch.mean(d ** 2))


def nlpd(Y_test_pred, Y_test, Y_std):
    """Joint predictive log-density / n_test, rescaled (utils/metrics.py:42-47; note: joint, not marginal)."""
    lpd = Y_test_pred.log_prob(Y_test)
    return -(lpd.detach() / len(Y_test) - math.log(_std(Y_std)))


def nlpd_marginal(Y_test_pred, Y_test, Y_std):
    """Mean of the per-point Gaussian log-densities (utils/metrics.py:49-58)."""
    y = Y_test.detach().to("cpu", torch.float64).numpy()
    mu = Y_test_pred.loc.detach().to("cpu", torch.float64).numpy()
    var = Y_test_pred.variance.detach().to("cpu", torch.float64).numpy()
    lp = -0.5 * np.log(2.0 * np.pi * var) - 0.5 * (y - mu) ** 2 / var - math.log(_std(Y_std))
    return float(-np.mean(lp))


def nlpd_mixture(Y_test_pred_list, Y_test, Y_std):
    """Mean over the per-sample joint nlpd (utils/metrics.py:61-67)."""
    return float(np.mean([float(nlpd(p, Y_test, Y_std)) for p in Y_test_pred_list]))
