"""Result records with the reference's JSON schema (experiments/regression.py:157-199) and file naming
(utils/experiment_tools.py:11-61), so the reference's aggregation scripts (experiments/aggregate_results.py) read the
output of the drivers in ``experiments/`` unchanged: one flat dict ``{**exp_info, **metrics}``."""
from __future__ import annotations

import json
import os
from datetime import datetime
from typing import Optional, Sequence

EXP_INFO_KEYS = ("date_str", "split_index", "dataset_name", "model_name", "num_inducing", "max_iter", "num_epochs", "batch_size",
                 "train_test_split", "step_sizes")
METRIC_KEYS = ("test_rmse", "test_nlpd", "wall_clock_secs", "perf_times")


def experiment_name(date_str, dataset_name, model_name, split_index, train_test_split, num_inducing=None, max_iter=None,
                    num_epochs=None, batch_size=None, step_sizes=None, num_samples=None) -> str:
    """``<date>_dataset-<name>_model_name-<model>_split-<i>_frac-<f>_...`` -- the fields the reference appends per model."""
    parts = [("dataset", dataset_name), ("model_name", model_name), ("split", split_index), ("frac", train_test_split)]
    if model_name in ("SGPR", "Bayesian_SGPR_HMC"):
        parts += [("num_inducing", num_inducing), ("max_iter", max_iter)]
    elif model_name in ("SVGP", "Bayesian_SVGP"):
        parts += [("num_inducing", num_inducing), ("num_epochs", num_epochs), ("batch_size", batch_size)]
    else:
        parts += [("max_iter", max_iter)]
    return str(date_str) + "".join("_%s-%s" % kv for kv in parts)


def result_record(dataset_name: str, model_name: str, test_rmse: float, test_nlpd: float, wall_clock_secs: float,
                  perf_times: Optional[Sequence[float]] = None, step_sizes: Optional[Sequence[float]] = None, split_index: int = 0,
                  num_inducing: Optional[int] = None, max_iter: Optional[int] = None, num_epochs: Optional[int] = None,
                  batch_size: Optional[int] = None, train_test_split: float = 0.9, date_str: Optional[str] = None, **extra) -> dict:
    """The reference's ``experiment_dict``; ``extra`` keys (throughput, sizes ...) are appended after the reference's."""
    rec = {"date_str": date_str or datetime.now().strftime("%b_%d"), "split_index": int(split_index), "dataset_name": dataset_name,
           "model_name": model_name, "num_inducing": num_inducing, "max_iter": max_iter, "num_epochs": num_epochs,
           "batch_size": batch_size, "train_test_split": train_test_split,
           "step_sizes": [float(v) for v in step_sizes] if step_sizes is not None else None,
           "test_rmse": round(float(test_rmse), 4), "test_nlpd": round(float(test_nlpd), 4),
           "wall_clock_secs": float(wall_clock_secs), "perf_times": [float(v) for v in perf_times] if perf_times is not None else None}
    rec.update(extra)
    return rec


def save_record(rec: dict, log_dir: str) -> str:
    """``<log_dir>/<date>/<experiment_name>__.json`` like the reference."""
    info = {k: rec.get(k) for k in EXP_INFO_KEYS}
    path = os.path.join(log_dir, str(rec["date_str"]))
    os.makedirs(path, exist_ok=True)
    fn = os.path.join(path, experiment_name(**info) + "__.json")
    with open(fn, "w") as fp:
        json.dump(rec, fp, indent=4)
    return fn
