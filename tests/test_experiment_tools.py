"""Result records: the reference's JSON schema and file naming (experiments/regression.py:157-199, utils/experiment_tools.py)."""
import json

import ggp_amd
from ggp_amd import experiment_tools as ET


def test_result_record_has_the_reference_schema(tmp_path):
    rec = ET.result_record("Elevator", "Bayesian_SGPR_HMC", 0.123456, 0.654321, 12.5, perf_times=[1.0, 2.0], step_sizes=[0.1],
                           split_index=2, num_inducing=100, max_iter=1000, date_str="Oct_03", leapfrogs_per_s=11000.0)
    keys = list(rec)
    assert keys[:10] == list(ET.EXP_INFO_KEYS) and keys[10:14] == list(ET.METRIC_KEYS)   # {**exp_info, **metrics}, then extras
    assert rec["test_rmse"] == 0.1235 and rec["test_nlpd"] == 0.6543                      # np.round(..., 4) in the reference
    fn = ET.save_record(rec, str(tmp_path))
    assert fn.endswith("Oct_03/Oct_03_dataset-Elevator_model_name-Bayesian_SGPR_HMC_split-2_frac-0.9_num_inducing-100_max_iter-1000__.json")
    assert json.load(open(fn))["perf_times"] == [1.0, 2.0]
    assert ET.experiment_name("d", "x", "SVGP", 0, 0.9, 50, None, 25, 100).endswith("num_inducing-50_num_epochs-25_batch_size-100")
