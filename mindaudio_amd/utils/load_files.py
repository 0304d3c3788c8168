"""Mirror of mindaudio/utils/load_files.py:9-36: global CMVN statistics file -> (mean, inverse standard deviation)."""
import json
import math

import numpy as np


def load_cmvn(cmvn_file, is_json=True):
    """json {"mean_stat": [...], "var_stat": [...], "frame_num": n} (sums over all frames) -> (mean, istd) float64 arrays;
    the variance is floored at 1e-20 before the inverse square root (load_files.py:24-28)."""
    if not is_json:
        raise NotImplementedError("only the json statistics format of compute_cmvn_stats.py is read")
    with open(cmvn_file) as fh:
        stats = json.load(fh)
    count = stats["frame_num"]
    mean = np.asarray(stats["mean_stat"], dtype=np.float64) / count
    var = np.asarray(stats["var_stat"], dtype=np.float64) / count - mean * mean
    var = np.maximum(var, 1.0e-20)
    return mean, 1.0 / np.sqrt(var)
