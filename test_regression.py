#!/usr/bin/env python
"""Evaluation driver of the regression head on the HIP hot path.  Same flags, checkpoint location and report lines as
the reference's test_regression.py (:12-39): load `<save_dir>checkpoints/<dataset>/<model>_<method>`, draw
`--n_test_epochs` test tasks, condition each on `--n_support` random frames and report the mean / std of the MSE.

  python test_regression.py --method DKT [--spectral] [--n_support 5] [--n_test_epochs 10]
"""
import numpy as np
import torch

import dkt_amd
from dkt_amd.data import SyntheticHeadPoseSampler
from dkt_amd.io_utils import parse_args_regression
from train_regression import build_model, checkpoint_path, seed_everything


def evaluate(model, n_support, n_tasks):
    """One `test_loop` call per task; the reference passes optimizer=None for DKT (test_regression.py:22-24)."""
    return [model.test_loop(n_support, None).item() for _ in range(n_tasks)]


def main(argv=None):
    params = parse_args_regression('test_regression', argv)
    seed_everything(params.seed)
    model = build_model(params, SyntheticHeadPoseSampler(seed=params.seed + 1))
    model.load_checkpoint(checkpoint_path(params))
    mse = evaluate(model, params.n_support, params.n_test_epochs)
    bar = "-" * 19
    print("%s\nAverage MSE: %s +- %s\n%s" % (bar, str(np.mean(mse)), str(np.std(mse)), bar))
    return mse


if __name__ == '__main__':
    main()
