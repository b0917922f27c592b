#!/usr/bin/env python
"""Training driver of the regression head on the HIP hot path.  Same flags, optimizer groups and checkpoint location as
the reference's train_regression.py (:13-39): Conv3 backbone, one Adam with lr 1e-3 for the GP hyper-parameters and the
backbone, `--stop_epoch` calls of `train_loop`, then `save_checkpoint(<save_dir>checkpoints/<dataset>/<model>_<method>)`.

  python train_regression.py --method DKT [--spectral] [--stop_epoch 100] [--seed 1]
"""
import os

import numpy as np
import torch

import dkt_amd
from dkt_amd import backbone, configs
from dkt_amd.data import SyntheticHeadPoseSampler
from dkt_amd.io_utils import parse_args_regression


def seed_everything(seed):
    np.random.seed(seed)
    torch.manual_seed(seed)


def regression_kernel(params):
    """`--spectral` wins; otherwise configs.kernel_type when it names a regression kernel, else rbf."""
    if params.spectral:
        return 'spectral'
    return configs.kernel_type if configs.kernel_type in ('rbf', 'RBF', 'spectral') else 'rbf'


def checkpoint_path(params):
    os.makedirs(os.path.join(configs.save_dir, 'checkpoints', params.dataset), exist_ok=True)
    return os.path.join(configs.save_dir, 'checkpoints', params.dataset, '%s_%s' % (params.model, params.method))


def build_model(params, sampler):
    if params.method != 'DKT':
        raise ValueError('Unrecognised method (only DKT is built)')
    if params.dataset != 'synthetic':
        raise NotImplementedError("dataset '%s' needs the reference's image tree and torchvision; use --dataset synthetic"
                                  % params.dataset)
    if params.model != 'Conv3':
        raise ValueError('the regression drivers use the Conv3 backbone')
    return dkt_amd.DKTRegression(backbone.Conv3(), regression_kernel(params), batch_fn=sampler).cuda()


def main(argv=None):
    params = parse_args_regression('train_regression', argv)
    seed_everything(params.seed)
    model = build_model(params, SyntheticHeadPoseSampler(seed=params.seed))
    groups = [{'params': part.parameters(), 'lr': 0.001} for part in (model.model, model.feature_extractor)]
    optimizer = torch.optim.Adam(groups)
    for epoch in range(params.stop_epoch):
        model.train_loop(epoch, optimizer)
    model.save_checkpoint(checkpoint_path(params))
    return model


if __name__ == '__main__':
    main()
