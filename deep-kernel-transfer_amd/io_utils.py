"""Command-line flags and checkpoint-path helpers with the reference's names and defaults
(reference io_utils.py:17-47 parse_args, :66-86 get_assigned_file / get_resume_file / get_best_file).
Only the DKT method is built; `--dataset synthetic` (default here: no datasets exist in this environment)
draws class-structured random episodes, the dataset names of the reference are accepted when a
`filelists/<dataset>/{base,val,novel}.json` tree and torchvision are present."""
from __future__ import annotations

import argparse
import glob
import os

import numpy as np

from .backbone import model_dict  # noqa: F401  (re-exported like the reference's io_utils.model_dict)


def parse_args(script, argv=None):
    parser = argparse.ArgumentParser(description='few-shot script %s' % script)
    parser.add_argument('--seed', default=0, type=int, help='Seed for Numpy and pyTorch. Default: 0 (None)')
    parser.add_argument('--dataset', default='synthetic', help='synthetic/CUB/miniImagenet/cross/omniglot/cross_char')
    parser.add_argument('--model', default='Conv4', help='model: Conv{4|6} / Conv4S / ResNet{10|18|34}')
    parser.add_argument('--method', default='DKT', help='DKT (the other meta-learners of the reference are out of scope)')
    parser.add_argument('--train_n_way', default=5, type=int, help='class num to classify for training')
    parser.add_argument('--test_n_way', default=5, type=int, help='class num to classify for testing (validation)')
    parser.add_argument('--n_shot', default=5, type=int, help='number of labeled data in each class, same as n_support')
    parser.add_argument('--train_aug', action='store_true', help='perform data augmentation or not during training')
    parser.add_argument('--kernel_type', default=None, help='override configs.kernel_type')
    parser.add_argument('--image_size', default=None, type=int, help='override the backbone-dependent image size')
    parser.add_argument('--n_episode', default=None, type=int, help='episodes per epoch (train: 100) / per test run (600)')
    parser.add_argument('--meta_batch', default=1, type=int, help='[train, this build] episodes per Adam step through the batched hot path (1 = the reference: one step per episode)')
    if script == 'train':
        parser.add_argument('--num_classes', default=200, type=int, help='(baseline only; kept for CLI compatibility)')
        parser.add_argument('--save_freq', default=50, type=int, help='Save frequency')
        parser.add_argument('--start_epoch', default=0, type=int, help='Starting epoch')
        parser.add_argument('--stop_epoch', default=-1, type=int, help='Stopping epoch')
        parser.add_argument('--resume', action='store_true', help='continue from previous trained model with largest epoch')
        parser.add_argument('--warmup', action='store_true', help='continue from baseline (never used in the paper)')
    elif script == 'test':
        parser.add_argument('--split', default='novel', help='base/val/novel')
        parser.add_argument('--save_iter', default=-1, type=int, help='use the model trained in x epoch, best model if -1')
        parser.add_argument('--adaptation', action='store_true', help='further adaptation in test time or not')
        parser.add_argument('--repeat', default=5, type=int, help='Repeat the test N times with different seeds')
    else:
        raise ValueError('Unknown script')
    return parser.parse_args(argv)


def parse_args_regression(script, argv=None):
    """Flags of the reference's regression drivers (io_utils.py:48-63).  `--dataset synthetic` (default here) draws
    head-pose-like trajectories from `data.SyntheticHeadPoseSampler`; QMUL needs the image tree + torchvision.
    `--spectral` selects the SpectralMixture kernel (the reference parses the flag but only reads configs.kernel_type)."""
    parser = argparse.ArgumentParser(description='few-shot script %s' % script)
    parser.add_argument('--seed', default=0, type=int, help='Seed for Numpy and pyTorch. Default: 0 (None)')
    parser.add_argument('--model', default='Conv3', help='model: Conv{3}')
    parser.add_argument('--method', default='DKT', help='DKT (the feature-transfer baseline is out of scope)')
    parser.add_argument('--dataset', default='synthetic', help='synthetic / QMUL')
    parser.add_argument('--spectral', action='store_true', help='Use a spectral covariance kernel function')
    if script == 'train_regression':
        parser.add_argument('--start_epoch', default=0, type=int, help='Starting epoch')
        parser.add_argument('--stop_epoch', default=100, type=int, help='Stopping epoch')
        parser.add_argument('--resume', action='store_true', help='continue from previous trained model with largest epoch')
    elif script == 'test_regression':
        parser.add_argument('--n_support', default=5, type=int, help='Number of points on trajectory to be given as support points')
        parser.add_argument('--n_test_epochs', default=10, type=int, help='How many test people?')
    else:
        raise ValueError('Unknown script')
    return parser.parse_args(argv)


def get_assigned_file(checkpoint_dir, num):
    return os.path.join(checkpoint_dir, '{:d}.tar'.format(num))


def get_resume_file(checkpoint_dir):
    filelist = [x for x in glob.glob(os.path.join(checkpoint_dir, '*.tar')) if os.path.basename(x) != 'best_model.tar']
    if len(filelist) == 0:
        return None
    epochs = np.array([int(os.path.splitext(os.path.basename(x))[0]) for x in filelist])
    return os.path.join(checkpoint_dir, '{:d}.tar'.format(int(np.max(epochs))))


def get_best_file(checkpoint_dir):
    best_file = os.path.join(checkpoint_dir, 'best_model.tar')
    return best_file if os.path.isfile(best_file) else get_resume_file(checkpoint_dir)


def checkpoint_dir_for(params, save_dir):
    """reference train.py:178-182 / test.py:109-115"""
    d = '%s/checkpoints/%s/%s_%s' % (save_dir, params.dataset, params.model, params.method)
    if params.train_aug:
        d += '_aug'
    d += '_%dway_%dshot' % (params.train_n_way, params.n_shot)
    return d


def default_image_size(model, dataset):
    """reference train.py:84-90"""
    if 'Conv' in model:
        return 28 if dataset in ('omniglot', 'cross_char') or model == 'Conv4S' else 84
    return 224
