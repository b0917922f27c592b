#!/usr/bin/env python
"""DKT evaluation driver with the reference's protocol and output (reference test.py:62-199): 600 episodes,
n_query = 15, 95% CI = 1.96 sigma / sqrt(600), repeated over `--repeat` seeds, one line appended to
./record/results.txt in the reference's format."""
import os
import time

import numpy as np
import torch

import dkt_amd
from dkt_amd import configs, distributed
from dkt_amd.data import get_episode_loader
from dkt_amd.io_utils import checkpoint_dir_for, default_image_size, get_assigned_file, get_best_file, model_dict, parse_args
from train import _set_seed


def single_test(params, seed=0):
    if params.method != 'DKT':
        raise ValueError('Unknown method (only DKT is built)')
    if params.kernel_type:
        configs.kernel_type = params.kernel_type
    if params.dataset in ('omniglot', 'cross_char'):
        params.model = 'Conv4S'
    iter_num = params.n_episode or 600
    model = dkt_amd.DKT(model_dict[params.model], n_way=params.test_n_way, n_support=params.n_shot,
                        kernel_type=configs.kernel_type)
    model = model.to(torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
    checkpoint_dir = checkpoint_dir_for(params, configs.save_dir)
    modelfile = get_assigned_file(checkpoint_dir, params.save_iter) if params.save_iter != -1 else get_best_file(checkpoint_dir)
    if modelfile is not None and os.path.isfile(modelfile):
        tmp = torch.load(modelfile, map_location=model.device)
        model.load_state_dict(tmp['state'])
    else:
        print("[WARNING] no checkpoint found in %s: evaluating randomly initialised weights" % checkpoint_dir)
    split = params.split
    split_str = split + ('_' + str(params.save_iter) if params.save_iter != -1 else '')
    image_size = params.image_size or default_image_size(params.model, params.dataset)
    shard = distributed.shard_episodes(iter_num)
    loader = get_episode_loader(params, split, params.test_n_way, params.n_shot, 15, len(shard), image_size,
                                seed=seed + 1000 * distributed.rank())
    model.eval()
    acc_mean, acc_std = model.test_loop(loader, return_std=True)
    if distributed.rank() == 0:
        os.makedirs('./record', exist_ok=True)
        with open('./record/results.txt', 'a') as f:
            timestamp = time.strftime("%Y%m%d-%H%M%S", time.localtime())
            aug_str = '-aug' if params.train_aug else ''
            aug_str += '-adapted' if params.adaptation else ''
            exp_setting = '%s-%s-%s-%s%s %sshot %sway_train %sway_test' % (
                params.dataset, split_str, params.model, params.method, aug_str, params.n_shot, params.train_n_way, params.test_n_way)
            acc_str = '%d Test Acc = %4.2f%% +- %4.2f%%' % (iter_num, acc_mean, 1.96 * acc_std / np.sqrt(iter_num))
            f.write('Time: %s, Setting: %s, Acc: %s \n' % (timestamp, exp_setting, acc_str))
    return acc_mean


def main(argv=None):
    params = parse_args('test', argv)
    distributed.init_from_env()
    accuracy_list = []
    for i in range(params.seed, params.seed + params.repeat):   # reference test.py:193-196
        _set_seed(i if params.seed != 0 else 0)
        accuracy_list.append(single_test(parse_args('test', argv), seed=i))
    print("-----------------------------")
    print('Seeds = %d | Overall Test Acc = %4.2f%% +- %4.2f%%' % (params.repeat, np.mean(accuracy_list), np.std(accuracy_list)))
    print("-----------------------------")
    return accuracy_list


if __name__ == '__main__':
    main()
