#!/usr/bin/env python
"""DKT training driver with the reference's command line, epoch loop, checkpoint format and resume logic
(reference train.py:24-67 `_set_seed` / `train`, :70-219 `__main__`), running on the HIP hot path.

  python train.py --method DKT --model Conv4 --train_n_way 5 --test_n_way 5 --n_shot 5 [--stop_epoch 3]
  torchrun --nproc-per-node 8 train.py ...     (episode-parallel: every rank draws its own episodes)
"""
import os
import random

import numpy as np
import torch

import dkt_amd
from dkt_amd import configs, distributed
from dkt_amd.data import get_episode_loader
from dkt_amd.io_utils import checkpoint_dir_for, default_image_size, get_resume_file, model_dict, parse_args


def _set_seed(seed, verbose=True):
    if seed != 0:
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        torch.backends.cudnn.deterministic = True             # reference train.py:31-32 (MIOpen honours both flags on ROCm)
        torch.backends.cudnn.benchmark = False
        if verbose:
            print("[INFO] Setting SEED: " + str(seed))
    elif verbose:
        print("[INFO] Setting SEED: None")


def train(base_loader, val_loader, model, optimization, start_epoch, stop_epoch, params):
    print("Tot epochs: " + str(stop_epoch))
    if optimization != 'Adam':
        raise ValueError('Unknown optimization, please define by yourself')
    optimizer = torch.optim.Adam(model.parameters())     # ignored by DKT.train_loop, as in the reference
    max_acc = 0
    for epoch in range(start_epoch, stop_epoch):
        model.train()
        model.train_loop(epoch, base_loader, optimizer)
        # every rank saw its own episodes: the BatchNorm running estimates differ per rank.  Average them (SURVEY.md 8e:
        # "BN running stats: broadcast / average buffers at checkpoint time") so that the checkpoint does not depend on
        # which rank writes it, and every rank validates / continues with the same buffers.
        distributed.average_module_buffers(model)
        model.eval()
        if not os.path.isdir(params.checkpoint_dir):
            os.makedirs(params.checkpoint_dir, exist_ok=True)
        acc = model.test_loop(val_loader)
        if distributed.rank() != 0:
            continue
        if acc > max_acc:
            print("--> Best model! save...")
            max_acc = acc
            torch.save({'epoch': epoch, 'state': model.state_dict()}, os.path.join(params.checkpoint_dir, 'best_model.tar'))
        if (epoch % params.save_freq == 0) or (epoch == stop_epoch - 1):
            torch.save({'epoch': epoch, 'state': model.state_dict()}, os.path.join(params.checkpoint_dir, '{:d}.tar'.format(epoch)))
    return model


def main(argv=None):
    params = parse_args('train', argv)
    local = distributed.init_from_env()
    _set_seed(params.seed + (distributed.rank() if params.seed else 0))
    if params.method != 'DKT':
        raise ValueError('Unknown method (only DKT is built)')
    if params.kernel_type:
        configs.kernel_type = params.kernel_type
    if params.dataset in ('omniglot', 'cross_char'):
        assert params.model == 'Conv4' and not params.train_aug, 'omniglot only support Conv4 without augmentation'
        params.model = 'Conv4S'
    image_size = params.image_size or default_image_size(params.model, params.dataset)
    if params.stop_epoch == -1:                               # reference train.py:97-113
        params.stop_epoch = 600 if params.n_shot == 1 else 400
    n_query = max(1, int(16 * params.test_n_way / params.train_n_way))
    n_ep = params.n_episode or 100
    base_loader = get_episode_loader(params, 'base', params.train_n_way, params.n_shot, n_query, n_ep, image_size,
                                     seed=params.seed + 100 * distributed.rank())
    # validation episodes are sharded like test.py's: rank r evaluates its own len(shard) episodes (rank-dependent seed) and
    # DKT.test_loop gathers the per-episode accuracies -- n_ep independent episodes in total, not W copies of the same ones
    val_shard = distributed.shard_episodes(n_ep)
    val_loader = get_episode_loader(params, 'val', params.test_n_way, params.n_shot, n_query, len(val_shard), image_size,
                                    seed=params.seed + 1000 * distributed.rank())

    model = dkt_amd.DKT(model_dict[params.model], n_way=params.train_n_way, n_support=params.n_shot,
                        kernel_type=configs.kernel_type)
    model.init_summary()
    model.meta_batch = max(1, getattr(params, 'meta_batch', 1))
    model = model.to(torch.device('cuda', local))
    distributed.broadcast_module_state(model)

    params.checkpoint_dir = checkpoint_dir_for(params, configs.save_dir)
    os.makedirs(params.checkpoint_dir, exist_ok=True)
    start_epoch, stop_epoch = params.start_epoch, params.stop_epoch
    if params.resume:
        resume_file = get_resume_file(params.checkpoint_dir)
        if resume_file is not None:
            tmp = torch.load(resume_file, map_location=model.device)
            start_epoch = tmp['epoch'] + 1
            model.load_state_dict(tmp['state'])
    elif params.warmup:
        raise ValueError('No warm_up file (the baseline method is not built)')
    return train(base_loader, val_loader, model, 'Adam', start_epoch, stop_epoch, params)


if __name__ == '__main__':
    main()
