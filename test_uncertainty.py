#!/usr/bin/env python
"""Calibration (ECE) evaluation of DKT on top of `DKT.get_logits` -- the reference's test_uncertainty.py
pipeline restricted to the DKT branch (ECELoss :39-94 incl. temperature calibration with LBFGS :62-74,
get_logits_targets :96-225 (DKT: :196-200), main :228-262): 15-bin expected calibration error of
softmax(logits * T) over `n_episode` test episodes, T fitted on a calibration run of validation episodes."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import dkt_amd
from dkt_amd import configs
from dkt_amd.data import get_episode_loader
from dkt_amd.io_utils import checkpoint_dir_for, default_image_size, get_best_file, model_dict, parse_args
from train import _set_seed


class ECELoss(nn.Module):
    """Expected calibration error with equal-width confidence bins + a scalar temperature."""

    def __init__(self, n_bins=15):
        super().__init__()
        bounds = torch.linspace(0, 1, n_bins + 1)
        self.bin_lowers, self.bin_uppers = bounds[:-1], bounds[1:]
        self.temperature = nn.Parameter(torch.ones(1))

    def calibrate(self, logits, labels, iterations=50, lr=0.01):
        nll = nn.CrossEntropyLoss()
        optimizer = torch.optim.LBFGS([self.temperature], lr=lr, max_iter=iterations)

        def closure():
            optimizer.zero_grad()
            loss = nll(logits * self.temperature, labels)
            loss.backward()
            return loss
        optimizer.step(closure)
        return self.temperature.detach().clone()

    def forward(self, logits, labels, temperature=1.0, onevsrest=False):
        logits = logits * temperature
        probs = torch.sigmoid(logits) / torch.sigmoid(logits).sum(1, keepdim=True) if onevsrest else F.softmax(logits, dim=1)
        confidences, predictions = torch.max(probs, 1)
        accuracies = predictions.eq(labels)
        ece = torch.zeros(1, device=logits.device)
        for lo, hi in zip(self.bin_lowers, self.bin_uppers):
            in_bin = confidences.gt(lo.item()) * confidences.le(hi.item())
            prop = in_bin.float().mean()
            if prop.item() > 0:
                ece += torch.abs(confidences[in_bin].mean() - accuracies[in_bin].float().mean()) * prop
        return ece


def get_logits_targets(params, split, n_episode, seed):
    image_size = params.image_size or default_image_size(params.model, params.dataset)
    model = dkt_amd.DKT(model_dict[params.model], n_way=params.test_n_way, n_support=params.n_shot,
                        kernel_type=configs.kernel_type).cuda()
    modelfile = get_best_file(checkpoint_dir_for(params, configs.save_dir))
    if modelfile is not None and os.path.isfile(modelfile):
        model.load_state_dict(torch.load(modelfile, map_location=model.device)['state'])
    model.eval()
    loader = get_episode_loader(params, split, params.test_n_way, params.n_shot, 15, n_episode, image_size, seed=seed)
    logits_list, targets_list = [], []
    for x, _ in loader:
        logits_list.append(model.get_logits(x).detach())
        targets_list.append(torch.arange(params.test_n_way, device=model.device).repeat_interleave(15))
    return torch.cat(logits_list, 0), torch.cat(targets_list, 0)


def main(argv=None):
    params = parse_args('test', argv)
    if params.kernel_type:
        configs.kernel_type = params.kernel_type
    _set_seed(params.seed)
    n_ep = params.n_episode or 600
    ece_module = ECELoss().cuda()
    logits, targets = get_logits_targets(params, 'val', max(n_ep // 2, 1), params.seed)       # calibration split
    temperature = ece_module.calibrate(logits, targets)
    print("Calibration: temperature = %.4f" % temperature.item())
    ece_list = []
    for i in range(params.repeat):
        logits, targets = get_logits_targets(params, params.split, n_ep, params.seed + 1 + i)
        ece = ece_module(logits, targets, temperature).item()
        acc = (logits.argmax(1) == targets).float().mean().item() * 100.0
        print("Repeat %d | ECE = %.4f | Acc = %.2f%%" % (i, ece, acc))
        ece_list.append(ece)
    print("-----------------------------")
    print('Seeds = %d | Overall ECE = %.4f +- %.4f' % (params.repeat, np.mean(ece_list), np.std(ece_list)))
    print("-----------------------------")
    return ece_list


if __name__ == '__main__':
    main()
