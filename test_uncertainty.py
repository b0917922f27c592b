#!/usr/bin/env python
"""Calibration (ECE) evaluation of DKT on top of `DKT.get_logits` -- the reference's test_uncertainty.py
pipeline restricted to the DKT branch (ECELoss :39-94 incl. temperature calibration with LBFGS :62-74,
get_logits_targets :96-225 (DKT: :196-200), main :228-262): 15-bin expected calibration error of
softmax(logits / T) over `n_episode` test episodes.  Protocol as in the reference's main(): T = mean over `repeat`
calibration runs (each: seed 0 -- "unseeded" --, the evaluated split, LBFGS lr 0.01 / 300 iterations on the raw
temperature, runs with T <= 0 dropped; T = 1 if none is left), then `repeat` evaluation runs with seeds
seed .. seed + repeat - 1 (seed 0: every run unseeded)."""
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

    def calibrate(self, logits, labels, iterations=50, lr=0.01):
        """One LBFGS step (max_iter = iterations) on the RAW temperature, initial value 1, NLL of logits / T
        (reference test_uncertainty.py:62-74).  Returns the temperature (it may come out <= 0: the caller filters)."""
        temperature_raw = torch.ones(1, requires_grad=True, device=logits.device)
        nll = nn.CrossEntropyLoss()
        optimizer = torch.optim.LBFGS([temperature_raw], lr=lr, max_iter=iterations)

        def closure():
            if torch.is_grad_enabled():
                optimizer.zero_grad()
            loss = nll(logits / temperature_raw.expand_as(logits), labels)
            if loss.requires_grad:
                loss.backward()
            return loss
        optimizer.step(closure)
        return temperature_raw.detach().clone()

    def forward(self, logits, labels, temperature=1.0, onevsrest=False):
        logits = logits / temperature
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
    if seed is None:                                          # seed 0 = "unseeded" in the reference (train.py:24-35)
        seed = int(np.random.randint(1 << 30))
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
    n_ep = params.n_episode or 600
    ece_module = ECELoss()
    # 1. temperature: `repeat` calibration runs at seed 0 on the evaluated split, T <= 0 dropped, mean (reference :236-245)
    print("Calibration: finding temperature hyperparameter...")
    temperature_list = []
    for r in range(params.repeat):
        _set_seed(0)
        logits, targets = get_logits_targets(params, params.split, n_ep, None if params.seed == 0 else 10007 * (r + 1))
        t = ece_module.calibrate(logits, targets, iterations=300, lr=0.01).item()
        if t > 0:
            temperature_list.append(t)
        print("Calibration: temperature", t, "; mean temperature", np.mean(temperature_list) if temperature_list else float('nan'))
    temperature = float(np.mean(temperature_list)) if temperature_list else 1.0
    # 2. ECE over seeds seed .. seed + repeat - 1 (reference :247-257)
    ece_list = []
    for i in range(params.seed, params.seed + params.repeat):
        _set_seed(i if params.seed != 0 else 0)
        logits, targets = get_logits_targets(params, params.split, n_ep, i if params.seed != 0 else None)
        ece = ece_module(logits, targets, temperature, onevsrest=False).item()
        ece_list.append(ece)
        print("ECE:", np.mean(ece_list), "+-", np.std(ece_list))
    print("-----------------------------")
    print('Seeds = %d | Overall ECE = %4.4f +- %4.4f' % (params.repeat, np.mean(ece_list), np.std(ece_list)))
    print("-----------------------------")
    return ece_list, temperature


if __name__ == '__main__':
    main()
