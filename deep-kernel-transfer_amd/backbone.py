"""Feature extractors that FEED the DKT hot path (host-side PyTorch-ROCm / MIOpen code, not the
product).  Same factories, `.trunk` / `.final_feat_dim` surface and state-dict key names as the
reference's backbone.py (ConvNet :250-268, ConvNetS :287-310, ResNet :330-376, Conv3 :379-402,
factories :404-426), because DKT appends `bn_out` to `feature_extractor.trunk` (DKT.py:48) and
reference checkpoints address parameters as `trunk.<i>.C.weight` / `trunk.<i>.trunk.0.weight`.
MAML fast-weight layers, ResNet50/101 and the *NP variants are out of scope.
"""
from __future__ import annotations

import math

import torch.nn as nn
import torch.nn.functional as F


def init_layer(layer: nn.Module) -> None:
    """He-style fan-OUT normal init for convs, (1, 0) for 2-d batch norms (backbone.py:13-20)."""
    if isinstance(layer, nn.Conv2d):
        fan = layer.kernel_size[0] * layer.kernel_size[1] * layer.out_channels
        layer.weight.data.normal_(0.0, math.sqrt(2.0 / float(fan)))
    elif isinstance(layer, nn.BatchNorm2d):
        layer.weight.data.fill_(1.0)
        layer.bias.data.fill_(0.0)


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class ConvBlock(nn.Module):
    """conv3x3 -> BN -> ReLU (-> maxpool 2).  Sub-modules are registered both by name (C, BN, relu,
    pool) and inside `trunk`, which is what produces the reference's duplicated state-dict keys."""

    def __init__(self, indim: int, outdim: int, pool: bool = True, padding: int = 1):
        super().__init__()
        self.indim, self.outdim = indim, outdim
        self.C = nn.Conv2d(indim, outdim, 3, padding=padding)
        self.BN = nn.BatchNorm2d(outdim)
        self.relu = nn.ReLU(inplace=True)
        layers = [self.C, self.BN, self.relu]
        if pool:
            self.pool = nn.MaxPool2d(2)
            layers.append(self.pool)
        for layer in layers:
            init_layer(layer)
        self.trunk = nn.Sequential(*layers)

    def forward(self, x):
        return self.trunk(x)


class ConvNet(nn.Module):
    """Conv-`depth`, 3-channel input; 84x84 -> 1600 features for depth 4."""

    def __init__(self, depth: int, in_channels: int = 3, feat_dim: int = 1600, first_channel_only: bool = False):
        super().__init__()
        blocks = [ConvBlock(in_channels if i == 0 else 64, 64, pool=(i < 4)) for i in range(depth)]
        blocks.append(Flatten())
        self.trunk = nn.Sequential(*blocks)
        self.final_feat_dim = feat_dim
        self.first_channel_only = first_channel_only

    def forward(self, x):
        if self.first_channel_only:       # Conv4S reads channel 0 only (backbone.py:307)
            x = x[:, 0:1, :, :]
        return self.trunk(x)


class SimpleBlock(nn.Module):
    def __init__(self, indim: int, outdim: int, half_res: bool):
        super().__init__()
        stride = 2 if half_res else 1
        self.indim, self.outdim, self.half_res = indim, outdim, half_res
        self.C1 = nn.Conv2d(indim, outdim, 3, stride=stride, padding=1, bias=False)
        self.BN1 = nn.BatchNorm2d(outdim)
        self.C2 = nn.Conv2d(outdim, outdim, 3, padding=1, bias=False)
        self.BN2 = nn.BatchNorm2d(outdim)
        self.relu1 = nn.ReLU(inplace=True)
        self.relu2 = nn.ReLU(inplace=True)
        layers = [self.C1, self.C2, self.BN1, self.BN2]
        if indim != outdim:
            self.shortcut = nn.Conv2d(indim, outdim, 1, stride, bias=False)
            self.BNshortcut = nn.BatchNorm2d(outdim)
            layers += [self.shortcut, self.BNshortcut]
            self.shortcut_type = "1x1"
        else:
            self.shortcut_type = "identity"
        for layer in layers:
            init_layer(layer)

    def forward(self, x):
        out = self.relu1(self.BN1(self.C1(x)))
        out = self.BN2(self.C2(out))
        res = x if self.shortcut_type == "identity" else self.BNshortcut(self.shortcut(x))
        return self.relu2(out + res)


class ResNet(nn.Module):
    """224x224 -> 512 features (simple blocks)."""

    def __init__(self, layers_per_stage, stage_dims=(64, 128, 256, 512)):
        super().__init__()
        assert len(layers_per_stage) == 4, "Can have only four stages"
        conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        bn1 = nn.BatchNorm2d(64)
        init_layer(conv1)
        init_layer(bn1)
        trunk = [conv1, bn1, nn.ReLU(), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
        indim = 64
        for stage, (count, outdim) in enumerate(zip(layers_per_stage, stage_dims)):
            for j in range(count):
                trunk.append(SimpleBlock(indim, outdim, half_res=(stage >= 1 and j == 0)))
                indim = outdim
        trunk += [nn.AvgPool2d(7), Flatten()]
        self.trunk = nn.Sequential(*trunk)
        self.final_feat_dim = indim

    def forward(self, x):
        return self.trunk(x)


class Conv3(nn.Module):
    """QMUL head-pose backbone: three stride-2 dilated convs, 100x100 -> 2916 features."""

    def __init__(self):
        super().__init__()
        self.layer1 = nn.Conv2d(3, 36, 3, stride=2, dilation=2)
        self.layer2 = nn.Conv2d(36, 36, 3, stride=2, dilation=2)
        self.layer3 = nn.Conv2d(36, 36, 3, stride=2, dilation=2)
        self.final_feat_dim = 2916

    def return_clones(self):
        return [l.weight.data.clone().detach() for l in (self.layer1, self.layer2, self.layer3)]

    def assign_clones(self, weights_list):
        for l, w in zip((self.layer1, self.layer2, self.layer3), weights_list):
            l.weight.data.copy_(w)

    def forward(self, x):
        out = F.relu(self.layer1(x))
        out = F.relu(self.layer2(out))
        out = F.relu(self.layer3(out))
        return out.view(out.size(0), -1)


def Conv4():
    return ConvNet(4)


def Conv6():
    return ConvNet(6)


def Conv4S():
    return ConvNet(4, in_channels=1, feat_dim=64, first_channel_only=True)


def ResNet10():
    return ResNet([1, 1, 1, 1])


def ResNet18():
    return ResNet([2, 2, 2, 2])


def ResNet34():
    return ResNet([3, 4, 6, 3])


model_dict = dict(Conv4=Conv4, Conv4S=Conv4S, Conv6=Conv6, ResNet10=ResNet10, ResNet18=ResNet18, ResNet34=ResNet34)
