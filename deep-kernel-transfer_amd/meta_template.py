"""Base class giving the method surface DKT keeps (reference methods/meta_template.py:10-100):
ctor fields n_way / n_support / n_query / feature / feat_dim / change_way, `forward`,
`parse_feature`, and the abstract `set_forward` / `set_forward_loss`.  The generic
`train_loop` / `test_loop` / `correct` of the reference serve the non-DKT baselines
(ProtoNet, MatchingNet, ...), which are out of scope; DKT overrides all three."""
from __future__ import annotations

from abc import abstractmethod

import torch.nn as nn


class MetaTemplate(nn.Module):
    def __init__(self, model_func, n_way, n_support, change_way=True):
        super(MetaTemplate, self).__init__()
        self.n_way = n_way
        self.n_support = n_support
        self.n_query = -1                      # set per batch from the input
        self.feature = model_func()
        self.feat_dim = self.feature.final_feat_dim
        self.change_way = change_way

    @abstractmethod
    def set_forward(self, x, is_feature):
        pass

    @abstractmethod
    def set_forward_loss(self, x):
        pass

    def forward(self, x):
        return self.feature.forward(x)

    def parse_feature(self, x, is_feature):
        dev = next(self.parameters()).device
        x = x.to(dev)
        if is_feature:
            z_all = x
        else:
            x = x.contiguous().view(self.n_way * (self.n_support + self.n_query), *x.size()[2:])
            z_all = self.feature.forward(x).view(self.n_way, self.n_support + self.n_query, -1)
        return z_all[:, :self.n_support], z_all[:, self.n_support:]
