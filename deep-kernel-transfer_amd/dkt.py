"""Classification DKT on the HIP hot path -- keeps the method surface of the reference's
`methods/DKT.py` (class DKT :32-335) so it drops into train.py / test.py / test_uncertainty.py:

    DKT(model_func, n_way, n_support)        train.py:145, test.py:80
    .init_summary()                          train.py:146
    .train_loop(epoch, loader, optimizer)    train.py:50     (optimizer ignored, Adam re-created: DKT.py:114)
    .test_loop(loader, record=None, return_std=False)        train.py:56, test.py:161
    .correct(x, N=0, laplace=False) -> (top1_correct, count, avg_loss)       DKT.py:199-272
    .get_logits(x) -> [n_way*n_query, n_way]                 test_uncertainty.py:197
    .set_forward / .set_forward_loss         stubs (DKT.py:73-77)
    .feature / .feature_extractor / .model / .likelihood / .mll / .normalize / .iteration / .writer

What changed underneath: the n_way GPyTorch ExactGP models + SumMarginalLogLikelihood are replaced by
stacked hyper-parameters (gp.ExactGPHypers) and the HIP kernels behind ops.* -- ONE Gram per episode
instead of one per class, all class Choleskys in one launch, gradients produced in the same launch,
no per-class host syncs (DKT.py:151-154, 180, 190).
"""
from __future__ import annotations

from time import gmtime, strftime

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import configs, distributed, ops
from .gp import ExactGPHypers, LINEAR_KINDS, RBF_KINDS
from .meta_template import MetaTemplate

try:  # optional, as in the reference (DKT.py:16-21)
    from tensorboardX import SummaryWriter
    IS_TBX_INSTALLED = True
except ImportError:
    IS_TBX_INSTALLED = False


class _LikelihoodView(nn.Module):
    """`model.likelihood`: the Gaussian noise lives in ExactGPHypers.raw_noise (shared storage)."""

    def __init__(self, hypers: ExactGPHypers):
        super().__init__()
        object.__setattr__(self, "_hypers", hypers)

    @property
    def noise(self):
        return self._hypers.noise


class _MllView(nn.Module):
    """`model.mll`: callable kept for surface parity; evaluates the exact MLL of stored train data."""

    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, "_owner", owner)

    def forward(self, z, targets):
        loss, _ = self._owner._episode_loss(z, targets)
        return -loss


class _FusableBatchNorm1d(nn.BatchNorm1d):
    """bn_out (DKT.py:48) with the same parameters / buffers / state-dict keys as nn.BatchNorm1d; `bypass` lets the fused
    front end (ops.episode_loss_bn) take the trunk output in front of it and run the normalisation inside the Gram kernels."""
    bypass = False

    def forward(self, x):
        return x if self.bypass else super().forward(x)


class _MetaBatched:
    """Groups `nb` consecutive episodes of an episodic loader into one item x:[nb, C, S+Q, ch, H, W] (a short tail is dropped)."""

    def __init__(self, loader, nb):
        self.loader, self.nb = loader, nb
        if len(loader) // nb == 0:
            raise ValueError("meta_batch = %d exceeds the %d episodes of the loader: no optimizer step would be taken" % (nb, len(loader)))

    def __len__(self):
        return len(self.loader) // self.nb

    def __iter__(self):
        xs, ys = [], []
        for x, y in self.loader:
            xs.append(x)
            ys.append(y)
            if len(xs) == self.nb:
                yield torch.stack(xs, 0), ys
                xs, ys = [], []


class _GraphedTrainStep:
    """One training step (backbone forward / backward, the GP hot path, Adam) captured into a hipGraph (torch.cuda.CUDAGraph):
    three eager warm-up steps on a side stream -- they are real steps on real episodes --, then capture, then one graph launch
    per episode.  Static input buffer `x`; the outputs (loss, aux, z_train) are static tensors overwritten by every replay."""

    WARMUP = 3

    def __init__(self, model, optimizer, x_like, y_targets, nb, n_ep, key):
        self.m, self.opt, self.y, self.nb, self.n_ep, self.key = model, optimizer, y_targets, nb, n_ep, key
        self.x = torch.empty_like(x_like)
        self.bad = torch.zeros((), device=x_like.device, dtype=torch.float32)
        self.graph, self.out, self.steps = None, None, 0
        self.side = torch.cuda.Stream(device=x_like.device)

    def _step(self):
        loss, aux, z_train, fused = self.m._train_forward(self.x, self.y, self.nb, self.n_ep, True)
        loss.backward()
        self.bad.add_(aux["info"].abs().max().float())
        self.opt.step()
        return loss.detach(), aux, z_train, fused

    def run(self, x_dev):
        self.x.copy_(x_dev)
        if self.graph is not None:
            self.graph.replay()
            return self.out
        cur = torch.cuda.current_stream()
        if self.steps < self.WARMUP:
            self.steps += 1
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.opt.zero_grad(set_to_none=True)
                out = self._step()
            cur.wait_stream(self.side)
            return out
        self.opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.out = self._step()
        self.graph = g
        g.replay()                       # capture only records: run the step for this episode
        return self.out


class DKT(MetaTemplate):
    def __init__(self, model_func, n_way, n_support, kernel_type=None):
        super(DKT, self).__init__(model_func, n_way, n_support)
        self.kernel_type = configs.kernel_type if kernel_type is None else kernel_type
        self.leghtscale_list = None
        self.noise_list = None
        self.outputscale_list = None
        self.iteration = 0
        self.writer = None
        self.feature_extractor = self.feature          # same module under two names (DKT.py:41)
        self.get_model_likelihood_mll()
        if self.kernel_type == "cossim":
            self.normalize = True
        elif self.kernel_type == "bncossim":
            self.normalize = True
            latent_size = int(np.prod(self.feature_extractor.final_feat_dim))
            self.feature_extractor.trunk.add_module("bn_out", _FusableBatchNorm1d(latent_size))
        else:
            self.normalize = False
        self.jitter0 = 1e-6       # gpytorch psd_safe_cholesky fp32 default
        self.max_tries = 3
        self._target_cache = {}
        self._grad_bucket = None
        self._last = {}

    # ------------------------------------------------------------------ construction
    def init_summary(self):
        if IS_TBX_INSTALLED:
            time_string = strftime("%d%m%Y_%H%M%S", gmtime())
            self.writer = SummaryWriter(log_dir="./log/" + time_string)

    def get_model_likelihood_mll(self, train_x_list=None, train_y_list=None):
        """n_way exact GPs sharing the deep kernel; noise = 0.1 frozen (DKT.py:58-71, 346-347)."""
        self.model = ExactGPHypers(self.n_way, self.kernel_type, fixed_noise=0.1)
        self.likelihood = _LikelihoodView(self.model)
        self.mll = _MllView(self)
        return self.model, self.likelihood, self.mll

    def load_state_dict(self, state_dict, strict=True):
        """Accepts this module's own state dict AND one written by the reference's DKT (train.py:61,65: {'epoch', 'state'} with
        the GPyTorch module tree `model.models.{c}.*`, `likelihood.likelihoods.{c}.*`, `mll.*` next to `feature.*` /
        `feature_extractor.*`): the backbone tensors load by name, the GP hyper-parameters of every class through
        ExactGPHypers.load_reference_state_dict."""
        if not ExactGPHypers.is_reference_state_dict(state_dict):
            return super().load_state_dict(state_dict, strict=strict)
        for prefix in ("feature_extractor.", "feature."):
            fe = {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}
            if fe:
                res = self.feature_extractor.load_state_dict(fe, strict=strict)
                break
        else:
            raise RuntimeError("reference state dict without feature_extractor.* / feature.* tensors")
        if self.model.load_reference_state_dict(state_dict) == 0:
            raise RuntimeError("reference state dict without model.models.{c}.* hyper-parameters")
        return res

    def set_forward(self, x, is_feature=False):
        pass

    def set_forward_loss(self, x):
        pass

    # ------------------------------------------------------------------ helpers
    @property
    def device(self):
        return self.model.mean_constant.device

    def _targets(self, n_way, per_class, device):
        """+-1 one-vs-rest targets [C, N], built on the device once per shape (DKT.py:129-136)."""
        key = (n_way, per_class, str(device))
        y = self._target_cache.get(key)
        if y is None:
            cls = torch.arange(n_way, device=device).repeat_interleave(per_class)
            y = torch.where(cls.unsqueeze(0) == torch.arange(n_way, device=device).unsqueeze(1), 1.0, -1.0)
            y = y.to(torch.float32).contiguous()
            self._target_cache[key] = y
        return y

    def _check_way(self, n_way):
        if n_way != self.model.n_models:
            raise RuntimeError("DKT was built with %d GP models but the episode has %d classes "
                               "(train_n_way must equal test_n_way for DKT)" % (self.model.n_models, n_way))

    def _embed(self, x):
        z = self.feature_extractor.forward(x)
        if self.normalize:
            z = F.normalize(z, p=2, dim=1)
        return z

    def _hypers(self):
        m = self.model
        return m.scale_times_variance(), m.mean, m.noise

    # ---- fused front end: bn_out + F.normalize folded into the Gram kernels (ops.episode_loss_bn) ----
    def _trunk_features(self, x):
        """Backbone output BEFORE bn_out (the module the reference appends to the trunk at DKT.py:48)."""
        bn = getattr(self.feature_extractor.trunk, "bn_out", None)
        if bn is None:
            return self.feature_extractor.forward(x)
        bn.bypass = True
        try:
            return self.feature_extractor.forward(x)
        finally:
            bn.bypass = False

    def _fused_front_end(self, n, d):
        # (n <= 128: the episode-resident kernels of dkt_frontend.hip; up to 448 rows: the streaming kernels of dkt_frontend_big.hip in front of the large-N Gram kernels)
        return (self.kernel_type in ("bncossim", "cossim") and n <= 448 and d % 4 == 0
                and os.environ.get("DKT_FUSED_FRONTEND", "1") != "0")

    def _episode_loss_from_trunk(self, x_feat, y, want_z=True):
        """Training loss of ONE episode from the trunk output x_feat:[N,D] (or the mean over a meta-batch [B,N,D]); bn_out runs
        in train mode (batch statistics of EACH episode, running estimates updated exactly as nn.BatchNorm1d would after seeing
        the episodes one by one) inside the fused kernels.
        Returns (loss, aux, z_train) with z_train the normalised train-mode features (detached) of the first episode, which
        the in-loop evaluation conditions on (DKT.py:170-192)."""
        xb = (x_feat if x_feat.dim() == 3 else x_feat.unsqueeze(0)).contiguous()
        n = xb.shape[1]
        c = y.shape[-2]
        sv, mean, noise = self._hypers()
        cw = torch.full((c,), -1.0 / (c * n), device=xb.device, dtype=torch.float32)
        bn = getattr(self.feature_extractor.trunk, "bn_out", None) if self.kernel_type == "bncossim" else None
        if bn is not None:
            outs = ops.episode_loss_bn(xb, bn.weight, bn.bias, y, sv, mean, noise, cw, eps=bn.eps, jitter0=self.jitter0,
                                       max_tries=self.max_tries, use_bn=True, full=True)
        else:
            outs = ops.episode_loss_bn(xb, None, None, y, sv, mean, noise, cw, jitter0=self.jitter0, max_tries=self.max_tries,
                                       use_bn=False, full=True)
        obj, logp, alpha, info, jit, e, bmean, bvar, a, s, rnorm = outs
        with torch.no_grad():
            if bn is not None and bn.track_running_stats:
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item() + 1)
                nb = xb.shape[0]
                if bn.momentum is None and nb > 1:
                    # cumulative moving average: the factor changes with every episode, 1 / (num_batches_tracked + 1) -- the episodes one by one
                    nbt = int(bn.num_batches_tracked.item())
                    for bi in range(nb):
                        f_ = 1.0 / float(nbt + bi + 1)
                        bn.running_mean.mul_(1.0 - f_).add_(bmean[bi], alpha=f_)
                        bn.running_var.mul_(1.0 - f_).add_(bvar[bi], alpha=f_)
                elif nb == 1:
                    bn.running_mean.mul_(1.0 - mom).add_(bmean[0], alpha=mom)
                    bn.running_var.mul_(1.0 - mom).add_(bvar[0], alpha=mom)
                else:       # nb sequential momentum updates in closed form: r <- (1-m)^nb r + m sum_b (1-m)^(nb-1-b) x_b
                    wts = mom * (1.0 - mom) ** torch.arange(nb - 1, -1, -1, device=xb.device, dtype=torch.float32)
                    bn.running_mean.mul_((1.0 - mom) ** nb).add_((bmean * wts[:, None]).sum(0))
                    bn.running_var.mul_((1.0 - mom) ** nb).add_((bvar * wts[:, None]).sum(0))
                bn.num_batches_tracked += nb
            z_train = None
            if want_z:
                z_train = (xb[0].detach() * a.reshape(-1, xb.shape[2])[0] + s.reshape(-1, xb.shape[2])[0]) * rnorm[0].unsqueeze(1)
        aux = dict(logp=logp, alpha=alpha, info=info, jitter=jit, e=None if e is None else e.detach())
        return obj.mean(), aux, z_train

    def _episode_loss(self, z, y):
        """loss = -(1/C) sum_c logp_c / N for ONE episode z:[N,D] (or a batch [B,N,D] -> mean over B)."""
        zb = z if z.dim() == 3 else z.unsqueeze(0)
        n = zb.shape[1]
        c = y.shape[-2]
        sv, mean, noise = self._hypers()
        cw = torch.full((c,), -1.0 / (c * n), device=zb.device, dtype=torch.float32)
        if self.kernel_type in LINEAR_KINDS:
            obj, logp, alpha, info, jit, e = ops.episode_loss_linear(zb, y, sv, mean, noise, cw, self.jitter0, self.max_tries,
                                                                     unit_rows=bool(self.normalize))
        else:
            # rbf / matern / polynomial: every class model owns its lengthscale / offset (one ExactGPLayer per class,
            # DKT.py:63-66), so the base matrix differs per class
            ls, off = self.model.lengthscale, self.model.offset
            if ops.mll_per_class_supported(n, c):
                # ONE contraction per episode (squared distances / Gram), the C class maps in one launch, ONE marginal-likelihood
                # launch over all (episode, class) matrices (DKT_MLL_E_PER_CLASS), the chain rule back in two launches
                obj, logp, alpha, info, jit, e = ops.episode_loss_class_kernel(zb, y, sv, mean, noise, cw, self.kernel_type, ls, off,
                                                                               self.jitter0, self.max_tries)
            elif ops.mll_per_class_supported(n, 1):
                # more than 32 classes (dkt_class_kernel_bwd_f32 takes up to 32 class maps per launch): the same one-launch path per GROUP of 32 classes --
                # ceil(C / 32) groups instead of C single-model calls; the class weights already carry 1 / (C N), so the groups' objectives add up
                parts = [ops.episode_loss_class_kernel(zb, y[..., k:k + 32, :].contiguous(), sv[k:k + 32], mean[k:k + 32], noise[k:k + 32], cw[k:k + 32],
                                                       self.kernel_type, None if ls is None else ls[k:k + 32], None if off is None else off[k:k + 32],
                                                       self.jitter0, self.max_tries) for k in range(0, c, 32)]
                obj = torch.stack([pt[0] for pt in parts], 0).sum(0)
                logp, alpha, info, jit, e = (torch.cat([pt[i] for pt in parts], 1) for i in range(1, 6))
            else:
                # N > 447: one Gram + one single-model launch per class (the blocked path serves those sizes)
                objs, logps, alphas, infos, jits = [], [], [], [], []
                for k in range(c):
                    e = ops.base_matrix(zb, self.kernel_type, None if ls is None else ls[k:k + 1], None if off is None else off[k:k + 1])
                    yk = y[..., k:k + 1, :].contiguous()
                    o, lp, al, inf, jt = ops.mll_objective(e, yk, sv[k:k + 1], mean[k:k + 1], noise[k:k + 1], cw[k:k + 1], self.jitter0, self.max_tries)
                    objs.append(o); logps.append(lp); alphas.append(al); infos.append(inf); jits.append(jt)
                obj = torch.stack(objs, 0).sum(0)
                logp, alpha, info, jit = torch.cat(logps, 1), torch.cat(alphas, 1), torch.cat(infos, 1), torch.cat(jits, 1)
        aux = dict(logp=logp, alpha=alpha, info=info, jitter=jit, e=None if e is None else e.detach())
        return obj.mean(), aux

    def _posterior(self, z_cond, y, z_star, e_cond=None):
        """Mean cache on the conditioning set, posterior means [C,M] and labels [M] at z_star."""
        sv, mean, noise = self._hypers()
        ls, off = self.model.lengthscale, self.model.offset
        ls = None if ls is None else ls.detach()
        off = None if off is None else off.detach()
        zc = z_cond.detach().unsqueeze(0)
        zs = z_star.detach().unsqueeze(0)
        sv, mean, noise = sv.detach(), mean.detach(), noise.detach()
        if self.kernel_type in LINEAR_KINDS:
            if e_cond is None:
                e_cond = ops.kernel_matrix(zc, None, self.kernel_type)
            out = ops.mll(e_cond, y, sv, mean, noise, jitter0=self.jitter0, max_tries=self.max_tries)
            mu, labels = ops.predict(ops.kernel_matrix(zs, zc, self.kernel_type), out["alpha"], sv, mean)
            return mu[0], labels[0], out
        # per-class base matrices (E depends on the class model's own, post-step, lengthscale / offset)
        if ops.mll_per_class_supported(zc.shape[1], y.shape[-2]):
            # one contraction for the conditioning set, one for the cross kernel; the class maps element-wise; one launch for all classes
            e_c = ops.kernel_matrix_per_class(zc, None, self.kernel_type, ls, off)               # [1, C, N, N]
            out = ops.mll(e_c, y, sv, mean, noise, jitter0=self.jitter0, max_tries=self.max_tries)
            ex_c = ops.kernel_matrix_per_class(zs, zc, self.kernel_type, ls, off)                # [1, C, M, N]
            mu, labels = ops.predict(ex_c, out["alpha"], sv, mean)                               # dkt_predict_per_class_f32
            return mu[0], labels[0], {key: out[key] for key in ("logp", "alpha", "jitter", "info")}
        mus, outs = [], []
        if ops.mll_per_class_supported(zc.shape[1], 1):
            # more than 32 classes: the one-launch path per group of 32 classes
            for k in range(0, y.shape[-2], 32):
                lk, ok = (None if ls is None else ls[k:k + 32]), (None if off is None else off[k:k + 32])
                o = ops.mll(ops.kernel_matrix_per_class(zc, None, self.kernel_type, lk, ok), y[..., k:k + 32, :].contiguous(), sv[k:k + 32], mean[k:k + 32],
                            noise[k:k + 32], jitter0=self.jitter0, max_tries=self.max_tries)
                m, _ = ops.predict(ops.kernel_matrix_per_class(zs, zc, self.kernel_type, lk, ok), o["alpha"], sv[k:k + 32], mean[k:k + 32], want_labels=False)
                mus.append(m); outs.append(o)
        else:
            for k in range(y.shape[-2]):                     # N > 447: one single-model call per class
                lk, ok = (None if ls is None else ls[k:k + 1]), (None if off is None else off[k:k + 1])
                o = ops.mll(ops.kernel_matrix(zc, None, self.kernel_type, lk, ok), y[..., k:k + 1, :].contiguous(), sv[k:k + 1], mean[k:k + 1],
                            noise[k:k + 1], jitter0=self.jitter0, max_tries=self.max_tries)
                m, _ = ops.predict(ops.kernel_matrix(zs, zc, self.kernel_type, lk, ok), o["alpha"], sv[k:k + 1], mean[k:k + 1], want_labels=False)
                mus.append(m); outs.append(o)
        mu = torch.cat(mus, 1)
        out = {key: torch.cat([o[key] for o in outs], 1) for key in ("logp", "alpha", "jitter", "info")}
        # first maximum wins, as np.argmax (torch.argmax does not promise which of several equal maxima it returns on the GPU)
        cidx = torch.arange(mu.shape[1], device=mu.device, dtype=torch.int32).view(1, -1, 1)
        labels = torch.where(mu == mu.max(1, keepdim=True).values, cidx, torch.full_like(cidx, mu.shape[1])).min(1).values
        return mu[0], labels[0], out

    def _posterior_fused_eval(self, x_support, x_query, y):
        """Test-time episode with bn_out (eval mode: running statistics) + F.normalize folded into ONE Gram launch over the
        stacked [support; query] trunk features (one backbone pass instead of two): E_all = Zn Zn^T, the conditioning matrix is
        its [:ns, :ns] block and the cross kernel its [ns:, :ns] block.  Returns None when the fused kernels do not apply."""
        ns, nq = x_support.shape[0], x_query.shape[0]
        bn = getattr(self.feature_extractor.trunk, "bn_out", None) if self.kernel_type == "bncossim" else None
        if self.feature_extractor.training or (bn is not None and not bn.track_running_stats):
            return None
        x_feat = self._trunk_features(torch.cat([x_support, x_query], 0)).detach()
        d = x_feat.shape[1]
        if x_feat.dim() != 2 or not self._fused_front_end(ns + nq, d):
            return None
        if bn is not None:
            a = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps) if bn.affine else torch.rsqrt(bn.running_var + bn.eps)
            s = (bn.bias.detach() if bn.affine else 0.0) - bn.running_mean * a
        else:
            a = torch.ones(d, device=x_feat.device, dtype=torch.float32)
            s = torch.zeros(d, device=x_feat.device, dtype=torch.float32)
        if ns + nq <= ops.FUSED_EP_MAX_N:
            e_all, _ = ops.gram_bn(x_feat.unsqueeze(0).contiguous(), a.contiguous(), s.contiguous())
        else:                                                 # (more than 128 rows: one normalisation kernel in front of the large-N Gram kernel)
            zn, _ = ops.affine_normalize(x_feat.unsqueeze(0).contiguous(), a.contiguous(), s.contiguous())
            e_all = ops.gram(zn, None, ops.KERNEL_LINEAR_UNIT)
        sv, mean, noise = self._hypers()
        out = ops.mll(e_all[:, :ns, :ns].contiguous(), y, sv.detach(), mean.detach(), noise.detach(), jitter0=self.jitter0,
                      max_tries=self.max_tries)
        mu, labels = ops.predict(e_all[:, ns:, :ns].contiguous(), out["alpha"], sv.detach(), mean.detach())
        return mu[0], labels[0], out

    def _train_forward(self, x_all, y_targets, nb, n_ep, want_z=True):
        """Forward of one training step from the uploaded images x_all:[nb * N, ch, H, W]: returns (loss, aux, z_train, fused)."""
        self.model.train()
        self.likelihood.train()
        self.feature_extractor.train()
        # ONE backbone pass over the nb * N images of the step (meta-batch: the backbone's own BatchNorm2d layers then see
        # all of them as one batch, as any mini-batch training does; bn_out and the GPs stay per episode)
        x_feat = self._trunk_features(x_all)
        fused = x_feat.dim() == 2 and self._fused_front_end(n_ep, x_feat.shape[1])
        if fused:
            if nb > 1:
                x_feat = x_feat.view(nb, n_ep, -1)
            loss, aux, z_train = self._episode_loss_from_trunk(x_feat, y_targets, want_z=want_z)
            return loss, aux, z_train, True
        bn = getattr(self.feature_extractor.trunk, "bn_out", None)        # torch bn_out / F.normalize in front of the Gram kernels
        if bn is None:
            z_train = x_feat
        elif nb == 1:
            z_train = bn(x_feat)
        else:                                                               # per-episode batch statistics, like the fused path
            z_train = torch.cat([bn(x_feat[k * n_ep:(k + 1) * n_ep]) for k in range(nb)], 0)
        if self.normalize:
            z_train = F.normalize(z_train, p=2, dim=1)
        if nb > 1:
            z_train = z_train.view(nb, n_ep, -1)
        loss, aux = self._episode_loss(z_train, y_targets)
        return loss, aux, z_train, False

    def _bucket(self):
        """The flat gradient bucket of the multi-rank step (None outside torch.distributed).  Every fp32 parameter's `.grad` is a VIEW of
        it (`attach`), so the backward writes the bucket directly and the all-reduce needs no pack / unpack copies -- provided the
        training loop clears the gradients with `zero_()` / `set_to_none=False` and masks in place, which `train_loop` does."""
        if not distributed.is_distributed():
            return None
        if self._grad_bucket is None:
            self._grad_bucket = distributed.GradBucket(self.parameters())
        self._grad_bucket.attach()           # (re-attaches a gradient somebody replaced; a no-op when the views are in place)
        return self._grad_bucket

    def _zero_grads(self, optimizer):
        """Start of a step: outside torch.distributed the reference's `optimizer.zero_grad()`; with ranks ONE fill of the flat bucket the
        gradients are views of (set_to_none would drop the views and turn the all-reduce back into pack -> reduce -> scatter)."""
        bucket = self._bucket()
        if bucket is None:
            optimizer.zero_grad()
        else:
            bucket.zero_()

    def _sync_grads(self, flag=None):
        """One all-reduce of the flat gradient bucket; `flag` (max |info| of this rank's step) is summed over the ranks in the
        same collective.  Returns the (global) flag."""
        if distributed.is_distributed():
            if self._grad_bucket is None:
                self._grad_bucket = distributed.GradBucket(self.parameters())
            return self._grad_bucket.allreduce_mean(flag)
        return flag

    # ------------------------------------------------------------------ training
    def train_loop(self, epoch, train_loader, optimizer, print_freq=10):
        # the optimizer argument is ignored and Adam re-created every call, as the reference does
        # (same update rule; on the GPU the fused implementation: one launch per parameter group instead of a dozen element-wise
        # ones, and it takes a device-side `found_inf` flag -- used below to SKIP the update of a step whose factorisation failed)
        fused_adam = os.environ.get("DKT_FUSED_ADAM", "1") == "1" and self.device.type == "cuda"
        optimizer = torch.optim.Adam([{'params': self.model.parameters(), 'lr': 1e-4},
                                      {'params': self.feature_extractor.parameters(), 'lr': 1e-3}], **({"fused": True} if fused_adam else {}))
        dev = self.device
        self._bad_steps = None
        # DKT_TRAIN_GRAPH=1: capture the per-episode step into a hipGraph (the loop is launch-bound: ~150 launches for ~1 ms of
        # GPU work).  Needs static shapes, no TensorBoard writer inside the step and a single process.
        use_graph = os.environ.get("DKT_TRAIN_GRAPH", "0") == "1" and not distributed.is_distributed()
        graph_step = None
        if use_graph:
            optimizer = torch.optim.Adam([{'params': self.model.parameters(), 'lr': 1e-4},
                                          {'params': self.feature_extractor.parameters(), 'lr': 1e-3}], capturable=True)
        mb = max(1, int(getattr(self, "meta_batch", 1) or 1))
        if mb > 1:          # opt-in (train.py --meta_batch B): B episodes per Adam step; 1 = the reference's semantics (DKT.py:160-164)
            train_loader = _MetaBatched(train_loader, mb)
        for i, (x, _) in enumerate(train_loader):
            xe = x if x.dim() == 6 else x.unsqueeze(0)            # [B, C, S+Q, ch, H, W]
            nb = xe.size(0)
            self.n_query = xe.size(2) - self.n_support
            if self.change_way:
                self.n_way = xe.size(1)
            self._check_way(self.n_way)
            per = self.n_support + self.n_query
            n_ep = self.n_way * per
            x_all = xe.contiguous().view(nb * n_ep, *xe.size()[3:]).to(dev, non_blocking=True)
            y_targets = self._targets(self.n_way, per, dev)

            graphed = None
            if use_graph:
                key = (tuple(x_all.shape), nb)
                if graph_step is None or graph_step.key != key:
                    graph_step = _GraphedTrainStep(self, optimizer, x_all, y_targets, nb, n_ep, key)
                graphed = graph_step

            # hyper-parameter means for the log line, read BEFORE the step (DKT.py:145-157); kept on
            # the device, converted to Python floats only when printed
            need_eval = self.writer is not None or i % print_freq == 0 or i == len(train_loader) - 1
            if i % print_freq == 0:
                with torch.no_grad():
                    log_outputscale = self.model.outputscale.mean()
                    log_noise = self.model.noise.mean()
                    ls = self.model.lengthscale
                    log_lengthscale = ls.mean() if ls is not None else torch.zeros((), device=dev)

            if graphed is not None:
                # the whole step (backbone forward / backward, the GP kernels, Adam) as ONE hipGraph launch
                loss, aux, z_train, fused = graphed.run(x_all)
                self._bad_steps = graphed.bad
            else:
                self._zero_grads(optimizer)
                loss, aux, z_train, fused = self._train_forward(x_all, y_targets, nb, n_ep, need_eval)
                loss.backward()
                # failure flag of the step (not positive definite after every jitter retry), kept on the device, summed over the
                # ranks with the gradients and accumulated over the iterations: checked -- on every rank alike -- at the next
                # print point (GPyTorch raises NotPSDError synchronously; a failed step has poisoned the update with NaN)
                bad = self._sync_grads(aux["info"].abs().max().float())
                self._bad_steps = bad if self._bad_steps is None else self._bad_steps + bad
                if fused_adam:
                    optimizer.found_inf = (bad != 0).to(torch.float32).reshape(())    # no NaN ever reaches the weights or Adam's moments
                else:
                    # the default (foreach) implementation has no device-side skip: the poisoned gradients are zeroed IN PLACE (the views
                    # into the gradient bucket survive; no allocation, no host sync), so no NaN reaches the weights or the moments.  This
                    # is NOT a skipped step: Adam still advances its step count, decays its moments and moves the weights by the existing
                    # momentum -- only the fused path (found_inf) leaves the optimizer state untouched.
                    failed = (bad != 0).reshape(())
                    for group in optimizer.param_groups:
                        for p_ in group['params']:
                            if p_.grad is not None:
                                p_.grad.masked_fill_(failed, 0.0)
                optimizer.step()
            x_all = x_all[:n_ep]                          # the in-loop evaluation looks at the step's first episode

            self.iteration = i + (epoch * len(train_loader))
            if self.writer is not None:
                self.writer.add_scalar('loss', loss, self.iteration)

            # evaluation on support / query with eval-mode features, conditioning on the (stale)
            # train-mode features and the post-step hyper-parameters (DKT.py:170-192).  Its only consumers are the TensorBoard
            # writer and the log line, and it has no side effect (eval-mode BatchNorm), so it runs only when one of them will
            # read it (the reference runs it -- with 2C blocking read-backs -- on every iteration)
            if not need_eval:
                self._last = dict(loss=loss.detach(), acc_support=None, acc_query=None, info=aux["info"])
                continue
            if nb > 1 and not fused:
                z_train = z_train[0].detach()
            e_first = None if aux["e"] is None else aux["e"][:1]      # (None: the episode ran in feature space, ops.lowrank_applies)
            with torch.no_grad():
                self.model.eval()
                self.likelihood.eval()
                self.feature_extractor.eval()
                z_eval = self._embed(x_all).detach().view(self.n_way, per, -1)
                z_support = z_eval[:, :self.n_support].reshape(self.n_way * self.n_support, -1)
                z_query = z_eval[:, self.n_support:].reshape(self.n_way * self.n_query, -1)
                z_star = torch.cat([z_support, z_query], 0)
                _, labels, _ = self._posterior(z_train, y_targets, z_star, e_cond=e_first)
                cls = torch.arange(self.n_way, device=dev, dtype=torch.int32)
                ns = self.n_way * self.n_support
                acc_support = (labels[:ns] == cls.repeat_interleave(self.n_support)).float().mean() * 100.0
                acc_query = (labels[ns:] == cls.repeat_interleave(self.n_query)).float().mean() * 100.0
                if self.writer is not None:
                    self.writer.add_scalar('GP_support_accuracy', acc_support, self.iteration)
                    self.writer.add_scalar('GP_query_accuracy', acc_query, self.iteration)
            self._last = dict(loss=loss.detach(), acc_support=acc_support, acc_query=acc_query, info=aux["info"])

            if i % print_freq == 0:
                if self.writer is not None:
                    self.writer.add_histogram('z_support', z_support, self.iteration)
                if float(self._bad_steps.item()) != 0.0:
                    raise RuntimeError("DKT: kernel matrix not positive definite after jitter retries "
                                       "(GPyTorch raises NotPSDError here)")
                print('Epoch [{:d}] [{:d}/{:d}] | Outscale {:f} | Lenghtscale {:f} | Noise {:f} | Loss {:f} | Supp. {:f} | Query {:f}'.format(
                    epoch, i, len(train_loader), log_outputscale.item(), log_lengthscale.item(), log_noise.item(),
                    loss.item(), acc_support.item(), acc_query.item()))
        # failures after the last print point of the epoch (the flag is reset by the next call): raise here, on every rank alike
        if self._bad_steps is not None and float(self._bad_steps.item()) != 0.0:
            raise RuntimeError("DKT: kernel matrix not positive definite after jitter retries "
                               "(GPyTorch raises NotPSDError here)")

    # ------------------------------------------------------------------ evaluation
    def _upload(self, x):
        """ONE host-to-device copy of the episode's images (the reference uploads support and query separately, DKT.py:201-203).
        Deliberately a blocking copy: it is the only host/device rendez-vous of a test episode, and it keeps the host from running
        arbitrarily far ahead of the GPU -- measured on MI355X, a fully asynchronous pipeline (pinned staging ring, no blocking call
        at all) is 10x SLOWER (11.8 vs 0.73 ms per Conv4S test episode), while every additional blocking call costs ~1.5 ms."""
        return x if x.is_cuda else x.to(self.device)

    def _split(self, x):
        xd = self._upload(x)
        x_support = xd[:, :self.n_support].contiguous().view(self.n_way * self.n_support, *xd.size()[2:])
        x_query = xd[:, self.n_support:].contiguous().view(self.n_way * self.n_query, *xd.size()[2:])
        return x_support, x_query

    def correct(self, x, N=0, laplace=False):
        out = self._correct_device(x, N, laplace)
        if not isinstance(out[0], torch.Tensor):
            return out
        stats, count_this, avg_loss = out
        stats = stats.cpu()                                   # the only read-back of the episode
        if stats[1].item() != 0:
            raise RuntimeError("DKT.correct: kernel matrix not positive definite after jitter retries")
        return float(stats[0].item()), count_this, avg_loss

    def _correct_device(self, x, N=0, laplace=False):
        """`correct` without the read-back: returns (stats, count, avg_loss) with stats = [top1_correct, max |info|] on the device."""
        self._check_way(self.n_way)
        x_support, x_query = self._split(x)
        y_query = np.repeat(range(self.n_way), self.n_query)

        if laplace:   # sklearn Laplace GPC, "not the method used in the paper" (DKT.py:207-222)
            from sklearn.gaussian_process import GaussianProcessClassifier
            from sklearn.gaussian_process.kernels import RBF
            y_support = np.repeat(range(self.n_way), self.n_support)
            kernel = 1.0 * RBF(length_scale=0.1, length_scale_bounds=(0.1, 10.0))
            gp = GaussianProcessClassifier(kernel=kernel, optimizer=None)
            with torch.no_grad():
                z_support = self._embed(x_support).detach()
                z_query = self._embed(x_query).detach()
            gp.fit(z_support.cpu().numpy(), y_support)
            y_pred = gp.predict(z_query.cpu().numpy())
            return float(np.sum(y_pred == y_query)), len(y_query), 0.0

        dev = self.device
        y_targets = self._targets(self.n_way, self.n_support, dev)
        fused = None
        if N == 0:
            with torch.no_grad():
                fused = self._posterior_fused_eval(x_support, x_query, y_targets)
        z_train = self._embed(x_support).detach() if fused is None else None

        self.model.train()
        self.likelihood.train()
        self.feature_extractor.eval()

        avg_loss = 0.0
        if N > 0:   # test-time adaptation of the GP hyper-parameters only (DKT.py:242-256)
            optimizer = torch.optim.Adam([{'params': self.model.parameters()}], lr=1e-3)
            for _ in range(0, N):
                optimizer.zero_grad()
                loss, _ = self._episode_loss(z_train, y_targets)
                loss.backward()
                optimizer.step()
                avg_loss = avg_loss + loss.item()

        with torch.no_grad():
            self.model.eval()
            self.likelihood.eval()
            self.feature_extractor.eval()
            if fused is None:
                z_query = self._embed(x_query).detach()
                _, labels, out = self._posterior(z_train, y_targets, z_query)
            else:
                _, labels, out = fused
            y_q = torch.arange(self.n_way, device=dev, dtype=torch.int32).repeat_interleave(self.n_query)
            stats = torch.stack([(labels == y_q).sum().float(), out["info"].abs().max().float()])
            count_this = len(y_query)
        return stats, count_this, avg_loss / float(N + 1e-10)

    def test_loop(self, test_loader, record=None, return_std=False):
        acc_all, pending = [], []
        iter_num = len(test_loader)
        for i, (x, _) in enumerate(test_loader):
            self.n_query = x.size(1) - self.n_support
            if self.change_way:
                self.n_way = x.size(0)
            # the per-episode counts stay on the device and are read back in one go at the print points: no blocking call per episode
            stats, count_this, loss_value = self._correct_device(x)
            pending.append((stats, count_this))
            if i % 100 == 0 or i == iter_num - 1:
                got = torch.stack([p[0] for p in pending]).cpu().numpy()
                if (got[:, 1] != 0).any():
                    raise RuntimeError("DKT.test_loop: kernel matrix not positive definite after jitter retries")
                acc_all.extend((got[:, 0] / np.asarray([p[1] for p in pending]) * 100).tolist())
                pending = []
            if i % 100 == 0:
                acc_mean = np.mean(np.asarray(acc_all))
                print('Test | Batch {:d}/{:d} | Loss {:f} | Acc {:f}'.format(i, len(test_loader), loss_value, acc_mean))
        if pending:                                            # a loader whose length is unknown / shorter than announced
            got = torch.stack([p[0] for p in pending]).cpu().numpy()
            if (got[:, 1] != 0).any():
                raise RuntimeError("DKT.test_loop: kernel matrix not positive definite after jitter retries")
            acc_all.extend((got[:, 0] / np.asarray([p[1] for p in pending]) * 100).tolist())
        if distributed.is_distributed():   # every rank evaluated its own shard of the episode list
            acc_all = distributed.gather_accuracies(acc_all)
            iter_num = len(acc_all)
        acc_all = np.asarray(acc_all)
        acc_mean = np.mean(acc_all)
        acc_std = np.std(acc_all)
        print('%d Test Acc = %4.2f%% +- %4.2f%%' % (iter_num, acc_mean, 1.96 * acc_std / np.sqrt(iter_num)))
        if self.writer is not None:
            self.writer.add_scalar('test_accuracy', acc_mean, self.iteration)
        if return_std:
            return acc_mean, acc_std
        return acc_mean

    def get_logits(self, x):
        self.n_query = x.size(1) - self.n_support
        self._check_way(self.n_way)
        x_support, x_query = self._split(x)
        y_targets = self._targets(self.n_way, self.n_support, self.device)
        with torch.no_grad():
            fused = self._posterior_fused_eval(x_support, x_query, y_targets)
        z_train = self._embed(x_support).detach() if fused is None else None
        with torch.no_grad():
            self.model.eval()
            self.likelihood.eval()
            self.feature_extractor.eval()
            if fused is None:
                z_query = self._embed(x_query).detach()
                mu, _, _ = self._posterior(z_train, y_targets, z_query)
            else:
                mu = fused[0]
        return mu.t().contiguous()    # [n_way*n_query, n_way] raw posterior means (DKT.py:331-335)
