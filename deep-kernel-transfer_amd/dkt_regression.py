"""Regression DKT (QMUL head pose) on the HIP hot path -- keeps the surface of the reference's
`methods/DKT_regression.py` (class DKT :18-110):

    DKT(backbone)                               train_regression.py:27
    .model.parameters() / .feature_extractor.parameters()   external Adam, train_regression.py:33-34
    .train_loop(epoch, optimizer)               train_regression.py:37
    .test_loop(n_support, optimizer=None) -> 0-dim MSE tensor    test_regression.py:34
    .save_checkpoint(path) / .load_checkpoint(path)  keys 'gp', 'likelihood', 'net'   :99-110

The QMUL sampler (data/qmul_loader.get_batch) needs the dataset images; here it is injected
(`batch_fn(split) -> (inputs[P,19,3,100,100], labels[P,19])`) so the class runs on synthetic tasks too.
Single GP (C = 1), RBF kernel, noise LEARNED (init softplus(0) + 1e-4).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import configs, ops
from .gp import ExactGPHypers, RBF_KINDS, SPECTRAL_KINDS


class DKT(nn.Module):
    def __init__(self, backbone, kernel_type=None, batch_fn=None, num_mixtures=4, ard_num_dims=2916):
        super(DKT, self).__init__()
        self.kernel_type = configs.kernel_type if kernel_type is None else kernel_type
        if self.kernel_type not in RBF_KINDS + SPECTRAL_KINDS:
            raise ValueError("[ERROR] the kernel '" + str(self.kernel_type) +
                             "' is not supported for regression, use 'rbf' or 'spectral'.")
        # SpectralMixtureKernel(num_mixtures=4, ard_num_dims=2916), DKT_regression.py:122 (2916 = Conv3's feature size)
        self.num_mixtures, self.ard_num_dims = num_mixtures, ard_num_dims
        self.feature_extractor = backbone
        self.batch_fn = batch_fn
        self.jitter0 = 1e-6
        self.max_tries = 3
        self.get_model_likelihood_mll()

    def get_model_likelihood_mll(self, train_x=None, train_y=None):
        self.model = ExactGPHypers(1, self.kernel_type, fixed_noise=None, num_mixtures=self.num_mixtures,
                                   ard_num_dims=self.ard_num_dims)
        self.likelihood = self.model       # the Gaussian noise is stored with the GP hyper-parameters
        self.mse = nn.MSELoss()
        return self.model, self.likelihood, None

    def set_forward(self, x, is_feature=False):
        pass

    def set_forward_loss(self, x):
        pass

    @property
    def device(self):
        return self.model.mean_constant.device

    def _loss(self, z, labels):
        """loss = -logp / N for one task z:[N,D] or a batch [B,N,D] with labels [N] / [B,N] (mean over B)."""
        zb = z if z.dim() == 3 else z.unsqueeze(0)
        yb = labels.reshape(zb.shape[0], 1, zb.shape[1]).to(torch.float32)
        n = zb.shape[1]
        m = self.model
        cw = torch.full((1,), -1.0 / n, device=zb.device, dtype=torch.float32)
        if self.kernel_type in SPECTRAL_KINDS:
            e = ops.spectral_mixture_matrix(zb, m.mixture_weights, m.mixture_means, m.mixture_scales)
        else:
            e = ops.base_matrix(zb, self.kernel_type, m.lengthscale)
        obj, logp, alpha, info, jit = ops.mll_objective(e, yb, m.scale_times_variance(), m.mean, m.noise, cw,
                                                        self.jitter0, self.max_tries)
        return obj.mean(), dict(logp=logp, alpha=alpha, info=info, jitter=jit)

    def train_loop(self, epoch, optimizer, batch=None, batch_labels=None):
        if batch is None:
            batch, batch_labels = self.batch_fn("train")
        batch, batch_labels = batch.to(self.device), batch_labels.to(self.device)
        for inputs, labels in zip(batch, batch_labels):
            optimizer.zero_grad()
            z = self.feature_extractor(inputs)
            loss, _ = self._loss(z, labels)
            loss.backward()
            optimizer.step()
            # train-mode `predictions.mean` is the PRIOR mean (constant), DKT_regression.py:53,58
            prior_mean = self.model.mean.detach().expand(labels.shape[0])
            mse = self.mse(prior_mean, labels)
            if epoch % 10 == 0:
                print('[%d] - Loss: %.3f  MSE: %.3f noise: %.3f' % (epoch, loss.item(), mse.item(), self.model.noise.item()))

    @torch.no_grad()
    def predict(self, z_support, y_support, z_all, with_variance=False):
        """Condition on the support frames, posterior mean (and variance + noise) at z_all."""
        m = self.model
        sv, mean, noise, ls = m.scale_times_variance(), m.mean, m.noise, m.lengthscale
        zs = z_support.unsqueeze(0)
        spectral = self.kernel_type in SPECTRAL_KINDS
        if spectral:
            mix = (m.mixture_weights, m.mixture_means, m.mixture_scales)
            e = ops.smk(zs, None, *mix)[0]
        else:
            e = ops.gram(zs, None, ops.KERNEL_RBF, ls)
        out = ops.mll(e, y_support.reshape(1, 1, -1).to(torch.float32), sv, mean, noise, want_chol=with_variance,
                      jitter0=self.jitter0, max_tries=self.max_tries)
        ex = ops.smk(z_all.unsqueeze(0), zs, *mix)[0] if spectral else ops.gram(z_all.unsqueeze(0), zs, ops.KERNEL_RBF, ls)
        mu, _ = ops.predict(ex, out["alpha"], sv, mean, want_labels=False)
        if not with_variance:
            return mu[0, 0], None
        # k(x,x): 1 for RBF, sum of the mixture weights for the spectral mixture (tau = 0)
        kxx = float(m.mixture_weights.sum()) if spectral else 1.0
        exx = torch.full((1, z_all.shape[0]), kxx, device=z_all.device, dtype=torch.float32)
        var = ops.predict_var(ex, exx, out["chol"], sv, noise)
        return mu[0, 0], var[0, 0]

    def test_loop(self, n_support, optimizer=None, inputs=None, targets=None):
        if inputs is None:
            inputs, targets = self.batch_fn("test")
        support_ind = list(np.random.choice(list(range(inputs.shape[1])), replace=False, size=n_support))
        x_all = inputs.to(self.device)
        y_all = targets.to(self.device)
        x_support = x_all[:, support_ind]
        y_support = y_all[:, support_ind]
        n = np.random.randint(0, max(inputs.shape[0] - 1, 1))    # never the last person (DKT_regression.py:81)

        self.model.eval()
        self.feature_extractor.eval()
        with torch.no_grad():
            z_support = self.feature_extractor(x_support[n]).detach()
            z_query = self.feature_extractor(x_all[n]).detach()
            mean, var = self.predict(z_support, y_support[n], z_query, with_variance=True)
            lower, upper = mean - 2.0 * var.sqrt(), mean + 2.0 * var.sqrt()   # confidence_region(), unused
        return self.mse(mean, y_all[n])

    def save_checkpoint(self, checkpoint):
        torch.save({'gp': self.model.state_dict(), 'likelihood': {}, 'net': self.feature_extractor.state_dict()}, checkpoint)

    def load_checkpoint(self, checkpoint):
        """Own checkpoints and the reference's (DKT_regression.py:99-110: 'gp' = ExactGPLayer.state_dict() with GPyTorch's key
        names, 'likelihood' = GaussianLikelihood.state_dict(), 'net' = the backbone)."""
        ckpt = torch.load(checkpoint, map_location=self.device)
        gp = ckpt['gp']
        if any(k.startswith(("mean_module.", "covar_module.", "likelihood.noise_covar.")) for k in gp):
            ref = {"model.models.0." + k: v for k, v in gp.items()}
            for k, v in (ckpt.get('likelihood') or {}).items():           # 'noise_covar.raw_noise'
                ref.setdefault("model.models.0.likelihood." + k, v)
            if self.model.load_reference_state_dict(ref, strict_gp_keys=True) == 0:
                raise RuntimeError("reference regression checkpoint without known GP keys")
        else:
            self.model.load_state_dict(gp)
        self.feature_extractor.load_state_dict(ckpt['net'])
