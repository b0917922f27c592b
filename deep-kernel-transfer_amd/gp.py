"""Hyper-parameters of the C independent exact GPs, stored STACKED ([C] tensors) so one kernel launch
serves every class model of an episode.

Mirrors what the reference builds from GPyTorch objects (reference methods/DKT.py:58-71, 337-378;
methods/DKT_regression.py:25-37, 112-129):
  * ConstantMean            -> `mean_constant` [C], init 0, learned
  * ScaleKernel             -> `raw_outputscale` [C], outputscale = softplus(raw), init raw 0 -> ln 2
  * LinearKernel.variance   -> `raw_variance` [C]; cossim/bncossim: variance = 1.0 and frozen (DKT.py:366-370)
  * RBFKernel / MaternKernel(nu=2.5).lengthscale -> `raw_lengthscale` [C], lengthscale = softplus(raw), init ln 2
  * PolynomialKernel.offset -> `raw_offset` [C] (poli1, poli2), offset = softplus(raw), init ln 2
    (the reference builds one ExactGPLayer per class, DKT.py:63-66: every class model owns its base-kernel parameters)
  * SpectralMixtureKernel(num_mixtures=Q, ard_num_dims=D) (regression only, DKT_regression.py:121-122; NOT wrapped in a
                               ScaleKernel) -> `raw_mixture_weights` [Q], `raw_mixture_means` [Q,1,D],
                               `raw_mixture_scales` [Q,1,D] (GPyTorch's shapes), all softplus(raw), raw init 0
  * GaussianLikelihood      -> `raw_noise` [C], noise = softplus(raw) + 1e-4 (GreaterThan(1e-4));
                               classification: noise forced to 0.1 and frozen (DKT.py:346-347);
                               regression: learned, init softplus(0) + 1e-4.
`models[c]` exposes the attribute paths the reference's logging code reads
(`single_model.covar_module.base_kernel.lengthscale`, `.likelihood.noise`, `.covar_module.outputscale`,
DKT.py:148-154).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

NOISE_LOWER_BOUND = 1e-4
LINEAR_KINDS = ("linear", "cossim", "bncossim")
RBF_KINDS = ("rbf", "RBF")
MATERN_KINDS = ("matern",)
POLY_KINDS = ("poli1", "poli2")
SPECTRAL_KINDS = ("spectral",)
SUPPORTED_CLASSIFICATION = LINEAR_KINDS + RBF_KINDS + MATERN_KINDS + POLY_KINDS


def inv_softplus(y: float) -> float:
    return math.log(math.expm1(y)) if y < 30.0 else y


class ExactGPHypers(nn.Module):
    def __init__(self, n_models: int, kernel: str = "bncossim", fixed_noise: Optional[float] = 0.1,
                 num_mixtures: int = 4, ard_num_dims: Optional[int] = None):
        super().__init__()
        if kernel not in SUPPORTED_CLASSIFICATION + SPECTRAL_KINDS:
            raise ValueError("[ERROR] the kernel '" + str(kernel) + "' is not supported!")
        self.n_models = n_models
        self.kernel = kernel
        self.mean_constant = nn.Parameter(torch.zeros(n_models))
        if kernel in SPECTRAL_KINDS:      # no ScaleKernel around the spectral mixture: the base matrix enters K unscaled
            if n_models != 1 or not ard_num_dims:
                raise ValueError("the spectral kernel needs a single GP and ard_num_dims")
            self.register_parameter("raw_outputscale", None)
            self.raw_mixture_weights = nn.Parameter(torch.zeros(num_mixtures))
            self.raw_mixture_means = nn.Parameter(torch.zeros(num_mixtures, 1, ard_num_dims))
            self.raw_mixture_scales = nn.Parameter(torch.zeros(num_mixtures, 1, ard_num_dims))
        else:
            self.raw_outputscale = nn.Parameter(torch.zeros(n_models))
        if kernel in ("cossim", "bncossim"):
            # variance = 1.0, frozen
            self.raw_variance = nn.Parameter(torch.full((n_models,), inv_softplus(1.0)), requires_grad=False)
        elif kernel == "linear":
            self.raw_variance = nn.Parameter(torch.zeros(n_models))
        else:
            self.register_parameter("raw_variance", None)
        if kernel in RBF_KINDS + MATERN_KINDS:
            self.raw_lengthscale = nn.Parameter(torch.zeros(n_models))
        else:
            self.register_parameter("raw_lengthscale", None)
        if kernel in POLY_KINDS:          # PolynomialKernel.offset = softplus(raw_offset), init raw 0
            self.raw_offset = nn.Parameter(torch.zeros(n_models))
        else:
            self.register_parameter("raw_offset", None)
        if fixed_noise is not None:
            raw = inv_softplus(fixed_noise - NOISE_LOWER_BOUND)
            self.raw_noise = nn.Parameter(torch.full((n_models,), raw), requires_grad=False)
        else:
            self.raw_noise = nn.Parameter(torch.zeros(n_models))

    # ---- constrained values (differentiable torch ops on [C]-sized tensors) ----
    @property
    def outputscale(self) -> torch.Tensor:
        if self.raw_outputscale is None:
            return torch.ones(self.n_models, device=self.mean_constant.device, dtype=torch.float32)
        return F.softplus(self.raw_outputscale)

    @property
    def mixture_weights(self) -> torch.Tensor:
        return F.softplus(self.raw_mixture_weights)

    @property
    def mixture_means(self) -> torch.Tensor:
        return F.softplus(self.raw_mixture_means)

    @property
    def mixture_scales(self) -> torch.Tensor:
        return F.softplus(self.raw_mixture_scales)

    @property
    def variance(self) -> Optional[torch.Tensor]:
        return None if self.raw_variance is None else F.softplus(self.raw_variance)

    @property
    def lengthscale(self) -> Optional[torch.Tensor]:
        return None if self.raw_lengthscale is None else F.softplus(self.raw_lengthscale)

    @property
    def offset(self) -> Optional[torch.Tensor]:
        return None if self.raw_offset is None else F.softplus(self.raw_offset)

    @property
    def noise(self) -> torch.Tensor:
        return F.softplus(self.raw_noise) + NOISE_LOWER_BOUND

    @property
    def mean(self) -> torch.Tensor:
        return self.mean_constant

    def scale_times_variance(self) -> torch.Tensor:
        """sv[c] = outputscale[c] * variance: the factor of the base matrix E in K_c."""
        s = self.outputscale
        v = self.variance
        return s if v is None else s * v

    @property
    def models(self):
        return [_ModelView(self, c) for c in range(self.n_models)]

    def __len__(self):
        return self.n_models

    # ---- reference (GPyTorch IndependentModelList) checkpoint keys ----
    REFERENCE_KEYS = {                                  # key below `models.{c}.` -> parameter here (element c)
        "mean_module.constant": "mean_constant",
        "covar_module.raw_outputscale": "raw_outputscale",
        "likelihood.noise_covar.raw_noise": "raw_noise",
        "covar_module.base_kernel.raw_variance": "raw_variance",
        "covar_module.base_kernel.raw_lengthscale": "raw_lengthscale",
        "covar_module.base_kernel.raw_offset": "raw_offset",
    }

    # SpectralMixtureKernel sits directly under covar_module in the reference (DKT_regression.py:121-122: no ScaleKernel around it)
    REFERENCE_SPECTRAL_KEYS = {
        "covar_module.raw_mixture_weights": "raw_mixture_weights",
        "covar_module.raw_mixture_means": "raw_mixture_means",
        "covar_module.raw_mixture_scales": "raw_mixture_scales",
    }

    def load_reference_state_dict(self, state: dict, prefix: str = "model.", strict_gp_keys: bool = False) -> int:
        """Copy hyper-parameters out of a reference DKT `state_dict()` (GPyTorch 1.0.1 key names:
        `model.models.{c}.mean_module.constant`, `.covar_module.raw_outputscale`,
        `.covar_module.base_kernel.raw_variance|raw_lengthscale|raw_offset`, `.likelihood.noise_covar.raw_noise`), every class.
        Returns the number of tensors consumed.  Key names could not be diffed against GPyTorch here."""
        used, consumed = 0, set()
        with torch.no_grad():
            for c in range(self.n_models):
                base = "%smodels.%d." % (prefix, c)
                for key, name in self.REFERENCE_KEYS.items():
                    dst = getattr(self, name, None)
                    if base + key in state and dst is not None:
                        dst[c] = state[base + key].reshape(-1)[0].to(dst)
                        used += 1
                        consumed.add(base + key)
                for key, name in self.REFERENCE_SPECTRAL_KEYS.items():        # whole tensors (one model: the regression head)
                    dst = getattr(self, name, None)
                    if base + key in state and dst is not None:
                        src = state[base + key]
                        if src.numel() != dst.numel():
                            raise RuntimeError("%s: checkpoint shape %s does not fit %s" % (key, tuple(src.shape), tuple(dst.shape)))
                        dst.copy_(src.reshape(dst.shape).to(dst))
                        used += 1
                        consumed.add(base + key)
        if strict_gp_keys:
            # a GP tensor of the checkpoint that nothing here took would be dropped silently (e.g. a kernel this head was not built with)
            left = [k for k in state if k.startswith(prefix + "models.") and k not in consumed
                    and any(t in k for t in (".mean_module.", ".covar_module.", ".likelihood."))]
            if left:
                raise RuntimeError("reference checkpoint holds GP tensors this model has no parameter for: %s" % ", ".join(sorted(left)))
        return used

    @staticmethod
    def is_reference_state_dict(state: dict) -> bool:
        """A state dict written by the reference's DKT (GPyTorch module tree) rather than by this module."""
        return any(k.startswith("model.models.0.") for k in state)


class _Attr:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _ModelView:
    """Read-only view of class model c with GPyTorch-like attribute paths."""

    def __init__(self, hyp: ExactGPHypers, c: int):
        self._h, self._c = hyp, c

    @property
    def covar_module(self):
        h, c = self._h, self._c
        ls, var = h.lengthscale, h.variance
        return _Attr(outputscale=h.outputscale[c],
                     raw_outputscale=None if h.raw_outputscale is None else h.raw_outputscale[c],
                     base_kernel=_Attr(lengthscale=None if ls is None else ls[c:c + 1], variance=None if var is None else var[c:c + 1]))

    @property
    def likelihood(self):
        return _Attr(noise=self._h.noise[self._c:self._c + 1])

    @property
    def mean_module(self):
        return _Attr(constant=self._h.mean_constant[self._c:self._c + 1])
