"""-m gpu: the HIP hot path (through the C ABI) against the float64 oracle on identical seeded
inputs, against the committed golden vectors, and -- at BASELINE.json's full sizes -- through
size-independent properties.  Tolerances (north_star): marginal log likelihood 1e-4 relative,
identical predicted labels; gradients rel-L2 1e-3."""
import glob
import os
import sys

import numpy as np
import pytest
import torch

import dkt_amd
from dkt_amd import ops
from oracle import dkt_oracle as O
from oracle import dkt_oracle_torch as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MLL_RTOL = 1e-4
GRAD_RTOL = 1e-3


def dev_t(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_loaded_native_library(cuda, lib):
    assert lib.dkt_abi_version() == 7
    assert lib.dkt_device_cu_count() == 256, "expected an MI355X (256 CUs)"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


# ----------------------------------------------------------------------------------------------
# Gram build
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b,n,d", [(1, 1, 4), (2, 5, 7), (3, 19, 2916), (2, 25, 64), (2, 64, 32), (2, 65, 36),
                                   (2, 85, 512), (3, 105, 64), (2, 105, 1600), (1, 130, 50), (1, 420, 512)])
def test_gram_linear_symmetric(cuda, b, n, d):
    rng = np.random.default_rng(n * 1000 + d)
    z = rng.standard_normal((b, n, d)).astype(np.float32)
    e = ops.gram(dev_t(z, cuda)).cpu().numpy()
    ref = np.einsum("bnd,bmd->bnm", z.astype(np.float64), z.astype(np.float64))
    assert np.abs(e - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6
    assert (e == e.transpose(0, 2, 1)).all(), "Gram must be exactly symmetric"


@pytest.mark.parametrize("b,m,n,d", [(2, 75, 25, 64), (1, 75, 5, 512), (2, 300, 100, 512), (1, 19, 5, 2916), (1, 7, 3, 5)])
def test_gram_linear_cross(cuda, b, m, n, d):
    rng = np.random.default_rng(m * 100 + n)
    a = rng.standard_normal((b, m, d)).astype(np.float32)
    bm = rng.standard_normal((b, n, d)).astype(np.float32)
    bm[0, 0, :] = np.arange(d)          # asymmetric content: a transposed write would be caught
    e = ops.gram(dev_t(a, cuda), dev_t(bm, cuda)).cpu().numpy()
    ref = np.einsum("bmd,bnd->bmn", a.astype(np.float64), bm.astype(np.float64))
    assert np.abs(e - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6


# episode-resident kernels (one workgroup per episode; B >= 32, 64 < N <= 128, D % 4 == 0): every tile count NT = 5..8,
# ragged last slices (D % 32 != 0, D % 64 != 0), odd stage counts, both arithmetic routes and every pipeline variant
EP_SHAPES = [(64, 65, 36), (64, 75, 512), (70, 80, 1024), (64, 96, 100), (64, 105, 1600), (64, 105, 2916), (64, 112, 1056),
             (64, 128, 160), (96, 100, 32)]


@pytest.mark.parametrize("b,n,d", EP_SHAPES)
@pytest.mark.parametrize("split", ["1", "0"])
def test_gram_episode_resident_kernels(cuda, b, n, d, split, monkeypatch):
    monkeypatch.setenv("DKT_GRAM_SPLIT", split)
    g = torch.Generator(device=cuda).manual_seed(n * 7 + d)
    z = torch.randn(b, n, d, generator=g, device=cuda) * torch.exp(2.0 * torch.randn(b, n, d, generator=g, device=cuda))
    e = ops.gram(z)
    ref = torch.einsum("bnd,bmd->bnm", z.double(), z.double())
    mag = torch.einsum("bnd,bmd->bnm", z.double().abs(), z.double().abs())      # what fp32 rounding errors scale with
    err = ((e.double() - ref).abs() / mag).max().item()
    tol = max(1e-6, 6.0 * np.sqrt(d) * 2.0 ** -24)      # random-walk growth of fp32 accumulation error over D heavy-tailed terms
    assert err < tol, (err, tol)
    assert torch.equal(e, e.transpose(1, 2)), "Gram must be exactly symmetric"
    assert torch.equal(e, ops.gram(z)), "deterministic"
    # the tile-per-workgroup kernel (B below the episode-kernel threshold) agrees
    e_gen = ops.gram(z[:2].contiguous())
    gen_err = ((e_gen.double() - ref[:2]).abs() / mag[:2]).max().item()
    assert gen_err < tol, (gen_err, tol)


# a handful of episodes (B < 8: the reference's one-episode-per-step loop): one workgroup per 16 x 16 output tile, D over its waves (gram_sym_fewep_kernel);
# every tile count up to the 20-way episode, ragged D (D % 16 != 0, fewer 16-feature groups than waves), rows beyond N, both linear kinds
@pytest.mark.parametrize("b,n,d", [(1, 105, 1600), (3, 64, 20), (7, 33, 4), (1, 128, 516), (2, 130, 64), (1, 420, 512), (2, 447, 36), (5, 85, 512), (1, 105, 64)])
def test_gram_few_episodes_tile_kernel(cuda, b, n, d):
    g = torch.Generator(device=cuda).manual_seed(n * 11 + d + b)
    z = torch.randn(b, n, d, generator=g, device=cuda) * torch.exp(1.5 * torch.randn(b, n, d, generator=g, device=cuda))
    e = ops.gram(z)
    ref = torch.einsum("bnd,bmd->bnm", z.double(), z.double())
    mag = torch.einsum("bnd,bmd->bnm", z.double().abs(), z.double().abs())
    tol = max(1e-6, 6.0 * np.sqrt(d) * 2.0 ** -24)
    err = ((e.double() - ref).abs() / mag).max().item()
    assert err < tol, (err, tol)
    assert torch.equal(e, e.transpose(1, 2)), "Gram must be exactly symmetric"
    assert torch.equal(e, ops.gram(z)), "deterministic"
    zn = torch.nn.functional.normalize(z, dim=2)
    eu = ops.gram(zn, None, ops.KERNEL_LINEAR_UNIT)
    refu = torch.einsum("bnd,bmd->bnm", zn.double(), zn.double())
    assert (eu.double() - refu).abs().max().item() < 1e-6
    # the same episodes inside a batch the episode-resident / large-N kernels take
    if n <= 128:
        zb = z.repeat(-(-8 // b), 1, 1).contiguous()
        eb = ops.gram(zb)
        assert ((eb[:b].double() - ref).abs() / mag).max().item() < tol


@pytest.mark.parametrize("b,c", [(1, 5), (1, 1), (5, 20), (300, 5), (8192, 5), (1000, 32)])
def test_objective_and_hyper_gradient_reductions(cuda, b, c, monkeypatch):
    g = torch.Generator(device=cuda).manual_seed(b * 31 + c)
    logp = torch.randn(b, c, generator=g, device=cuda) * 100.0
    cw = torch.randn(c, generator=g, device=cuda)
    gobj = torch.randn(b, generator=g, device=cuda)
    dsv, dmean, dnoise = (torch.randn(b, c, generator=g, device=cuda) for _ in range(3))
    obj = ops.objective(logp, cw)
    ref = (logp.double() * cw.double().view(1, c)).sum(1)
    assert ((obj.double() - ref).abs() / (logp.double().abs() * cw.double().abs().view(1, c)).sum(1)).max().item() < 1e-6
    assert torch.equal(ops.objective(logp, None), ops.objective(logp, torch.ones(c, device=cuda)))
    shapes = (torch.Size([c]), torch.Size([c, 1]), torch.Size([1, c]))
    gs = ops.hyper_grads(gobj, cw, dsv, dmean, dnoise, shapes)
    for gx, dx, sh in zip(gs, (dsv, dmean, dnoise), shapes):
        assert gx.shape == sh
        r = cw.double() * (gobj.double().view(b, 1) * dx.double()).sum(0)
        mag = cw.double().abs() * (gobj.double().abs().view(b, 1) * dx.double().abs()).sum(0)
        assert ((gx.double().reshape(-1) - r).abs() / mag).max().item() < 1e-6 * max(1.0, np.log2(b))
    again = ops.hyper_grads(gobj, cw, dsv, dmean, dnoise, shapes)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(gs, again)), "fixed summation order"
    only = ops.hyper_grads(gobj, cw, None, dmean, None, shapes)
    assert only[0] is None and only[2] is None and torch.equal(only[1], gs[1])
    assert ops.hyper_grads(gobj, cw, None, None, None, shapes) == (None, None, None)
    monkeypatch.setenv("DKT_FUSED_REDUCTIONS", "0")                       # the tensor expressions they replace
    tw = ops.hyper_grads(gobj, cw, dsv, dmean, dnoise, shapes)
    for gx, tx in zip(gs, tw):
        assert tx.shape == gx.shape and ((gx - tx).abs().max() / tx.abs().max()).item() < 1e-5
    assert ((ops.objective(logp, cw) - obj).abs().max() / obj.abs().max()).item() < 1e-6


@pytest.mark.parametrize("b,d", [(1, 64), (5, 1600), (256, 36), (257, 1600), (2048, 1600), (3000, 512)])
def test_bn_param_gradient_sums(cuda, b, d, monkeypatch):
    """dkt_bn_param_grads_f32: the sums over the episodes of the per-episode dgamma / dbeta parts of the fused backward (one launch up to 256 episodes, row chunks +
    a fold beyond) against float64, bitwise repeatable, and against the tensor reductions it replaces; at the C ABI: a missing / short workspace is refused."""
    g = torch.Generator(device=cuda).manual_seed(b * 7 + d)
    dg = torch.randn(b, d, generator=g, device=cuda) * 3.0
    db = torch.randn(b, d, generator=g, device=cuda) + 0.5
    og, ob = ops.bn_param_grads(dg, db)
    for o, p_ in ((og, dg), (ob, db)):
        assert o.shape == (d,)
        err = (o.double() - p_.double().sum(0)).abs() / p_.double().abs().sum(0)
        assert err.max().item() < 1e-6 * max(1.0, np.log2(b))
    og2, ob2 = ops.bn_param_grads(dg, db)
    assert torch.equal(og, og2) and torch.equal(ob, ob2), "fixed summation order"
    monkeypatch.setenv("DKT_FUSED_REDUCTIONS", "0")
    tg, tb = ops.bn_param_grads(dg, db)
    assert ((tg - og).abs().max() / tg.abs().max()).item() < 1e-5 and ((tb - ob).abs().max() / tb.abs().max()).item() < 1e-5
    monkeypatch.delenv("DKT_FUSED_REDUCTIONS")
    lib = dkt_amd._lib.load()
    need = int(lib.dkt_bn_param_grads_workspace_bytes(b, d))
    assert (need == 0) == (b <= 256)
    if need:
        st = lib.dkt_bn_param_grads_f32(dg.data_ptr(), db.data_ptr(), og.data_ptr(), ob.data_ptr(), b, d, None, 0, None)
        assert st == -3                                                   # DKT_ERR_WORKSPACE
    assert lib.dkt_bn_param_grads_f32(dg.data_ptr(), db.data_ptr(), og.data_ptr(), ob.data_ptr(), b, d + 2, None, 0, None) == -1      # D % 4


@pytest.mark.parametrize("var", ["11", "12", "21", "22", "611", "612"])
def test_gram_split_pipeline_variants_agree_bitwise(cuda, var, monkeypatch):
    z = torch.nn.functional.normalize(torch.randn(64, 105, 1632, device=cuda, generator=torch.Generator(device=cuda).manual_seed(3)), dim=2)
    monkeypatch.setenv("DKT_GRAM_SPLIT_VAR", "11")
    ref = ops.gram(z)
    monkeypatch.setenv("DKT_GRAM_SPLIT_VAR", var)
    assert torch.equal(ops.gram(z), ref)


@pytest.mark.parametrize("b,n,d", EP_SHAPES)
@pytest.mark.parametrize("split", ["1", "0"])
def test_gram_bwd_episode_resident_kernels(cuda, b, n, d, split, monkeypatch):
    monkeypatch.setenv("DKT_GRAM_SPLIT", split)
    monkeypatch.setenv("DKT_GRAM_BWD_SPLIT_MIND", "32")              # force the split kernel also at small D
    g = torch.Generator(device=cuda).manual_seed(n * 11 + d)
    z = torch.randn(b, n, d, generator=g, device=cuda)
    w = torch.randn(b, n, n, generator=g, device=cuda) * torch.exp(2.0 * torch.randn(b, n, n, generator=g, device=cuda))
    sc = torch.rand(b, generator=g, device=cuda) + 0.5
    dz = ops.gram_bwd(w, z, sc)
    ws = (w + w.transpose(1, 2)).double() * sc.double().view(-1, 1, 1)
    ref = ws @ z.double()
    mag = ws.abs() @ z.double().abs()
    assert ((dz.double() - ref).abs() / mag).max().item() < 2e-6
    assert torch.equal(dz, ops.gram_bwd(w, z, sc)), "deterministic"
    dz1 = ops.gram_bwd(w, z, None)                                  # no per-episode scale
    assert ((dz1.double() - (w + w.transpose(1, 2)).double() @ z.double()).abs() / (mag / sc.double().view(-1, 1, 1))).max().item() < 2e-6


@pytest.mark.parametrize("var", ["11", "12", "22"])
def test_gram_bwd_split_pipeline_variants_agree_bitwise(cuda, var, monkeypatch):
    g = torch.Generator(device=cuda).manual_seed(5)
    z = torch.randn(64, 105, 1088, generator=g, device=cuda)          # 17 slabs of 64: odd slab count
    w = torch.randn(64, 105, 105, generator=g, device=cuda)
    monkeypatch.setenv("DKT_GRAM_BWD_SPLIT_VAR", "11")
    ref = ops.gram_bwd(w, z, None)
    monkeypatch.setenv("DKT_GRAM_BWD_SPLIT_VAR", var)
    assert torch.equal(ops.gram_bwd(w, z, None), ref)


# ---- unit-norm rows (DKT_KERNEL_LINEAR_UNIT / DKT_GRAM_UNIT_ROWS): the scaled 2-way f16 split ----
def _unit_rows(b, n, d, seed, cuda, kind):
    g = torch.Generator(device=cuda).manual_seed(seed)
    z = torch.randn(b, n, d, generator=g, device=cuda)
    if kind == "heavy":          # a few dominant features per row, many tiny ones (exercises the f16 subnormal range of m)
        z = z * torch.exp(3.0 * torch.randn(b, n, d, generator=g, device=cuda))
    z = torch.nn.functional.normalize(z, dim=2)
    if kind == "onehot":         # rows with a single +-1 entry, exact zeros elsewhere, mixed with ordinary rows
        z[:, ::3] = 0.0
        idx = torch.randint(0, d, (b, (n + 2) // 3), generator=g, device=cuda)
        z[:, ::3].scatter_(2, idx.unsqueeze(2), -1.0)
    return z.contiguous()


@pytest.mark.parametrize("b,n,d", EP_SHAPES)
@pytest.mark.parametrize("kind", ["plain", "heavy", "onehot"])
def test_gram_unit_rows_f16_split(cuda, b, n, d, kind):
    z = _unit_rows(b, n, d, n * 13 + d, cuda, kind)
    e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
    ref = torch.einsum("bnd,bmd->bnm", z.double(), z.double())
    mag = torch.einsum("bnd,bmd->bnm", z.double().abs(), z.double().abs())
    tol = max(1e-6, 6.0 * np.sqrt(d) * 2.0 ** -24)
    # relative to sum_k |a_k||b_k| like the fp32 kernels, plus the absolute floor of the f16 range: the low piece of an element
    # below 2^-18 is flushed, at most 2^-30 |b_k| per product
    excess = ((e.double() - ref).abs() - tol * mag).max().item()
    assert excess < 2.0 ** -30 * np.sqrt(d), (excess, tol)
    assert (e.double() - ref).abs().max().item() < 1.5e-6                  # |E - E64| on cosine similarities
    assert torch.equal(e, e.transpose(1, 2)) and torch.equal(e, ops.gram(z, None, ops.KERNEL_LINEAR_UNIT))
    # same answer (to fp32 accuracy) as the range-agnostic bf16 split and as the generic tile kernel below the B threshold
    assert (e - ops.gram(z)).abs().max().item() < 1.5e-6
    # (below the B threshold the promise is simply unused: sequential fp32 MFMA chain, a few 1e-7 further from float64)
    assert (ops.gram(z[:2].contiguous(), None, ops.KERNEL_LINEAR_UNIT) - e[:2]).abs().max().item() < 4e-6


@pytest.mark.parametrize("var", ["211", "212", "2611", "26113", "26122", "2223", "2213", "22232"])
def test_gram_unit_pipeline_variants_agree_bitwise(cuda, var, monkeypatch):
    z = _unit_rows(64, 105, 1632, 3, cuda, "plain")
    monkeypatch.setenv("DKT_GRAM_UNIT_VAR", "2223")
    ref = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
    monkeypatch.setenv("DKT_GRAM_UNIT_VAR", var)
    assert torch.equal(ops.gram(z, None, ops.KERNEL_LINEAR_UNIT), ref)
    monkeypatch.delenv("DKT_GRAM_UNIT_VAR")
    assert torch.equal(ops.gram(z, None, ops.KERNEL_LINEAR_UNIT), ref), "the default variant"


@pytest.mark.parametrize("d", [64, 512, 1632])
def test_gram_bwd_unit_pipeline_variants_agree_bitwise(cuda, d, monkeypatch):
    """The staging variants of the f16-split Gram backward (DKT_GRAM_BWD_UNIT_VAR: LDS images / prefetch depth / store policy; the default picks by D) run the
    same arithmetic in the same order."""
    z = _unit_rows(64, 105, d, 5, cuda, "plain")
    g = torch.Generator(device=cuda).manual_seed(d)
    w = torch.randn(64, 105, 105, generator=g, device=cuda)
    ref = ops.gram_bwd(w, z, None, unit_rows=True)
    for var in ("211", "221", "212", "222", "1222", "2222", "3222"):
        monkeypatch.setenv("DKT_GRAM_BWD_UNIT_VAR", var)
        assert torch.equal(ops.gram_bwd(w, z, None, unit_rows=True), ref), var


def test_gram_unit_rows_promise_violation_is_loud(cuda):
    z = _unit_rows(64, 105, 256, 9, cuda, "plain")
    z[5, 17, 33] = 2.5                                   # |z| > 1.999: overflows the scaled f16 piece
    e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
    assert not torch.isfinite(e[5, 17]).all(), "a violated unit-row promise must poison the episode, not pass silently"
    assert torch.isfinite(e[4]).all() and torch.isfinite(e[6]).all()


@pytest.mark.parametrize("b,n,d", EP_SHAPES)
@pytest.mark.parametrize("kind", ["plain", "heavy"])
def test_gram_bwd_unit_rows_f16_split(cuda, b, n, d, kind, monkeypatch):
    monkeypatch.setenv("DKT_GRAM_BWD_UNIT_MIND", "32")               # force the split kernel also at D < 64
    z = _unit_rows(b, n, d, n * 17 + d, cuda, kind)
    g = torch.Generator(device=cuda).manual_seed(n + d)
    # gradient-like W: per-row AND per-entry dynamic range, one all-zero row, one tiny row
    w = torch.randn(b, n, n, generator=g, device=cuda) * torch.exp(2.0 * torch.randn(b, n, 1, generator=g, device=cuda)) \
        * torch.exp(1.5 * torch.randn(b, n, n, generator=g, device=cuda))
    sc = torch.rand(b, generator=g, device=cuda) + 0.5
    dz = ops.gram_bwd(w, z, sc, unit_rows=True)
    ws = (w + w.transpose(1, 2)).double() * sc.double().view(-1, 1, 1)
    ref = ws @ z.double()
    mag = ws.abs() @ z.double().abs()
    # 2e-6 relative to sum_j |W'_ij||z_jd| like the fp32 kernels + the f16 range floor (entries below 2^-17 of their row's
    # maximum / features below 2^-18 lose their low piece: <= 2^-29 rowmax per term)
    rowmax = ws.abs().amax(2, keepdim=True)
    excess = ((dz.double() - ref).abs() - 2e-6 * mag - 2.0 ** -29 * np.sqrt(n) * rowmax).max().item()
    assert excess < 0, excess
    assert torch.equal(dz, ops.gram_bwd(w, z, sc, unit_rows=True)), "deterministic"
    w0 = torch.zeros_like(w)
    w0[:, 3, 7] = 1e-30
    dz0 = ops.gram_bwd(w0, z, None, unit_rows=True)      # all-zero rows and a denormal-scale row: finite, exact zeros elsewhere
    assert torch.isfinite(dz0).all() and (dz0[:, 0] == 0).all()
    assert (dz0[:, 3].double() - 1e-30 * z[:, 7].double()).abs().max().item() < 1e-36


@pytest.mark.parametrize("var", ["222", "221", "212", "211"])
def test_gram_bwd_unit_pipeline_variants_agree_bitwise(cuda, var, monkeypatch):
    z = _unit_rows(64, 105, 1088, 5, cuda, "plain")
    w = torch.randn(64, 105, 105, device=cuda, generator=torch.Generator(device=cuda).manual_seed(6))
    monkeypatch.setenv("DKT_GRAM_BWD_UNIT_VAR", "222")
    ref = ops.gram_bwd(w, z, None, unit_rows=True)
    monkeypatch.setenv("DKT_GRAM_BWD_UNIT_VAR", var)
    assert torch.equal(ops.gram_bwd(w, z, None, unit_rows=True), ref)


@pytest.mark.parametrize("n,d,ls,shift", [(19, 2916, 30.0, 0.4), (5, 2916, 20.0, 0.4), (105, 64, 1.3, 0.0), (70, 33, 0.9, 5.0)])
def test_gram_rbf(cuda, n, d, ls, shift):
    rng = np.random.default_rng(n + d)
    z = (np.abs(rng.standard_normal((2, n, d))) * 0.5 + shift).astype(np.float32)   # ReLU-like, common offset
    lst = dev_t([ls], cuda)
    e = ops.gram(dev_t(z, cuda), None, ops.KERNEL_RBF, lst).cpu().numpy()
    for i in range(2):
        ref = O.gram_rbf(z[i].astype(np.float64), None, ls)
        assert np.abs(e[i] - ref).max() < 2e-5, np.abs(e[i] - ref).max()
    assert (np.diagonal(e, axis1=1, axis2=2) == 1.0).all()
    # cross
    q = (np.abs(rng.standard_normal((2, 11, d))) * 0.5 + shift).astype(np.float32)
    ex = ops.gram(dev_t(q, cuda), dev_t(z, cuda), ops.KERNEL_RBF, lst).cpu().numpy()
    for i in range(2):
        assert np.abs(ex[i] - O.gram_rbf(q[i].astype(np.float64), z[i].astype(np.float64), ls)).max() < 2e-5


# ----------------------------------------------------------------------------------------------
# marginal log likelihood: Cholesky, log-det, mean cache, gradients
# ----------------------------------------------------------------------------------------------
def _episode_case(c, per, d, seed, corr=0, b=2):
    n = c * per
    z = O.synthetic_features(b, n, d, seed, corr)
    hyp = O.perturbed_hypers(c, seed + 1)
    return z, hyp, n


@pytest.mark.parametrize("c,per,d,corr", [(1, 1, 8, 0), (2, 1, 8, 0), (5, 1, 64, 0), (5, 5, 64, 0), (5, 17, 512, 0),
                                          (3, 5, 16, 0), (4, 4, 16, 0), (1, 17, 16, 0), (3, 37, 24, 0), (4, 28, 24, 0), (7, 18, 24, 0), (1, 127, 24, 0),
                                          (5, 21, 64, 0), (5, 21, 1600, 5), (5, 26, 40, 0), (5, 38, 32, 0),
                                          (4, 50, 64, 0), (20, 16, 512, 0), (20, 21, 512, 20),
                                          # block-column boundaries of the two-pivots-per-barrier sweep (even / odd tails)
                                          (1, 3, 8, 0), (1, 31, 16, 0), (1, 32, 16, 0), (1, 33, 16, 0), (2, 32, 16, 0), (1, 65, 16, 0),
                                          (2, 48, 16, 0), (1, 97, 16, 0), (1, 113, 16, 0), (1, 126, 16, 0)])
@pytest.mark.parametrize("path", ["default", "h2e", "h2e_grow", "f32mfma", "reg", "generic"])
def test_mll_forward_and_gradients_vs_oracle(cuda, c, per, d, corr, path, monkeypatch):
    """default: the wave-per-matrix kernel on the f16 matrix pipe (scaled 2-way splits) for N <= 127 -- the training call, i.e. without
    the Cholesky output, at a small batch --, the tile-array / blocked / generic kernels above;  h2e: the wave-per-episode kernel that
    serves the same call from 1024 episodes per launch (DKT_MLL_H2E_MINB=1 selects it for any batch; N <= 111);  h2e_grow: the same with
    DKT_MLL_P2_GUARD=-1 -- the f16 scale of M = R^-T then starts from the first diagonal tile alone, so that every matrix walks through
    the grow-on-demand re-scaling of the stored tiles that production meets only when a tile row jumps above the diagonal;  f32mfma: the exact-fp32
    MFMA twin (DKT_MLL_FORCE_F32MFMA, also what serves want_chol);  reg: the register-sweep twin (DKT_MLL_FORCE_REG);  generic: the
    generic LDS / global kernel for every N."""
    force_generic, force_reg, force_f32 = path == "generic", path == "reg", path == "f32mfma"
    want_chol = path not in ("default", "h2e", "h2e_grow")
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1" if path.startswith("h2e") else "1000000000")
    monkeypatch.setenv("DKT_MLL_P2_GUARD", "-1" if path == "h2e_grow" else "1")
    z, hyp, n = _episode_case(c, per, d, 17 + n_hash(c, per, d), corr)
    y = O.one_vs_rest_targets(c, per)
    sv = hyp.outputscale
    cw = np.full(c, -1.0 / (c * n))
    e_dev = ops.gram(dev_t(z, cuda))
    out = ops.mll(e_dev, dev_t(y, cuda), dev_t(sv, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda),
                  want_grad=True, want_chol=want_chol, cls_weight=dev_t(cw, cuda), force_generic=force_generic, force_reg=force_reg,
                  force_f32mfma=force_f32)
    torch.cuda.synchronize()
    assert int(out["info"].abs().max().item()) == 0
    assert float(out["jitter"].abs().max().item()) == 0.0
    for i in range(z.shape[0]):
        e = O.gram_linear(z[i])
        res = O.mll_terms(e, y, sv, hyp.mean, hyp.noise)
        logp = out["logp"][i].cpu().numpy()
        assert np.abs(logp - res.logp).max() / np.abs(res.logp).max() < MLL_RTOL
        assert np.abs((logp - res.logp) / res.logp).max() < MLL_RTOL
        assert rel_l2(out["alpha"][i].cpu().numpy(), res.alpha) < 5e-4
        if want_chol:
            assert rel_l2(out["chol"][i].cpu().numpy(), res.chol) < 5e-5
        w_e, dsv, dmean, dnoise = O.mll_grads(e, res, sv, hyp.noise, np.ones(c))
        w_ref, _, _, _ = O.mll_grads(e, res, sv, hyp.noise, cw)
        wk = out["w"][i].cpu().numpy()
        assert rel_l2(wk, w_ref) < GRAD_RTOL
        assert (wk == wk.T).all()
        # dsv comes from scalar identities (tr K^-1, alpha.alpha, r.alpha) in the register kernel: allow an absolute
        # floor for the degenerate E = 0 case (single BN'ed row) where the true derivative is exactly 0
        assert np.linalg.norm(out["dsv"][i].cpu().numpy() - dsv) < GRAD_RTOL * np.linalg.norm(dsv) + 1e-4
        assert rel_l2(out["dmean"][i].cpu().numpy(), dmean) < GRAD_RTOL
        assert rel_l2(out["dnoise"][i].cpu().numpy(), dnoise) < GRAD_RTOL


@pytest.mark.parametrize("c,per", [(5, 21), (4, 26), (1, 97), (3, 37), (2, 63)])
def test_mll_full_occupancy_is_race_free(cuda, c, per):
    """2048 episodes = 4 workgroups per CU, twice over: every LDS hand-off of the sweep (pivot pairs, the odd leftover pivot,
    block-column prologues, gradient chunks) runs under contention.  Three launches must agree bitwise and sampled
    episodes must match the oracle (a hand-off race shows up as run-to-run differences in a few dozen episodes)."""
    n, d, b = c * per, 64, 2048
    g = torch.Generator(device=cuda).manual_seed(n)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=cuda), dim=2)
    e = ops.gram(z)
    hyp = O.perturbed_hypers(c, 3)
    y = O.one_vs_rest_targets(c, per)
    cw = np.full(c, -1.0 / (c * n))
    args = (e, dev_t(y, cuda), dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda))
    runs = [ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda)) for _ in range(3)]
    for k in ("logp", "alpha", "w", "dsv", "dmean", "dnoise"):
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
    assert int(runs[0]["info"].abs().max().item()) == 0
    e_np = e.cpu().numpy().astype(np.float64)
    for i in (0, 1, 511, 1024, 1500, 2047):
        res = O.mll_terms(e_np[i], y, hyp.outputscale, hyp.mean, hyp.noise)
        assert np.abs((runs[0]["logp"][i].cpu().numpy() - res.logp) / res.logp).max() < MLL_RTOL
        w_ref, _, _, _ = O.mll_grads(e_np[i], res, hyp.outputscale, hyp.noise, cw)
        assert rel_l2(runs[0]["w"][i].cpu().numpy(), w_ref) < GRAD_RTOL
    # every episode against the generic twin (different algorithmic path through LDS)
    gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    assert ((runs[0]["logp"] - gen["logp"]).abs() / gen["logp"].abs()).max().item() < 1e-5
    assert ((runs[0]["w"] - gen["w"]).flatten(1).norm(dim=1) / gen["w"].flatten(1).norm(dim=1)).max().item() < 1e-4


def test_h2_tile_primitives(cuda):
    """The f16-split tile primitives of csrc/dkt_h2_tiles.h (what dkt_mll_h2.hip is built on): the scaled 2-way split (h, m) of an
    accumulator-layout tile (22 significand bits, exact re-join), X^T Y as three v_mfma_f32_16x16x16_f16 plane products on the
    packed tiles, and the transposition of a split tile through the matrix pipe."""
    import ctypes
    import dkt_amd
    lib = dkt_amd._lib.load_diag()
    lib.dkt_diag_h2_primitives.restype = ctypes.c_int
    lib.dkt_diag_h2_primitives.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    rng = np.random.default_rng(7)
    for sx, sy, mag in ((2.0 ** 15, 2.0 ** 15, 1.0), (2.0 ** 9, 2.0 ** 13, 1.0), (2.0 ** 15, 2.0 ** 12, 1e-3)):
        xt = (rng.uniform(-1, 1, (16, 16)) * mag).astype(np.float32)         # asymmetric operands
        yt = rng.uniform(-1, 1, (16, 16)).astype(np.float32)
        xt[3, 5] = 0.0
        out = torch.zeros(5 * 256, device=cuda)
        assert lib.dkt_diag_h2_primitives(dev_t(np.concatenate([xt.ravel(), yt.ravel()]), cuda).data_ptr(), out.data_ptr(), sx, sy, None) == 0
        torch.cuda.synchronize()
        o = out.cpu().numpy().astype(np.float64).reshape(5, 16, 16)
        x64, y64 = xt.astype(np.float64), yt.astype(np.float64)
        h_ref = (xt * np.float32(sx)).astype(np.float16)
        m_ref = (xt * np.float32(sx) - h_ref.astype(np.float32)).astype(np.float16)
        np.testing.assert_array_equal(o[3], h_ref.astype(np.float64))
        np.testing.assert_array_equal(o[4], m_ref.astype(np.float64))
        np.testing.assert_array_equal(o[0], o[3] + o[4])                     # the join is exact
        assert np.abs(o[0] - sx * x64).max() <= 2.0 ** -21 * np.abs(sx * x64).max()
        assert np.abs(o[1] - sx * sy * (x64.T @ y64)).max() < 3e-6 * sx * sy * (np.abs(x64).T @ np.abs(y64)).max()
        np.testing.assert_array_equal(o[2], -o[0].T)                         # the transposition is exact


def test_lane_primitives(cuda):
    """The hardware idioms the MFMA marginal-likelihood kernel is built on (csrc/dkt_mll_mfma.hip): DPP row_newbcast, the
    v_permlane32/16_swap row spread, X^T Y straight from accumulator registers, and the DPP-fused FMA of the sweep."""
    import ctypes
    import dkt_amd
    lib = dkt_amd._lib.load_diag()
    lib.dkt_diag_lane_primitives.restype = ctypes.c_int
    lib.dkt_diag_lane_primitives.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(5)
    xt, yt = rng.standard_normal((16, 16)).astype(np.float32), rng.standard_normal((16, 16)).astype(np.float32)
    lane = np.arange(64)
    g, c = lane >> 4, lane & 15
    acc = lambda t: np.stack([t[4 * g + q, c] for q in range(4)])          # accumulator layout [4][64]
    inp = np.concatenate([acc(xt).ravel(), acc(yt).ravel()]).astype(np.float32)
    out = torch.zeros(640, device=cuda)
    assert lib.dkt_diag_lane_primitives(dev_t(inp, cuda).data_ptr(), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    x, y = acc(xt), acc(yt)
    np.testing.assert_array_equal(o[:64], x[0][(lane & ~15) | 3])
    for k in range(4):                                                     # rows 1, 5, 9, 13 of X in every row group
        np.testing.assert_array_equal(o[64 + 64 * k:128 + 64 * k], xt[4 * k + 1, c])
    np.testing.assert_allclose(o[320:576].reshape(4, 64), acc(xt.T.astype(np.float64) @ yt.astype(np.float64)), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(o[576:640], x[2] + x[2][(lane & ~15) | 5] * y[0], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("c,per,d,corr", [(5, 21, 64, 0), (5, 21, 1600, 5), (1, 104, 32, 0), (3, 37, 24, 0), (20, 5, 64, 0), (7, 9, 32, 0),
                                          (6, 4, 16, 0), (2, 8, 16, 0)])
def test_mll_mfma_kernel_against_register_twin(cuda, c, per, d, corr, monkeypatch):
    """The default wave-per-matrix MFMA kernel and the register-sweep twin are different algorithms (upper blocked
    factorisation on the matrix pipe vs a right-looking sweep on the VALU): every output must agree to rounding, with and
    without the gradient / Cholesky outputs (different template instantiations), C > 5 running the classes in rounds."""
    monkeypatch.setenv("DKT_TWINS", "force")             # every instantiation of the MFMA kernel lives in the twins library (the product keeps the Cholesky-output ones for N <= 31)
    z, hyp, n = _episode_case(c, per, d, 5 + n_hash(c, per, d), corr, b=3)
    y = dev_t(O.one_vs_rest_targets(c, per), cuda)
    cw = dev_t(np.full(c, -1.0 / (c * n)), cuda)
    e = ops.gram(dev_t(z, cuda))
    args = (e, y, dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda))
    a = ops.mll(*args, want_grad=True, want_chol=True, cls_weight=cw)                       # Cholesky output: the exact-fp32 MFMA kernel
    r = ops.mll(*args, want_grad=True, want_chol=True, cls_weight=cw, force_reg=True)
    torch.cuda.synchronize()
    for i in range(3):
        res = O.mll_terms(O.gram_linear(z[i]), O.one_vs_rest_targets(c, per), hyp.outputscale, hyp.mean, hyp.noise)
        assert np.abs((a["logp"][i].cpu().numpy() - res.logp) / res.logp).max() < MLL_RTOL
    for key in ("logp", "alpha", "chol", "w", "dsv", "dmean", "dnoise"):
        assert rel_l2(a[key].cpu().numpy(), r[key].cpu().numpy()) < 1e-4, key
    assert int(a["info"].abs().max().item()) == 0
    g = ops.mll(*args, want_grad=True, cls_weight=cw, force_f32mfma=True)
    f = ops.mll(*args, force_f32mfma=True)
    ch = ops.mll(*args, want_chol=True)
    for key in ("logp", "alpha"):                       # the four template instantiations agree to rounding
        for other in (g, f, ch):
            assert rel_l2(a[key].cpu().numpy(), other[key].cpu().numpy()) < 2e-6, key
    assert rel_l2(a["w"].cpu().numpy(), g["w"].cpu().numpy()) < 2e-6 and rel_l2(a["chol"].cpu().numpy(), ch["chol"].cpu().numpy()) < 2e-6
    # the default training / test-time calls (no Cholesky output): the f16-split kernels (wave per matrix; wave per episode), against
    # the exact-fp32 twin
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1")
    he = ops.mll(*args, want_grad=True, cls_weight=cw)
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1000000000")
    h = ops.mll(*args, want_grad=True, cls_weight=cw)
    hf = ops.mll(*args)
    torch.cuda.synchronize()
    assert int(he["info"].abs().max().item()) == 0
    for key, tol in (("logp", 2e-5), ("alpha", 5e-5), ("w", 5e-5), ("dsv", 1e-3), ("dmean", 5e-5), ("dnoise", 5e-5)):
        assert rel_l2(he[key].cpu().numpy(), a[key].cpu().numpy()) < tol, ("h2e", key, rel_l2(he[key].cpu().numpy(), a[key].cpu().numpy()))
    assert (he["w"].cpu().numpy() == he["w"].cpu().numpy().transpose(0, 2, 1)).all()
    assert int(h["info"].abs().max().item()) == 0 and int(hf["info"].abs().max().item()) == 0
    for key, tol in (("logp", 2e-5), ("alpha", 5e-5), ("w", 5e-5), ("dsv", 1e-3), ("dmean", 5e-5), ("dnoise", 5e-5)):      # dsv: scalar identities with cancellation
        assert rel_l2(h[key].cpu().numpy(), a[key].cpu().numpy()) < tol, (key, rel_l2(h[key].cpu().numpy(), a[key].cpu().numpy()))
    for key in ("logp", "alpha"):
        assert rel_l2(hf[key].cpu().numpy(), h[key].cpu().numpy()) < 2e-6, key
    wk = h["w"].cpu().numpy()
    assert (wk == wk.transpose(0, 2, 1)).all()


@pytest.mark.parametrize("scale", [1.0, 90.0, 3e-3, 4100.0])
def test_mll_large_and_small_magnitude_base_matrices(cuda, scale, monkeypatch):
    """Polynomial / un-normalised linear kernels hand the marginal-likelihood kernel base matrices whose diagonal is far from 1
    (DKT.py:352-365).  The MFMA kernel factors K / 4^m with every pivot <= 1 (exact power-of-two scaling): log-likelihood,
    alpha, the Cholesky factor and all gradients must stay at fp32 accuracy for any magnitude.  At scale 4100 the condition
    number (~4e6) is beyond what an fp32 factorisation resolves to 1e-4 (the register-sweep twin measures 5e-3 on the gradient
    scalars there, this kernel 7e-2: both form the inverse factor explicitly): only finiteness and a 1e-2 bound on the
    log-likelihood are required."""
    c, per, d = 5, 12, 24
    z, hyp, n = _episode_case(c, per, d, 91, 3, b=2)
    y = O.one_vs_rest_targets(c, per)
    cw = np.full(c, -1.0 / (c * n))
    e64 = np.stack([scale * (O.gram_linear(z[i]) + 0.3) for i in range(2)])
    args = (dev_t(e64, cuda), dev_t(y, cuda), dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda))
    out = ops.mll(*args, want_grad=True, want_chol=True, cls_weight=dev_t(cw, cuda))
    twin = ops.mll(*args, want_grad=True, want_chol=True, cls_weight=dev_t(cw, cuda), force_reg=True)
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1000000000")
    h2 = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))          # the default training call: f16-split kernel, wave per matrix, + the kappa-aware fix-up launch
    h2_raw = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), no_kappa_guard=True)      # the split kernel alone
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1")
    h2e = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))         # ... wave per episode
    h2e_raw = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), no_kappa_guard=True)
    torch.cuda.synchronize()
    # the a-priori bound 1 + sv trace(E) / noise (5e3) sends scale 90 (4.9e4) and 4100 to the exact-fp32 generic kernel and leaves scale 1 (540) and 3e-3 to the splits
    gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    for o, raw in ((h2, h2_raw), (h2e, h2e_raw)):
        assert torch.equal(o["w"], gen["w"] if scale > 10.0 else raw["w"]) and torch.equal(o["logp"], gen["logp"] if scale > 10.0 else raw["logp"])
    assert int(out["info"].abs().max().item()) == 0 and int(h2["info"].abs().max().item()) == 0 and int(h2e["info"].abs().max().item()) == 0
    hard = scale > 1000.0
    for i in range(2):
        e = e64[i].astype(np.float32).astype(np.float64)
        res = O.mll_terms(e, y, hyp.outputscale, hyp.mean, hyp.noise)
        w_ref, _, _, _ = O.mll_grads(e, res, hyp.outputscale, hyp.noise, cw)
        _, dsv1, dmean1, dnoise1 = O.mll_grads(e, res, hyp.outputscale, hyp.noise, np.ones(c))
        errs = {}
        for name, o in (("mfma", out), ("reg", twin), ("h2", h2), ("h2e", h2e), ("h2_raw", h2_raw), ("h2e_raw", h2e_raw)):
            errs[name] = dict(logp=np.abs((o["logp"][i].cpu().numpy() - res.logp) / res.logp).max(),
                              alpha=rel_l2(o["alpha"][i].cpu().numpy(), res.alpha),
                              chol=rel_l2(o["chol"][i].cpu().numpy(), res.chol) if o["chol"] is not None else 0.0,
                              w=rel_l2(o["w"][i].cpu().numpy(), w_ref), dsv=rel_l2(o["dsv"][i].cpu().numpy(), dsv1),
                              dmean=rel_l2(o["dmean"][i].cpu().numpy(), dmean1), dnoise=rel_l2(o["dnoise"][i].cpu().numpy(), dnoise1))
        tol = dict(logp=MLL_RTOL, alpha=5e-4, chol=5e-5, w=GRAD_RTOL, dsv=GRAD_RTOL, dmean=GRAD_RTOL, dnoise=GRAD_RTOL)
        if hard:
            for name in ("mfma", "h2", "h2e", "h2_raw", "h2e_raw"):
                assert errs[name]["logp"] < 1e-2 and all(np.isfinite(v) for v in errs[name].values()), errs
            continue
        # scale 90: cond(K_c) = 0.7 ... 1.6e4, 20 x the worst case of the reference's episodes (unit-norm features, noise 0.1: <= 730).  An
        # fp32 LAPACK factorisation resolves the log-likelihood to 1 ... 4e-5 there (the exact-fp32 kernel: 8e-5), the 22-bit splits of the
        # f16 kernels to 2 ... 3 x that (lane-level model, tools/mll_mfma_model.py): the RAW split kernels get 3 x every tolerance at this scale; the
        # DEFAULT dispatch (round 6: kappa-aware, no host read-back) holds 1 x -- it hands these matrices to the exact kernel.
        split_tol = {k: 3.0 * t for k, t in tol.items()} if scale > 10.0 else tol
        for k, t in tol.items():
            assert errs["mfma"][k] < t, (k, errs["mfma"][k], errs["reg"][k])
            assert errs["h2"][k] < t, (k, errs["h2"][k], errs["mfma"][k])
            assert errs["h2e"][k] < t, (k, errs["h2e"][k], errs["mfma"][k])
            assert errs["h2_raw"][k] < split_tol[k], (k, errs["h2_raw"][k], errs["mfma"][k])
            assert errs["h2e_raw"][k] < split_tol[k], (k, errs["h2e_raw"][k], errs["mfma"][k])


def test_small_batch_training_step_replayed_from_a_hipgraph_is_bitwise_the_eager_step(cuda):
    """The launch-bound end of SURVEY 8d's batch sizes (B = 1: what methods/DKT.py:117 issues): Gram -> marginal likelihood (+ its kappa-aware fix-up launch) -> Gram backward
    and the torch glue around them captured ONCE into a hipGraph and replayed (bench.py's `batch_sweep.hipgraph_ms_per_step`): loss, log-likelihoods and dZ of every replay equal
    the eager step bit for bit -- the ABI calls take the capturing stream, allocate nothing themselves and read no host-side state."""
    c, s_, q_, d = 5, 5, 16, 64
    n = c * (s_ + q_)
    for b in (1, 3):
        z = dev_t(O.synthetic_features(b, n, d, 3 + b), cuda).requires_grad_(True)
        y = dev_t(O.one_vs_rest_targets(c, s_ + q_), cuda)
        raw_s = dev_t(np.linspace(-0.3, 0.4, c), cuda).requires_grad_(True)
        mean = dev_t(0.02 * np.arange(c), cuda).requires_grad_(True)
        noise = dev_t(np.full(c, 0.1), cuda)
        cw = dev_t(np.full(c, -1.0 / (c * n)), cuda)

        def step():
            z.grad = None; raw_s.grad = None; mean.grad = None
            obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, torch.nn.functional.softplus(raw_s), mean, noise, cw, unit_rows=True)
            loss = obj.mean()
            loss.backward()
            return loss, logp

        side = torch.cuda.Stream(device=cuda)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
            loss_e, logp_e = step()
            eager = [t.detach().clone() for t in (loss_e, logp_e, z.grad, raw_s.grad, mean.grad)]
            del loss_e, logp_e                                   # (no reference into the eager step's autograd graph may survive into the capture)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss_g, logp_g = step()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        for a, bb in zip(eager, (loss_g, logp_g, z.grad, raw_s.grad, mean.grad)):
            assert torch.equal(a, bb)


def test_mll_regression_head_noise_at_its_lower_bound_takes_the_exact_kernel(cuda):
    """The regression head learns its noise (`DKT_regression.py:29, 53-54`: GaussianLikelihood, noise = softplus(raw) + 1e-4): driven to the 1e-4 bound, the 19 x 19 RBF
    model has cond(K) ~ sv N / noise ~ 1e5 -- far beyond what the 22-bit f16 splits resolve.  The default dispatch decides on the device (a-priori bound 1 + sv trace(E) / noise
    > 5e3, no host read-back) and hands these tasks to the exact-fp32 kernel: outputs bitwise equal to the generic twin, log-likelihood / gradients against float64 at the
    accuracy an fp32 factorisation has at this conditioning (eps kappa: stated below), where the raw split kernels are an order of magnitude off; a task at the INITIAL noise
    (softplus(0) + 1e-4 = 0.693) in the same batch stays on the split kernels."""
    rng = np.random.default_rng(19)
    b, n, d = 6, 19, 2916
    x = rng.standard_normal((b, 1, d)) + 0.02 * rng.standard_normal((b, n, d))      # 19 frames of one sequence: nearly the same image (QMUL head poses)
    d2 = ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)
    e = np.exp(-0.5 * d2 / 3200.0)                                     # off-diagonal kernel values ~ 0.9996: E is within 4e-4 of rank one
    y = rng.standard_normal((b, 1, n))
    sv, mean = np.array([0.8]), np.array([0.05])
    cw = np.array([-1.0 / n])
    for noise_v, guarded in ((1e-4 + 1e-6, True), (np.log(2.0) + 1e-4, False)):
        noise = np.array([noise_v])
        args = (dev_t(e, cuda), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda))
        o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
        raw = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), no_kappa_guard=True)
        gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
        twin = gen if guarded else raw
        for key in ("logp", "alpha", "w", "dsv", "dmean", "dnoise", "jitter", "info"):
            assert torch.equal(o[key], twin[key]), (guarded, key)
        worst = dict(logp=0.0, alpha=0.0, w=0.0, raw_logp=0.0, raw_w=0.0)
        for i in range(b):
            e32 = e[i].astype(np.float32).astype(np.float64)
            kk = sv[0] * e32 + noise_v * np.eye(n)
            r = y[i, 0] - mean[0]
            kinv = np.linalg.inv(kk)
            alpha = kinv @ r
            logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
            w_ref = cw[0] * sv[0] * 0.5 * (np.outer(alpha, alpha) - kinv)
            worst["logp"] = max(worst["logp"], abs(o["logp"][i, 0].item() - logp) / abs(logp))
            worst["alpha"] = max(worst["alpha"], rel_l2(o["alpha"][i, 0].cpu().numpy(), alpha))
            worst["w"] = max(worst["w"], rel_l2(o["w"][i].cpu().numpy(), w_ref))
            worst["raw_logp"] = max(worst["raw_logp"], abs(raw["logp"][i, 0].item() - logp) / abs(logp))
            worst["raw_w"] = max(worst["raw_w"], rel_l2(raw["w"][i].cpu().numpy(), w_ref))
            cond = np.linalg.cond(kk)
        if guarded:
            # cond(K) ~ 1e4 ... 1e5: an fp32 factorisation has eps kappa ~ 1e-3 ... 6e-3 on alpha / W; the log-likelihood (dominated by the log-determinant) holds 1e-4
            assert cond > 5e3, cond
            print("regression head at the noise bound: cond %.1e, exact kernel %s" % (cond, {k: "%.1e" % v for k, v in worst.items()}))
            # (measured: cond 4.4e4; exact kernel log-likelihood 1.5e-4, alpha 2.6e-4 -- eps kappa = 2.6e-3 is the bound, a float64 run of the same kernel would be needed
            #  for 1e-4 --; the raw split kernel 2.8e-4 / 8e-4 on W: the dispatch halves the error, it cannot buy digits fp32 does not have)
            assert worst["logp"] < 3 * MLL_RTOL and worst["alpha"] < 2e-3 and worst["w"] < 5e-3, (worst, cond)
            assert worst["logp"] <= worst["raw_logp"] and worst["w"] <= 2.0 * worst["raw_w"], worst
        else:
            assert worst["logp"] < MLL_RTOL and worst["alpha"] < 5e-4 and worst["w"] < GRAD_RTOL, worst


@pytest.mark.parametrize("guard", ["1", "0", "-1"])
def test_mll_wave_per_episode_class_weights_signs_and_units(cuda, monkeypatch, guard):
    """The wave-per-episode kernel (csrc/dkt_mll_h2.hip, mll_h2e_kernel) accumulates W over the classes inside the phase-3 products: the
    class weight is folded into the split scale, the accumulators carry one sign and one power-of-two unit.  Class weights of both signs,
    a zero weight, output scales three orders of magnitude apart (the unit grows from class to class) and C = 1 / 7 must reproduce the
    oracle, bitwise symmetrically and reproducibly -- at the default head room of the f16 scale of M (guard 1), without head room (0) and
    through the grow-on-demand path on every matrix (-1)."""
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1")
    monkeypatch.setenv("DKT_MLL_P2_GUARD", guard)
    rng = np.random.default_rng(11)
    for (c, per, d) in ((5, 21, 64), (7, 9, 32), (1, 40, 16), (3, 37, 24)):
        n = c * per
        z = O.synthetic_features(3, n, d, 40 + c, 0)
        y = O.one_vs_rest_targets(c, per)
        sv = np.array([0.7, 0.02, 9.0, 0.5, 300.0, 1.3, 0.05])[:c]
        mean = 0.05 * rng.standard_normal(c)
        noise = np.array([0.1, 0.03, 0.4, 0.1, 2.0, 0.2, 0.1])[:c]
        cw = np.array([-0.01, 0.02, 0.0, -0.3, 0.004, 0.05, -0.01])[:c]
        e_dev = ops.gram(dev_t(z, cuda))
        args = (e_dev, dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda))
        out = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
        again = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
        torch.cuda.synchronize()
        assert int(out["info"].abs().max().item()) == 0
        for key in ("logp", "alpha", "w", "dsv", "dmean", "dnoise"):
            assert torch.equal(out[key], again[key]), key
        for i in range(3):
            e = O.gram_linear(z[i])
            res = O.mll_terms(e, y, sv, mean, noise)
            assert np.abs((out["logp"][i].cpu().numpy() - res.logp) / res.logp).max() < MLL_RTOL
            assert rel_l2(out["alpha"][i].cpu().numpy(), res.alpha) < 5e-4
            w_ref, _, _, _ = O.mll_grads(e, res, sv, noise, cw)
            _, dsv, dmean, dnoise = O.mll_grads(e, res, sv, noise, np.ones(c))
            wk = out["w"][i].cpu().numpy()
            assert rel_l2(wk, w_ref) < GRAD_RTOL, (c, rel_l2(wk, w_ref))
            assert (wk == wk.T).all()
            assert rel_l2(out["dsv"][i].cpu().numpy(), dsv) < GRAD_RTOL
            assert rel_l2(out["dmean"][i].cpu().numpy(), dmean) < GRAD_RTOL
            assert rel_l2(out["dnoise"][i].cpu().numpy(), dnoise) < GRAD_RTOL


@pytest.mark.parametrize("c,per,d", [(5, 21, 48), (5, 5, 32), (3, 37, 24), (7, 9, 16), (5, 24, 16), (1, 127, 8), (2, 56, 8)])      # the last three: 112 <= N <= 127 (round 5)
def test_mll_per_class_base_matrices_one_launch(cuda, c, per, d):
    """DKT_MLL_E_PER_CLASS: rbf / matern / polynomial class models own their lengthscale / offset (one ExactGPLayer per class,
    methods/DKT.py:63-66, 352-365), so K_c = sv_c E[b, c] + noise_c I with a base matrix per class.  One launch over all (episode,
    class) matrices must reproduce the oracle per class -- log-likelihood, alpha, W[b, c] = d obj / d E[b, c], hyper-gradients -- with
    and without the gradient outputs, and the C single-model launches it replaces."""
    n = c * per
    rng = np.random.default_rng(c * 100 + per)
    z = O.synthetic_features(2, n, d, 300 + c, 0)
    y = O.one_vs_rest_targets(c, per)
    ls = np.linspace(0.8, 1.9, c)
    sv = np.linspace(0.5, 2.0, c)
    mean = 0.05 * rng.standard_normal(c)
    noise = np.linspace(0.08, 0.3, c)
    cw = np.full(c, -1.0 / (c * n))
    e64 = np.stack([np.stack([O.gram_rbf(z[i], None, ls[k]) for k in range(c)]) for i in range(2)])      # [2, C, N, N]
    args = (dev_t(e64, cuda), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda))
    out = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
    fwd = ops.mll(*args)
    torch.cuda.synchronize()
    assert out["w"].shape == (2, c, n, n) and int(out["info"].abs().max().item()) == 0 and int(fwd["info"].abs().max().item()) == 0
    for key in ("logp", "alpha"):
        assert rel_l2(fwd[key].cpu().numpy(), out[key].cpu().numpy()) < 2e-6, key
    for i in range(2):
        for k in range(c):
            e = e64[i, k].astype(np.float32).astype(np.float64)
            res = O.mll_terms(e, y[k:k + 1], sv[k:k + 1], mean[k:k + 1], noise[k:k + 1])
            assert abs((out["logp"][i, k].item() - res.logp[0]) / res.logp[0]) < MLL_RTOL
            assert rel_l2(out["alpha"][i, k].cpu().numpy(), res.alpha[0]) < 5e-4
            w_ref, dsv, dmean, dnoise = O.mll_grads(e, res, sv[k:k + 1], noise[k:k + 1], cw[k:k + 1])
            _, dsv1, dmean1, dnoise1 = O.mll_grads(e, res, sv[k:k + 1], noise[k:k + 1], np.ones(1))
            wk = out["w"][i, k].cpu().numpy()
            assert rel_l2(wk, w_ref) < GRAD_RTOL and (wk == wk.T).all()
            assert abs(out["dsv"][i, k].item() - dsv1[0]) < GRAD_RTOL * abs(dsv1[0]) + 1e-5
            assert abs(out["dmean"][i, k].item() - dmean1[0]) < GRAD_RTOL * abs(dmean1[0]) + 1e-6
            assert abs(out["dnoise"][i, k].item() - dnoise1[0]) < GRAD_RTOL * abs(dnoise1[0]) + 1e-5
            # the single-model launch this replaces (per-class Python loop of rounds 1 - 2)
            one = ops.mll(dev_t(e64[i:i + 1, k], cuda), dev_t(y[k:k + 1], cuda), dev_t(sv[k:k + 1], cuda), dev_t(mean[k:k + 1], cuda),
                          dev_t(noise[k:k + 1], cuda), want_grad=True, cls_weight=dev_t(cw[k:k + 1], cuda))
            assert rel_l2(out["w"][i, k].cpu().numpy(), one["w"][0].cpu().numpy()) < 2e-5
            assert abs(out["logp"][i, k].item() - one["logp"][0, 0].item()) < 5e-6 * abs(one["logp"][0, 0].item())
    # the generic kernel with one workgroup per matrix (DKT_MLL_FORCE_GENERIC: the validation twin of the per-class paths, ABI 4)
    gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    assert int(gen["info"].abs().max().item()) == 0
    assert rel_l2(out["logp"].cpu().numpy(), gen["logp"].cpu().numpy()) < 2e-6 and rel_l2(out["w"].cpu().numpy(), gen["w"].cpu().numpy()) < 5e-5
    assert rel_l2(out["alpha"].cpu().numpy(), gen["alpha"].cpu().numpy()) < 5e-5
    # argument errors: the Cholesky output and the other validation twins do not exist for per-class matrices
    with pytest.raises(RuntimeError):
        ops.mll(*args, want_chol=True)
    with pytest.raises(RuntimeError):
        ops.mll(*args, want_grad=True, force_reg=True)
    with pytest.raises(RuntimeError):
        ops.mll(*args, want_grad=True, force_blocked=True)


@pytest.mark.parametrize("b,c,per,chunk", [(2, 5, 30, None), (2, 20, 21, None), (5, 3, 100, "4"), (1, 2, 223, None)])
def test_mll_per_class_base_matrices_tile_array(cuda, monkeypatch, b, c, per, chunk):
    """DKT_MLL_E_PER_CLASS at 128 <= N <= 447: the tile-array pipeline with one base matrix, one factor and one W per (episode, class) matrix
    (the 20-way rbf / matern / polynomial episode of 420 rows in ONE dkt_mll_f32 call).  Against the float64 oracle per matrix, the single-model
    launch (shared-matrix pipeline with C = 1) it replaces, and the forward-only call (fp32 kernels); `chunk`: several passes over the workspace
    (a per-class pass covers half the episodes of a shared-matrix pass) with a ragged last pass."""
    if chunk is not None:
        monkeypatch.setenv("DKT_MLL_TILED_CHUNK", chunk)
    n = c * per
    rng = np.random.default_rng(c * 1000 + per)
    z = O.synthetic_features(b, n, 24, 500 + c, 0)
    y = O.one_vs_rest_targets(c, per)
    ls = np.linspace(0.8, 1.9, c)
    sv = np.linspace(0.5, 2.0, c)
    mean = 0.05 * rng.standard_normal(c)
    noise = np.linspace(0.08, 0.3, c)
    cw = np.full(c, -1.0 / (c * n))
    cw[c // 2] *= -2.0                                                       # a class weight of the other sign
    e64 = np.stack([np.stack([O.gram_rbf(z[i], None, ls[k]) for k in range(c)]) for i in range(b)])      # [B, C, N, N]
    args = (dev_t(e64, cuda), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda))
    out = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
    fwd = ops.mll(*args)
    torch.cuda.synchronize()
    assert out["w"].shape == (b, c, n, n) and int(out["info"].abs().max().item()) == 0 and int(fwd["info"].abs().max().item()) == 0
    assert rel_l2(fwd["logp"].cpu().numpy(), out["logp"].cpu().numpy()) < 1e-5
    assert rel_l2(fwd["alpha"].cpu().numpy(), out["alpha"].cpu().numpy()) < 2e-4
    w_all = out["w"].cpu().numpy()
    assert (w_all == w_all.transpose(0, 1, 3, 2)).all()
    for i in sorted({0, b - 1}):
        for k in sorted({0, c // 2, c - 1}):
            e = e64[i, k].astype(np.float32).astype(np.float64)
            res = O.mll_terms(e, y[k:k + 1], sv[k:k + 1], mean[k:k + 1], noise[k:k + 1])
            assert abs((out["logp"][i, k].item() - res.logp[0]) / res.logp[0]) < MLL_RTOL
            assert rel_l2(out["alpha"][i, k].cpu().numpy(), res.alpha[0]) < 5e-4
            w_ref, _, _, _ = O.mll_grads(e, res, sv[k:k + 1], noise[k:k + 1], cw[k:k + 1])
            _, dsv1, dmean1, dnoise1 = O.mll_grads(e, res, sv[k:k + 1], noise[k:k + 1], np.ones(1))
            assert rel_l2(w_all[i, k], w_ref) < GRAD_RTOL, (i, k, rel_l2(w_all[i, k], w_ref))
            assert abs(out["dsv"][i, k].item() - dsv1[0]) < GRAD_RTOL * abs(dsv1[0]) + 1e-5
            # (d logp / d mean = sum_i alpha_i cancels: the tolerance follows |alpha|_1, not the sum)
            assert abs(out["dmean"][i, k].item() - dmean1[0]) < GRAD_RTOL * abs(dmean1[0]) + 1e-6 + 2e-6 * np.abs(res.alpha[0]).sum()
            assert abs(out["dnoise"][i, k].item() - dnoise1[0]) < GRAD_RTOL * abs(dnoise1[0]) + 1e-5
            one = ops.mll(dev_t(e64[i:i + 1, k], cuda), dev_t(y[k:k + 1], cuda), dev_t(sv[k:k + 1], cuda), dev_t(mean[k:k + 1], cuda),
                          dev_t(noise[k:k + 1], cuda), want_grad=True, cls_weight=dev_t(cw[k:k + 1], cuda))
            assert rel_l2(w_all[i, k], one["w"][0].cpu().numpy()) < 2e-5
            assert abs(out["logp"][i, k].item() - one["logp"][0, 0].item()) < 5e-6 * abs(one["logp"][0, 0].item())
    # a matrix that is not positive definite: reported per matrix (info != 0, NaN), the other matrices of the call untouched
    bad = dev_t(e64, cuda).clone()
    bad[0, c - 1] = -bad[0, c - 1]
    o2 = ops.mll(bad, *args[1:], want_grad=True, cls_weight=dev_t(cw, cuda))
    assert int(o2["info"][0, c - 1].item()) != 0 and torch.isnan(o2["logp"][0, c - 1])
    keep = torch.ones(b, c, dtype=torch.bool, device=cuda)
    keep[0, c - 1] = False
    assert int(o2["info"][keep].abs().max().item()) == 0 and torch.equal(o2["logp"][keep], out["logp"][keep])
    assert torch.equal(o2["w"][keep], out["w"][keep])


def _class_kernel_ref(base, kernel, param):
    """float64 torch restatement of the per-class maps (gpytorch RBFKernel / MaternKernel(nu=2.5) / PolynomialKernel, DKT.py:352-365)."""
    p = param.reshape(1, -1, 1, 1)
    if kernel in ("poli1", "poli2"):
        return (base.unsqueeze(1) + p) ** (1 if kernel == "poli1" else 2)
    u = base.unsqueeze(1) / p ** 2
    if kernel == "rbf":
        return torch.exp(-0.5 * u)
    r = torch.sqrt(5.0 * u.clamp_min(1e-30))
    return (1.0 + r + r * r / 3.0) * torch.exp(-r)


@pytest.mark.parametrize("kernel", ["rbf", "matern", "poli1", "poli2"])
@pytest.mark.parametrize("b,c,n,d", [(3, 5, 25, 16), (2, 5, 105, 64), (2, 3, 64, 8), (1, 20, 33, 12), (2, 20, 420, 16), (3, 7, 258, 8), (1, 32, 130, 8)])
def test_class_kernel_maps_and_chain_rule(cuda, kernel, b, c, n, d):
    """dkt_class_kernel_f32 / dkt_class_kernel_bwd_f32: E[b,c] = f(base[b]; param_c) and, for a symmetric W[b,c] = d obj / d E[b,c], the matrix
    Wp with d obj / d Z = (Wp + Wp^T) Z and d obj / d param -- against float64 autograd of obj = sum W . f(base(z); param)."""
    rng = np.random.default_rng(n_hash(b, c, n, d) + len(kernel))
    z = (0.4 * rng.standard_normal((b, n, d))).astype(np.float32)
    param = np.linspace(0.7, 1.8, c).astype(np.float32)
    w = rng.standard_normal((b, c, n, n)).astype(np.float32)
    w = 0.5 * (w + w.transpose(0, 1, 3, 2))
    cmap, power, _, base_kind = ops._classmap_of(kernel, torch.tensor(param), torch.tensor(param))
    one = torch.ones(1, device=cuda)
    zd = dev_t(z, cuda)
    base = ops.gram(zd, None, base_kind, one if base_kind == ops.KERNEL_SQDIST else None)
    e = ops.class_kernel(base, cmap, power, dev_t(param, cuda))
    wp, dpar = ops.class_kernel_bwd(dev_t(w, cuda), base, cmap, power, dev_t(param, cuda))
    dz = ops.gram_bwd(wp, zd)
    # float64 autograd reference
    z64 = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    p64 = torch.tensor(param, dtype=torch.float64, requires_grad=True)
    if base_kind == ops.KERNEL_SQDIST:
        diff = z64.unsqueeze(2) - z64.unsqueeze(1)
        base64 = (diff * diff).sum(-1)
    else:
        base64 = z64 @ z64.transpose(1, 2)
    e64 = _class_kernel_ref(base64, kernel, p64)
    obj = (torch.tensor(w, dtype=torch.float64) * e64).sum((1, 2, 3))          # per episode
    assert rel_l2(e.cpu().numpy(), e64.detach().numpy()) < 2e-5
    dpar_ref = np.stack([torch.autograd.grad(obj[i], p64, retain_graph=True)[0].numpy() for i in range(b)])
    dz_ref = torch.autograd.grad(obj.sum(), z64)[0].numpy()
    assert rel_l2(dz.cpu().numpy(), dz_ref) < 1e-4, rel_l2(dz.cpu().numpy(), dz_ref)
    assert rel_l2(dpar.cpu().numpy(), dpar_ref) < 1e-4, rel_l2(dpar.cpu().numpy(), dpar_ref)
    # N > 128 took the 16-byte kernel of round 4: the dword kernel (round 3; what serves N <= 128 and N > 512) on the same inputs
    os.environ["DKT_CLASS_BWD_V4"] = "0"
    try:
        wp0, dpar0 = ops.class_kernel_bwd(dev_t(w, cuda), base, cmap, power, dev_t(param, cuda))
    finally:
        os.environ.pop("DKT_CLASS_BWD_V4")
    assert rel_l2(wp.cpu().numpy(), wp0.cpu().numpy()) < 2e-6 and rel_l2(dpar.cpu().numpy(), dpar0.cpu().numpy()) < 2e-5
    # cross matrices (test time): [B, M, N] bases go through the same map
    ex = ops.kernel_matrix_per_class(zd[:, :7], zd, kernel, dev_t(param, cuda), dev_t(param, cuda))
    assert ex.shape == (b, c, 7, n) and rel_l2(ex.cpu().numpy(), e64.detach().numpy()[:, :, :7]) < 2e-5


@pytest.mark.parametrize("n,d", [(33, 64), (85, 512), (105, 1600), (128, 100), (64, 36)])
def test_gram_distance_kinds_episode_resident(cuda, n, d):
    """dkt_gram_f32 kinds SQDIST / RBF at 32 < N <= 128 and a batch that fills the GPU: the episode-resident kernel (features shifted by row 0 while
    they are staged, split bf16 products, |y_i|^2 off the diagonal) against float64 and against the generic tile kernel (small batch)."""
    b = 64
    rng = np.random.default_rng(n * 13 + d)
    z = (np.abs(rng.standard_normal((b, n, d))) * 0.3 + rng.uniform(0.0, 2.0, (1, 1, d))).astype(np.float32)     # ReLU-like, common offset
    ls = dev_t(np.array([1.7 * np.sqrt(d) * 0.3]), cuda)
    zd = dev_t(z, cuda)
    z64 = z.astype(np.float64)
    for kind in (ops.KERNEL_SQDIST, ops.KERNEL_RBF):
        e = ops.gram(zd, None, kind, ls)
        eg = ops.gram(zd[:4].contiguous(), None, kind, ls)                       # B < 64: generic kernel
        assert torch.equal(e, e.transpose(1, 2))
        l2 = float(ls.item()) ** 2
        for i in (0, 1, b - 1):
            diff = z64[i][:, None, :] - z64[i][None, :, :]
            u = (diff * diff).sum(-1) / l2
            ref = np.exp(-0.5 * u) if kind == ops.KERNEL_RBF else u
            err = np.abs(e[i].cpu().numpy() - ref).max()
            assert err < 2e-5 * max(1.0, ref.max()), (kind, err)
        assert (e[:4] - eg).abs().max().item() < 2e-5 * max(1.0, float(eg.abs().max().item()))
        dg = torch.diagonal(e, dim1=1, dim2=2)
        assert torch.equal(dg, torch.ones_like(dg) if kind == ops.KERNEL_RBF else torch.zeros_like(dg))


def n_hash(*a):
    return int(sum((i + 1) * v for i, v in enumerate(a)))


def test_mll_per_episode_targets_and_residual_property(cuda):
    """Regression-style: y differs per episode ([B,C,N]); property check K alpha = r at full size."""
    rng = np.random.default_rng(3)
    b, n, d = 4, 19, 128
    z = np.abs(rng.standard_normal((b, n, d))).astype(np.float32)
    y = rng.standard_normal((b, 1, n)).astype(np.float32)
    ls, sv, mean, noise = 9.0, 0.8, 0.1, 0.3
    e = ops.gram(dev_t(z, cuda), None, ops.KERNEL_RBF, dev_t([ls], cuda))
    out = ops.mll(e, dev_t(y, cuda), dev_t([sv], cuda), dev_t([mean], cuda), dev_t([noise], cuda), want_chol=True)
    en = e.cpu().numpy().astype(np.float64)
    for i in range(b):
        res = O.mll_terms(O.gram_rbf(z[i].astype(np.float64), None, ls), y[i], sv, mean, noise)
        assert abs(out["logp"][i, 0].item() - res.logp[0]) < MLL_RTOL * abs(res.logp[0])
        k = sv * en[i] + noise * np.eye(n)
        r = y[i, 0] - mean
        assert np.abs(k @ out["alpha"][i, 0].cpu().numpy().astype(np.float64) - r).max() < 1e-4
        l = out["chol"][i, 0].cpu().numpy().astype(np.float64)
        assert np.abs(l @ l.T - k).max() < 1e-5 and np.abs(np.triu(l, 1)).max() == 0.0


@pytest.mark.parametrize("path", ["default", "h2e", "f32mfma", "reg", "generic"])
def test_jitter_retry_and_failure_info(cuda, path, monkeypatch):
    force_generic, force_reg, force_f32 = path == "generic", path == "reg", path == "f32mfma"
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1" if path == "h2e" else "1000000000")
    rng = np.random.default_rng(0)
    q, _ = np.linalg.qr(rng.standard_normal((8, 8)))
    y = np.ones((1, 8))

    def run(min_eig):
        e = q @ np.diag([min_eig, 0.3, 0.5, 0.7, 1.0, 1.2, 1.5, 2.0]) @ q.T
        e = 0.5 * (e + e.T)
        o = ops.mll(dev_t(e[None], cuda), dev_t(y, cuda), dev_t([1.0], cuda), dev_t([0.0], cuda), dev_t([0.1], cuda),
                    want_grad=True, force_generic=force_generic, force_reg=force_reg, force_f32mfma=force_f32)
        return e, o

    # K = E + 0.1 I has smallest eigenvalue -5e-5: plain, 1e-6, 1e-5 fail; total jitter 1e-4 succeeds
    e, o = run(-0.1 - 5e-5)
    assert int(o["info"].item()) == 0 and o["jitter"].item() == pytest.approx(1e-4, rel=1e-5)
    l, jit = O.psd_safe_cholesky(e + 0.1 * np.eye(8), 1e-6, 3)
    assert jit == pytest.approx(1e-4)
    # -5e-6: succeeds at 1e-5
    _, o = run(-0.1 - 5e-6)
    assert int(o["info"].item()) == 0 and o["jitter"].item() == pytest.approx(1e-5, rel=1e-5)
    # hopeless: every retry fails -> info > 0 (LAPACK-style pivot index), outputs poisoned with NaN
    _, o = run(-0.5)
    assert int(o["info"].item()) > 0 and torch.isnan(o["logp"]).all() and torch.isnan(o["alpha"]).all()
    # healthy matrix: no jitter
    _, o = run(0.2)
    assert int(o["info"].item()) == 0 and o["jitter"].item() == 0.0


@pytest.mark.parametrize("path", ["tiled", "blocked"])
@pytest.mark.parametrize("n", [150, 230, 333])
def test_jitter_retry_and_failure_info_blocked_path(cuda, n, path):
    """N > 127: the tile-array path (default) and its blocked twin (diagonal-block sweeps + batched GEMMs; also what serves
    want_chol).  A batch mixes healthy matrices, matrices that need 1e-5 / 1e-4 of jitter and a hopeless one; every matrix must
    retry on its own (GPyTorch psd_safe_cholesky per matrix)."""
    blocked = path == "blocked"
    rng = np.random.default_rng(n)
    qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    base = np.linspace(0.3, 2.0, n)
    mins = [0.2, -0.1 - 5e-6, -0.1 - 5e-5, -0.5, 0.25]
    es = []
    for mn in mins:
        ev = base.copy()
        ev[n // 2] = mn                                   # the bad direction sits in the middle (second diagonal block)
        e = qm @ np.diag(ev) @ qm.T
        es.append(0.5 * (e + e.T))
    e_all = np.stack(es)
    y = np.sign(rng.standard_normal((2, n)))
    sv, mean, noise = np.array([1.0, 1.0]), np.array([0.0, 0.1]), np.array([0.1, 0.1])
    o = ops.mll(dev_t(e_all, cuda), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda), want_grad=True, want_chol=blocked,
                cls_weight=dev_t([1.0, 1.0], cuda), force_blocked=blocked)
    info, jit = o["info"].cpu().numpy(), o["jitter"].cpu().numpy()
    assert (info[[0, 1, 2, 4]] == 0).all() and (info[3] > 0).all()
    assert np.allclose(jit[0], 0.0) and np.allclose(jit[4], 0.0)
    assert np.allclose(jit[1], 1e-5, rtol=1e-5) and np.allclose(jit[2], 1e-4, rtol=1e-5)
    assert torch.isnan(o["logp"][3]).all() and torch.isnan(o["alpha"][3]).all() and torch.isnan(o["w"][3]).all()
    for i in (0, 1, 2, 4):
        for c in range(2):
            k = sv[c] * e_all[i] + (noise[c] + jit[i, c]) * np.eye(n)
            l = np.linalg.cholesky(k)
            r = y[c] - mean[c]
            alpha = np.linalg.solve(k, r)
            logp = -0.5 * r @ alpha - np.log(np.diag(l)).sum() - 0.5 * n * np.log(2 * np.pi)
            # jittered matrices are ill-conditioned (min eigenvalue ~5e-5): compare at the accuracy fp32 allows there
            tol = 1e-4 if jit[i, c] == 0.0 else 5e-2
            assert abs(o["logp"][i, c].item() - logp) < tol * abs(logp)
            if jit[i, c] == 0.0:
                assert rel_l2(o["alpha"][i, c].cpu().numpy(), alpha) < 5e-4
                if blocked:
                    assert rel_l2(o["chol"][i, c].cpu().numpy(), l) < 5e-5
    # the generic twin agrees on what failed and which jitter was used
    g = ops.mll(dev_t(e_all, cuda), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda), force_generic=True)
    assert (g["info"].cpu().numpy() != 0).tolist() == (info != 0).tolist()
    assert np.allclose(g["jitter"].cpu().numpy(), jit)


@pytest.mark.parametrize("n", [8, 60, 105, 120, 150, 257])
def test_jitter_retry_and_failure_info_per_class_base_matrices(cuda, n):
    """DKT_MLL_E_PER_CLASS (rbf / matern / polynomial class models, methods/DKT.py:63-66, 352-365): every (episode, class) matrix retries on its
    own with psd_safe_cholesky's ladder (total jitter 1e-6, 1e-5, 1e-4), on every path -- N <= 111 the wave-per-episode-form kernel, 112 <= N <= 127
    the wave-per-matrix kernel (NT = 8), 128 <= N <= 447 the tile-array pipeline + the generic kernel's per-MATRIX fix-up launch (round 5; until
    round 4 that pipeline reported info != 0 + NaN where the reference retries).  A call mixes healthy matrices, matrices that need 1e-5 / 1e-4
    and a hopeless one; the healthy ones must come out exactly as in a call without the bad ones; the generic twin must agree."""
    rng = np.random.default_rng(n)
    qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    base = np.linspace(0.3, 2.0, n)
    mins = np.array([[0.2, -0.1 - 5e-6, 0.25], [-0.1 - 5e-5, -0.5, 0.3]])          # [B = 2, C = 3]: smallest eigenvalue of E[b, c]; noise 0.1, sv 1
    e_all = np.empty((2, 3, n, n))
    for i in range(2):
        for k in range(3):
            ev = base.copy()
            ev[n // 2] = mins[i, k]
            e = qm @ np.diag(ev) @ qm.T
            e_all[i, k] = 0.5 * (e + e.T)
    y = np.sign(rng.standard_normal((3, n)))
    sv, mean, noise, cw = np.ones(3), np.array([0.0, 0.1, -0.1]), np.full(3, 0.1), np.array([1.0, -0.5, 2.0])
    args = (dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda))
    o = ops.mll(dev_t(e_all, cuda), *args, want_grad=True, cls_weight=dev_t(cw, cuda))
    fwd = ops.mll(dev_t(e_all, cuda), *args)
    gen = ops.mll(dev_t(e_all, cuda), *args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    for res in (o, fwd, gen):
        info, jit = res["info"].cpu().numpy(), res["jitter"].cpu().numpy()
        assert (info[[0, 0, 0, 1, 1], [0, 1, 2, 0, 2]] == 0).all() and info[1, 1] > 0
        assert jit[0, 0] == 0.0 and jit[0, 2] == 0.0 and jit[1, 2] == 0.0
        assert np.isclose(jit[0, 1], 1e-5, rtol=1e-5) and np.isclose(jit[1, 0], 1e-4, rtol=1e-5)
        assert torch.isnan(res["logp"][1, 1]) and torch.isnan(res["alpha"][1, 1]).all()
    assert torch.isnan(o["w"][1, 1]).all() and torch.isnan(o["dsv"][1, 1])
    for i in range(2):
        for k in range(3):
            if (i, k) == (1, 1):
                continue
            _, jref = O.psd_safe_cholesky(sv[k] * e_all[i, k] + noise[k] * np.eye(n), 1e-6, 3)
            jit = o["jitter"][i, k].item()
            assert np.isclose(jit, jref, rtol=1e-5, atol=0.0)
            kk = sv[k] * e_all[i, k] + (noise[k] + jref) * np.eye(n)
            l = np.linalg.cholesky(kk)
            r = y[k] - mean[k]
            alpha = np.linalg.solve(kk, r)
            logp = -0.5 * r @ alpha - np.log(np.diag(l)).sum() - 0.5 * n * np.log(2 * np.pi)
            tol = 1e-4 if jref == 0.0 else 5e-2            # jittered matrices are ill-conditioned (min eigenvalue ~5e-5): the accuracy fp32 allows there
            assert abs(o["logp"][i, k].item() - logp) < tol * abs(logp)
            if jref == 0.0:
                assert rel_l2(o["alpha"][i, k].cpu().numpy(), alpha) < 5e-4
                w_ref = cw[k] * sv[k] * 0.5 * (np.outer(alpha, alpha) - np.linalg.inv(kk))
                assert rel_l2(o["w"][i, k].cpu().numpy(), w_ref) < GRAD_RTOL
    # the healthy matrices are bitwise what a call made of healthy matrices only gives (a retry / the fix-up launch touches nothing else)
    e_ok = e_all.copy()
    e_ok[0, 1] = e_all[0, 0]
    e_ok[1, 0] = e_all[0, 2]
    e_ok[1, 1] = e_all[1, 2]
    ok = ops.mll(dev_t(e_ok, cuda), *args, want_grad=True, cls_weight=dev_t(cw, cuda))
    for (i, k) in ((0, 0), (0, 2), (1, 2)):
        assert torch.equal(ok["logp"][i, k], o["logp"][i, k]) and torch.equal(ok["w"][i, k], o["w"][i, k]) and torch.equal(ok["alpha"][i, k], o["alpha"][i, k])


def test_predict_per_class_cross_kernels(cuda):
    """dkt_predict_per_class_f32 (ABI 4): posterior means + first-maximum labels from one base cross kernel PER class model (Ex [B,C,M,N]) -- what
    DKT._posterior runs for rbf / matern / polynomial kernels (methods/DKT.py:264-270, 352-365) instead of a torch einsum."""
    rng = np.random.default_rng(5)
    b, c, m, n = 3, 4, 37, 29
    ex = rng.standard_normal((b, c, m, n))
    alpha = rng.standard_normal((b, c, n))
    sv, mean = np.linspace(0.5, 2.0, c), 0.1 * rng.standard_normal(c)
    ex[0, :, 5] = 0.0
    mean_t = mean.copy()
    mu, labels = ops.predict(dev_t(ex, cuda), dev_t(alpha, cuda), dev_t(sv, cuda), dev_t(mean_t, cuda))
    ref = mean[None, :, None] + sv[None, :, None] * np.einsum("bcmn,bcn->bcm", ex.astype(np.float32).astype(np.float64), alpha.astype(np.float32).astype(np.float64))
    assert np.abs(mu.cpu().numpy() - ref).max() < 1e-5 * np.abs(ref).max()
    assert (labels.cpu().numpy() == mu.cpu().numpy().argmax(1)).all()
    # ties: equal means -> the first class wins (np.argmax)
    mu2, lab2 = ops.predict(torch.zeros(1, c, 3, n, device=cuda), dev_t(alpha[:1], cuda), dev_t(sv, cuda), torch.zeros(c, device=cuda))
    assert (lab2.cpu().numpy() == 0).all() and float(mu2.abs().max()) == 0.0
    # the shared-kernel entry on the same data repeated per class agrees bitwise
    ex_sh = rng.standard_normal((b, m, n))
    mu_a, lab_a = ops.predict(dev_t(ex_sh, cuda), dev_t(alpha, cuda), dev_t(sv, cuda), dev_t(mean, cuda))
    mu_b, lab_b = ops.predict(dev_t(np.repeat(ex_sh[:, None], c, 1), cuda), dev_t(alpha, cuda), dev_t(sv, cuda), dev_t(mean, cuda))
    assert torch.equal(mu_a, mu_b) and torch.equal(lab_a, lab_b)


@pytest.mark.parametrize("kernel", ["matern", "poli1", "poli2", "rbf", "linear"])
def test_kernel_matrix_single_model_maps_on_the_device(cuda, kernel):
    """ops.kernel_matrix (one model: the regression head, the per-class host loop beyond N = 447 / C = 32): Matern-2.5 and the polynomial kernels
    through dkt_gram_f32 + dkt_class_kernel_f32 (until round 4 the element-wise maps ran in torch), symmetric and cross."""
    rng = np.random.default_rng(9)
    a, bm = rng.standard_normal((2, 23, 12)) * 0.5, rng.standard_normal((2, 17, 12)) * 0.5
    ls, off = np.array([1.3]), np.array([0.7])
    ref = {"matern": lambda x, y_: O.gram_matern25(x, y_, ls[0]), "rbf": lambda x, y_: O.gram_rbf(x, y_, ls[0]), "linear": lambda x, y_: O.gram_linear(x, y_),
           "poli1": lambda x, y_: O.gram_poly(x, y_, 1, off[0]), "poli2": lambda x, y_: O.gram_poly(x, y_, 2, off[0])}[kernel]
    for second in (None, bm):
        e = ops.kernel_matrix(dev_t(a, cuda), None if second is None else dev_t(second, cuda), kernel, dev_t(ls, cuda), dev_t(off, cuda)).cpu().numpy()
        for i in range(2):
            r = ref(a[i].astype(np.float32).astype(np.float64), None if second is None else second[i].astype(np.float32).astype(np.float64))
            assert e[i].shape == r.shape and np.abs(e[i] - r).max() < 2e-5 * max(1.0, np.abs(r).max())


@pytest.mark.parametrize("n,c", [(128, 2), (143, 3), (144, 1), (257, 4), (320, 20), (360, 2), (400, 3), (447, 2), (460, 2)])      # 460: beyond the tile-array kernels -> blocked path; 360 / 400 / 447: W as an 8-wave and a 4-wave launch
def test_mll_tile_array_path_edges(cuda, n, c):
    """The tile-array kernels (N > 127) at the edges of their tiling -- the augmented row first / last in its tile (N = 128 / 143),
    a tile count that is not a multiple of the 4-tile blocks, the largest supported size -- against float64 and the blocked and
    generic twins, with and without gradients; W must be bitwise symmetric."""
    rng = np.random.default_rng(n + c)
    z = rng.standard_normal((3, n, 40))
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    e_np = np.einsum("bnd,bmd->bnm", z, z)
    y = np.sign(rng.standard_normal((c, n)))
    sv = 0.6 + 0.1 * np.arange(c)
    mean = 0.02 * np.arange(c)
    noise = np.full(c, 0.1)
    cw = np.full(c, -1.0 / (c * n))
    args = [dev_t(x, cuda) for x in (e_np, y, sv, mean, noise)]
    o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
    o_fwd = ops.mll(*args, want_grad=False)
    blk = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_blocked=True)
    gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    assert int(o["info"].abs().max().item()) == 0 and float(o["jitter"].abs().max().item()) == 0.0
    assert torch.equal(o["w"], o["w"].transpose(1, 2))
    w_ref = np.zeros((3, n, n))
    for b in range(3):
        for k in range(c):
            kk = sv[k] * e_np[b] + noise[k] * np.eye(n)
            l = np.linalg.cholesky(kk)
            r = y[k] - mean[k]
            alpha = np.linalg.solve(kk, r)
            logp = -0.5 * r @ alpha - np.log(np.diag(l)).sum() - 0.5 * n * np.log(2 * np.pi)
            assert abs(o["logp"][b, k].item() - logp) < MLL_RTOL * abs(logp)
            assert abs(o_fwd["logp"][b, k].item() - logp) < MLL_RTOL * abs(logp)
            assert rel_l2(o["alpha"][b, k].cpu().numpy(), alpha) < 2e-4
            assert rel_l2(o_fwd["alpha"][b, k].cpu().numpy(), alpha) < 2e-4
            w_ref[b] += cw[k] * sv[k] * 0.5 * (np.outer(alpha, alpha) - np.linalg.inv(kk))
    assert rel_l2(o["w"].cpu().numpy(), w_ref) < 2e-4
    for twin in (blk, gen):
        assert rel_l2(o["w"].cpu().numpy(), twin["w"].cpu().numpy()) < 1e-4
        np.testing.assert_allclose(o["logp"].cpu().numpy(), twin["logp"].cpu().numpy(), rtol=2e-5)
        for key in ("dsv", "dmean", "dnoise"):
            assert rel_l2(o[key].cpu().numpy(), twin[key].cpu().numpy()) < 2e-3, key


@pytest.mark.parametrize("n,c", [(130, 4), (257, 6)])
def test_mll_tile_array_w_kernel_class_weights_signs_and_hyper_ranges(cuda, n, c):
    """The tile-array W kernel folds the class weight into the scales of its f16 splits and takes the bound of M = R^-T from noise and kappa:
    class weights of both signs and of very different magnitude (one of them zero), noise from 0.05 to 0.5, outputscales from 0.3 to 5 (cond(K) up to
    ~25 x the reference's) -- W and the log-likelihood against float64, and W against the fp32-product kernel (DKT_MLL_TILED_F16=0)."""
    rng = np.random.default_rng(7 * n + c)
    z = rng.standard_normal((2, n, 48))
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    e_np = np.einsum("bnd,bmd->bnm", z, z)
    y = np.sign(rng.standard_normal((c, n)))
    sv = np.geomspace(0.3, 5.0, c)
    mean = 0.05 * rng.standard_normal(c)
    noise = np.geomspace(0.05, 0.5, c)[::-1].copy()
    cw = np.array([(-1.0) ** k * 10.0 ** (-k) for k in range(c)]) / n
    cw[c // 2] = 0.0
    args = [dev_t(x, cuda) for x in (e_np, y, sv, mean, noise)]
    o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
    assert int(o["info"].abs().max().item()) == 0
    assert torch.equal(o["w"], o["w"].transpose(1, 2))
    w_ref = np.zeros((2, n, n))
    for b in range(2):
        for k in range(c):
            kk = sv[k] * e_np[b] + noise[k] * np.eye(n)
            r = y[k] - mean[k]
            alpha = np.linalg.solve(kk, r)
            w_ref[b] += cw[k] * sv[k] * 0.5 * (np.outer(alpha, alpha) - np.linalg.inv(kk))
            logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
            assert abs(o["logp"][b, k].item() - logp) < MLL_RTOL * abs(logp)
    assert rel_l2(o["w"].cpu().numpy(), w_ref) < GRAD_RTOL, rel_l2(o["w"].cpu().numpy(), w_ref)
    os.environ["DKT_MLL_TILED_F16"] = "0"                # (ops.mll tells the library when a switch changed inside the process)
    try:
        o32 = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda))
    finally:
        del os.environ["DKT_MLL_TILED_F16"]
    assert rel_l2(o["w"].cpu().numpy(), o32["w"].cpu().numpy()) < 2e-5, rel_l2(o["w"].cpu().numpy(), o32["w"].cpu().numpy())
    assert rel_l2(o32["w"].cpu().numpy(), w_ref) < GRAD_RTOL


@pytest.mark.parametrize("switch", ["DKT_MLL_TILED_WRES=0", "DKT_MLL_TILED_F16=0", "DKT_MLL_TILED_INVRES=1", "DKT_MLL_TILED_WGS=2", "DKT_MLL_TILED_WNW=8", "DKT_MLL_TILED_WNW=4", "DKT_MLL_TILED_WNW=84"])
@pytest.mark.parametrize("n,c", [(320, 20), (420, 3), (190, 5)])
def test_mll_tile_array_pipeline_twins(cuda, n, c, switch):
    """The tile-array marginal likelihood's alternative pipelines -- round 3's kernels (fp32 tile arrays, block-column W), round 2's all-fp32 products, the
    resident-accumulator invert kernel (opt-in), two instead of three workgroups per CU -- against float64 and against the default (round 4: f16-split tile
    arrays, W with resident accumulators) on the same inputs."""
    rng = np.random.default_rng(n * c)
    b = 9                                                       # more than one 8-episode XCD group
    z = rng.standard_normal((b, n, 64))
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    e_np = np.einsum("bnd,bmd->bnm", z, z)
    y = np.sign(rng.standard_normal((c, n)))
    sv = 0.5 + 0.07 * np.arange(c)
    mean = 0.03 * rng.standard_normal(c)
    noise = np.full(c, 0.1)
    cw = np.full(c, -1.0 / (c * n))
    args = [dev_t(x, cuda) for x in (e_np, y, sv, mean, noise)]
    ref = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_tiled=True)      # (C = 20 would take the band reduction by default)
    k, v = switch.split("=")
    os.environ[k] = v
    try:
        o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_tiled=True)
    finally:
        del os.environ[k]
    assert int(o["info"].abs().max().item()) == 0 and torch.equal(o["w"], o["w"].transpose(1, 2))
    assert rel_l2(o["w"].cpu().numpy(), ref["w"].cpu().numpy()) < 2e-5
    np.testing.assert_allclose(o["logp"].cpu().numpy(), ref["logp"].cpu().numpy(), rtol=2e-5)
    assert rel_l2(o["alpha"].cpu().numpy(), ref["alpha"].cpu().numpy()) < 2e-5
    for key in ("dsv", "dmean", "dnoise"):
        assert rel_l2(o[key].cpu().numpy(), ref[key].cpu().numpy()) < 1e-3, key
    for bi in (0, b - 1):                                       # float64 on the first and the last episode
        w_ref = np.zeros((n, n))
        for kc in range(c):
            kk = sv[kc] * e_np[bi] + noise[kc] * np.eye(n)
            r = y[kc] - mean[kc]
            alpha = np.linalg.solve(kk, r)
            logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
            assert abs(o["logp"][bi, kc].item() - logp) < MLL_RTOL * abs(logp)
            w_ref += cw[kc] * sv[kc] * 0.5 * (np.outer(alpha, alpha) - np.linalg.inv(kk))
        assert rel_l2(o["w"][bi].cpu().numpy(), w_ref) < 2e-4


# ----------------------------------------------------------------------------------------------
# shared base matrix, many classes: ONE band reduction per episode (csrc/dkt_mll_band.hip, round 6)
# ----------------------------------------------------------------------------------------------
def _band_problem(b, n, c, d, corr, seed):
    """SURVEY 8d's synthetic episodes (oracle.synthetic_features: N(0,1) -> BatchNorm(train) -> L2 normalise; corr: 0.9 class mean + 0.1 noise before the
    BatchNorm, cond(K) ~ 1e2 .. 1e3 -- what trained features look like), class-major one-vs-rest targets, distinct hyper-parameters per class."""
    per = n // c
    z = np.zeros((b, n, d))
    z[:, :per * c] = O.synthetic_features(b, per * c, d, seed, c if corr else 0)
    if per * c < n:                                                 # N not a multiple of C: the tail rows join the last class
        z[:, per * c:] = O.synthetic_features(b, n - per * c, d, seed + 1, 0)
    cls = np.minimum(np.arange(n) // per, c - 1)
    e = np.einsum("bnd,bmd->bnm", z, z).astype(np.float32).astype(np.float64)
    y = np.where(cls[None, :] == np.arange(c)[:, None], 1.0, -1.0)
    sv = 0.5 + 0.04 * np.arange(c)
    mean = 0.01 * np.arange(c)
    noise = np.full(c, 0.1)
    cw = np.full(c, -1.0 / (c * n))
    return e, y, sv, mean, noise, cw


@pytest.mark.parametrize("n,c,d,corr", [(128, 8, 48, False), (150, 10, 64, True), (257, 16, 64, False), (320, 20, 128, False), (420, 20, 512, True),
                                        (431, 9, 64, True), (432, 32, 96, False), (400, 2, 64, False)])
def test_mll_band_reduction_vs_float64_and_tile_array_twin(cuda, n, c, d, corr):
    """The band reduction (the default of a shared-E call with 12 <= C <= 32, 128 <= N <= 432 and >= 192 episodes; named here through force_band): one orthogonal reduction of E to block-tridiagonal form per episode, every class a
    block LDL^T of B + mu_c I (methods/DKT.py:148-149, 161-163 at the 20-way shapes of train.py:132-133).  Against float64 on every episode and against the tile-array
    twin (force_tiled: one factorisation per class matrix); with and without gradients; W bitwise symmetric.  The tolerances are the file's (1e-4 / 1e-3) on
    uncorrelated AND on class-correlated features, where the reduction's backward error would cost 1e-4 on the quadratic form without the residual step."""
    b = 3
    e, y, sv, mean, noise, cw = _band_problem(b, n, c, d, corr, 100 * n + c)
    args = [dev_t(x, cuda) for x in (e, y, sv, mean, noise)]
    o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_band=True)          # (three episodes: below the default window of >= 192 episodes)
    o_fwd = ops.mll(*args, want_grad=False, force_band=True)
    twin = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_tiled=True)
    assert int(o["info"].abs().max().item()) == 0 and float(o["jitter"].abs().max().item()) == 0.0
    assert torch.equal(o["w"], o["w"].transpose(1, 2))
    worst = 0.0
    for bi in range(b):
        w_ref = np.zeros((n, n))
        hyper = ([], [], [])
        for k in range(c):
            kk = sv[k] * e[bi] + noise[k] * np.eye(n)
            r = y[k] - mean[k]
            kinv = np.linalg.inv(kk)
            alpha = kinv @ r
            logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
            worst = max(worst, abs(o["logp"][bi, k].item() - logp) / abs(logp))
            assert abs(o["logp"][bi, k].item() - logp) < 0.1 * MLL_RTOL * abs(logp), (bi, k, o["logp"][bi, k].item(), logp)
            assert abs(o_fwd["logp"][bi, k].item() - logp) < MLL_RTOL * abs(logp)
            assert rel_l2(o["alpha"][bi, k].cpu().numpy(), alpha) < 2e-4
            assert rel_l2(o_fwd["alpha"][bi, k].cpu().numpy(), alpha) < 2e-4
            m = 0.5 * (np.outer(alpha, alpha) - kinv)
            w_ref += cw[k] * sv[k] * m
            hyper[0].append((m * e[bi]).sum()); hyper[1].append(np.trace(m)); hyper[2].append(alpha.sum())
        assert rel_l2(o["w"][bi].cpu().numpy(), w_ref) < 2e-4
        for key, ref in zip(("dsv", "dnoise", "dmean"), hyper):
            assert rel_l2(o[key][bi].cpu().numpy(), np.array(ref)) < GRAD_RTOL, key
    assert rel_l2(o["w"].cpu().numpy(), twin["w"].cpu().numpy()) < 2e-4
    np.testing.assert_allclose(o["logp"].cpu().numpy(), twin["logp"].cpu().numpy(), rtol=1e-4)
    for key in ("dsv", "dmean", "dnoise"):
        assert rel_l2(o[key].cpu().numpy(), twin[key].cpu().numpy()) < 2e-3, key
    # bitwise repeatable
    o2 = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_band=True)
    for key in ("logp", "alpha", "w", "dsv", "dmean", "dnoise"):
        assert torch.equal(o[key], o2[key]), key


def test_mll_band_reduction_failure_goes_through_the_jitter_ladder(cuda):
    """A class whose matrix is singular at attempt 0 (rank-deficient E, zero noise): the band path runs attempt 0 only, flags the episode, and the generic kernel's
    fix-up launch redoes it with psd_safe_cholesky's ladder -- jitter, info and every output equal to the tile-array twin's (which ends in the same fix-up)."""
    rng = np.random.default_rng(5)
    b, c, n = 3, 8, 160
    z = rng.standard_normal((b, n, 16))
    e = np.einsum("bnd,bmd->bnm", z, z)
    y = np.where(np.arange(n)[None, :] % c == np.arange(c)[:, None], 1.0, -1.0)
    noise = np.full(c, 1e-2)
    noise[3] = 0.0
    args = [dev_t(x, cuda) for x in (e, y, np.ones(c), np.zeros(c), noise)]
    cw = dev_t(np.full(c, -1.0 / (c * n)), cuda)
    o = ops.mll(*args, want_grad=True, cls_weight=cw, force_band=True)
    twin = ops.mll(*args, want_grad=True, cls_weight=cw, force_tiled=True)
    assert float(o["jitter"][:, 3].min().item()) > 0.0 and float(o["jitter"][:, [0, 1, 2, 4, 5, 6, 7]].abs().max().item()) == 0.0
    for key in ("jitter", "info", "logp", "alpha", "w", "dsv", "dmean", "dnoise"):
        assert torch.equal(o[key], twin[key]), key


def test_mll_band_reduction_per_episode_targets_and_no_class_weights(cuda):
    """The ABI's other target layout (`Y[B,C,N]` with y_bstride = C N: every episode its own targets) and `cls_weight = NULL` (all ones) through the band reduction,
    against float64 on every episode and class."""
    rng = np.random.default_rng(31)
    b, n, c = 3, 176, 12
    e, _, sv, mean, noise, _ = _band_problem(b, n, c, 48, True, 31)
    y = np.sign(rng.standard_normal((b, c, n)))
    args = [dev_t(x, cuda) for x in (e, y, sv, mean, noise)]
    o = ops.mll(*args, want_grad=True, force_band=True)
    assert int(o["info"].abs().max().item()) == 0 and torch.equal(o["w"], o["w"].transpose(1, 2))
    for bi in range(b):
        w_ref = np.zeros((n, n))
        for k in range(c):
            kk = sv[k] * e[bi] + noise[k] * np.eye(n)
            r = y[bi, k] - mean[k]
            kinv = np.linalg.inv(kk)
            alpha = kinv @ r
            logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
            assert abs(o["logp"][bi, k].item() - logp) < MLL_RTOL * abs(logp)
            assert rel_l2(o["alpha"][bi, k].cpu().numpy(), alpha) < 2e-4
            w_ref += sv[k] * 0.5 * (np.outer(alpha, alpha) - kinv)
        assert rel_l2(o["w"][bi].cpu().numpy(), w_ref) < 2e-4


@pytest.mark.parametrize("seed", list(range(8)))
def test_mll_band_reduction_random_shapes_hypers_and_class_weights(cuda, seed):
    """Random (N, C, D), outputscales 0.2 ... 3, noises 0.05 ... 0.5, means, class weights of both signs and very different magnitude (one of them zero): the band
    reduction against the generic exact-fp32 kernel (same inputs, matrix factorised directly) and against float64 on one episode.  Covers tile counts 8 ... 27 with and
    without a partial last tile, every class-column layout of the chain kernel (waves with 1 ... 4 block columns) and the f16 scales of the back pass (|M| bound from |cw|)."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(128, 433))
    c = int(rng.integers(2, 33))
    d = int(rng.choice([24, 64, 200]))
    b = 2
    z = rng.standard_normal((b, n, d))
    if seed % 2:
        cm = rng.standard_normal((b, c, d))
        z = 0.8 * cm[:, rng.integers(0, c, n)] + 0.3 * z
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    e = np.einsum("bnd,bmd->bnm", z, z).astype(np.float32).astype(np.float64)
    y = np.sign(rng.standard_normal((c, n)))
    sv = np.exp(rng.uniform(np.log(0.2), np.log(3.0), c))
    noise = np.exp(rng.uniform(np.log(0.05), np.log(0.5), c))
    mean = 0.1 * rng.standard_normal(c)
    cw = rng.standard_normal(c) * 10.0 ** rng.integers(-3, 1, c) / n
    cw[rng.integers(0, c)] = 0.0
    if max(1.0 + sv * n / noise) > 1.5e4:                    # stay below the path's own condition guard (2e4): this test is about the reduction
        sv = sv * 1.5e4 / max(1.0 + sv * n / noise)
    args = [dev_t(x, cuda) for x in (e, y, sv, mean, noise)]
    o = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_band=True)
    gen = ops.mll(*args, want_grad=True, cls_weight=dev_t(cw, cuda), force_generic=True)
    assert int(o["info"].abs().max().item()) == 0 and torch.equal(o["w"], o["w"].transpose(1, 2))
    np.testing.assert_allclose(o["logp"].cpu().numpy(), gen["logp"].cpu().numpy(), rtol=MLL_RTOL)
    assert rel_l2(o["alpha"].cpu().numpy(), gen["alpha"].cpu().numpy()) < 5e-4
    assert rel_l2(o["w"].cpu().numpy(), gen["w"].cpu().numpy()) < GRAD_RTOL
    for key in ("dsv", "dmean", "dnoise"):
        assert rel_l2(o[key].cpu().numpy(), gen[key].cpu().numpy()) < 2 * GRAD_RTOL, key
    w_ref = np.zeros((n, n))
    for k in range(c):
        kk = sv[k] * e[0] + noise[k] * np.eye(n)
        r = y[k] - mean[k]
        kinv = np.linalg.inv(kk)
        alpha = kinv @ r
        logp = -0.5 * r @ alpha - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
        assert abs(o["logp"][0, k].item() - logp) < MLL_RTOL * abs(logp), (n, c, k)
        w_ref += cw[k] * sv[k] * 0.5 * (np.outer(alpha, alpha) - kinv)
    assert rel_l2(o["w"][0].cpu().numpy(), w_ref) < GRAD_RTOL, (n, c)


def test_mll_band_reduction_dispatch_window_and_condition_guard(cuda):
    """(i) The default takes the band reduction from 12 classes and 192 episodes per call and the tile-array kernels below (bitwise equal to the named paths).
    (ii) A class whose a-priori condition bound 1 + sv trace(E) / noise exceeds 2e4 (here: noise 1e-3 on unit rows) is not trusted to the reduction -- its episode is
    redone by the generic kernel on the matrix itself, so every output of the flagged episodes equals the generic twin's bit for bit and holds the file's tolerances."""
    n, c = 128, 12
    e, y, sv, mean, noise, cw = _band_problem(192, n, c, 32, False, 77)
    args = [dev_t(x, cuda) for x in (e, y, sv, mean, noise)]
    cwt = dev_t(cw, cuda)
    dflt = ops.mll(*args, want_grad=True, cls_weight=cwt)
    band = ops.mll(*args, want_grad=True, cls_weight=cwt, force_band=True)
    args191 = [args[0][:191].contiguous()] + args[1:]
    dflt191 = ops.mll(*args191, want_grad=True, cls_weight=cwt)
    tiled191 = ops.mll(*args191, want_grad=True, cls_weight=cwt, force_tiled=True)
    for key in ("logp", "alpha", "w", "dsv", "dmean", "dnoise"):
        assert torch.equal(dflt[key], band[key]), key
        assert torch.equal(dflt191[key], tiled191[key]), key
    assert not torch.equal(band["w"][:191], tiled191["w"])                      # (two different algorithms)
    # (ii)
    noise2 = noise.copy()
    noise2[5] = 1e-3
    args2 = [dev_t(x, cuda) for x in (e[:4], y, sv, mean, noise2)]
    o = ops.mll(*args2, want_grad=True, cls_weight=cwt, force_band=True)
    gen = ops.mll(*args2, want_grad=True, cls_weight=cwt, force_generic=True)
    assert int(o["info"].abs().max().item()) == 0
    for key in ("logp", "alpha", "w", "dsv", "dmean", "dnoise", "jitter"):
        assert torch.equal(o[key], gen[key]), key
    kk = sv[5] * e[0] + noise2[5] * np.eye(n)
    r = y[5] - mean[5]
    logp = -0.5 * r @ np.linalg.solve(kk, r) - 0.5 * np.linalg.slogdet(kk)[1] - 0.5 * n * np.log(2 * np.pi)
    assert abs(o["logp"][0, 5].item() - logp) < MLL_RTOL * abs(logp)


@pytest.mark.parametrize("b,n,d", [(72, 190, 512), (72, 320, 512), (72, 420, 512), (72, 431, 36), (72, 290, 64), (72, 447, 100), (136, 190, 512), (130, 250, 128)])
def test_gram_large_n_kernel_twins(cuda, b, n, d):
    """N > 128, unit rows, a batch that takes the round-4 kernels (episode-resident Gram for N <= 432; Gram backward in 128-row blocks for N > 256): against the
    round-2 kernels on the same inputs (DKT_GRAM_BIG_EP=0 / DKT_GRAM_BWD_ROWS8=0) and float64 on sampled episodes; symmetric, unit diagonal.  The rows of W differ in
    scale by orders of magnitude (the kernels scale every row by its own power of two), N is not always a multiple of 4 (row ends inside a 16-byte load) and
    the upstream gradient enters per episode.  The last two cases: 128 < N <= 256 with a batch that fills the GPU with 128-row blocks (the 64-row kernel below that)."""
    g = torch.Generator(device=cuda).manual_seed(n + d)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=cuda) * torch.exp(torch.randn(b, n, d, generator=g, device=cuda)), dim=2).contiguous()
    w = torch.randn(b, n, n, generator=g, device=cuda)
    rs = torch.exp(2.0 * torch.randn(b, n, 1, generator=g, device=cuda))
    w = ((w + w.transpose(1, 2)) * rs * rs.transpose(1, 2)).contiguous()
    eps = torch.linspace(0.5, 2.0, b, device=cuda)
    e_new = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
    dz_new = ops.gram_bwd(w, z, eps, unit_rows=True, w_symmetric=True)
    try:
        os.environ["DKT_GRAM_BIG_EP"] = "0"
        os.environ["DKT_GRAM_BWD_ROWS8"] = "0"
        e_old = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
        dz_old = ops.gram_bwd(w, z, eps, unit_rows=True, w_symmetric=True)
    finally:
        for key in ("DKT_GRAM_BIG_EP", "DKT_GRAM_BWD_ROWS8"):
            os.environ.pop(key, None)
    assert torch.equal(e_new, e_new.transpose(1, 2))
    assert (torch.diagonal(e_new, dim1=1, dim2=2) - 1.0).abs().max().item() < 2e-6
    assert (e_new - e_old).abs().max().item() < 2e-6           # (different summation orders of the same split products; the float64 comparison below is the accuracy bar)
    assert float((dz_new - dz_old).norm() / dz_old.norm()) < 1e-6
    assert float(((dz_new - dz_old).norm(dim=2) / dz_old.norm(dim=2).clamp_min(1e-30)).max()) < 5e-6       # row by row: every row has its own scale
    for bi in (0, b - 1):
        z64 = z[bi].double().cpu().numpy()
        ref = z64 @ z64.T
        mag = np.abs(z64) @ np.abs(z64).T
        assert (np.abs(e_new[bi].cpu().numpy() - ref) / mag).max() < 1.5e-6   # (fp32 accumulation over D / 32 slabs; the split itself carries 22 bits)
        dref = 2.0 * float(eps[bi].item()) * w[bi].double().cpu().numpy() @ z64
        assert rel_l2(dz_new[bi].cpu().numpy(), dref) < 2e-6
        rows = np.linalg.norm(dz_new[bi].cpu().numpy() - dref, axis=1) / np.linalg.norm(dref, axis=1)
        assert rows.max() < 1e-5, rows.max()


@pytest.mark.parametrize("n,d", [(129, 64), (190, 512), (320, 512), (420, 512), (257, 100)])
def test_gram_large_n_unit_rows_kernel(cuda, n, d):
    """Symmetric linear Gram at N > 128 with the unit-row promise: the 64 x 64-tile f16-split kernel (dkt_gram_big.hip) against float64
    and the exact-fp32 generic kernel (same call without the promise); exactly symmetric, unit diagonal."""
    rng = np.random.default_rng(n + d)
    z = rng.standard_normal((6, n, d)) * np.exp(1.5 * rng.standard_normal((6, n, d)))          # heavy-tailed before the normalisation
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    zt = dev_t(z, cuda)
    z = zt.double().cpu().numpy()                           # the fp32 values the kernels see
    ref = np.einsum("bnd,bmd->bnm", z, z)
    mag = np.einsum("bnd,bmd->bnm", np.abs(z), np.abs(z))
    fast = ops.gram(zt, kind=ops.KERNEL_LINEAR_UNIT)                      # 2-way f16 split (unit rows)
    mid = ops.gram(zt)                                                    # exact 3-way bf16 split (any range), same tile kernel
    os.environ["DKT_GRAM_SPLIT"] = "0"
    try:
        slow = ops.gram(zt)                                               # generic exact-fp32 MFMA kernel
    finally:
        os.environ.pop("DKT_GRAM_SPLIT")
    assert torch.equal(fast, fast.transpose(1, 2)) and torch.equal(mid, mid.transpose(1, 2))
    assert (np.abs(fast.cpu().numpy() - ref) / mag).max() < 6e-7          # measured <= 4.9e-7 (heavy-tailed rows), 2.5e-7 (Gaussian rows)
    assert (np.abs(mid.cpu().numpy() - ref) / mag).max() < 6e-7
    assert (np.abs(slow.cpu().numpy() - ref) / mag).max() < 3e-6          # the sequential fp32 chain: measured up to 2.0e-6
    zs = zt * 37.5                                                        # rows far from unit norm: only the range-free split may be used
    big = ops.gram(zs)
    assert (np.abs(big.cpu().numpy() - 37.5 * 37.5 * ref) / (37.5 * 37.5 * mag)).max() < 6e-7
    assert np.abs(np.diagonal(fast.cpu().numpy(), axis1=1, axis2=2) - 1.0).max() < 2e-6


@pytest.mark.parametrize("n,d", [(129, 64), (190, 512), (320, 512), (420, 512), (448, 36), (257, 100)])
def test_gram_bwd_large_n_symmetric_w_kernel(cuda, n, d):
    """128 < N <= 448 with unit rows and a W declared symmetric: the row-block f16-split kernel (dkt_gram_big.hip) against float64
    and against the generic kernel (same call without the two promises)."""
    rng = np.random.default_rng(n * d)
    b = 5
    z = rng.standard_normal((b, n, d))
    z /= np.linalg.norm(z, axis=2, keepdims=True)
    w = rng.standard_normal((b, n, n)) * np.exp(rng.standard_normal((b, n, 1)))       # rows of very different magnitude
    w = 0.5 * (w + w.transpose(0, 2, 1))
    s = rng.uniform(0.5, 2.0, b)
    ref = s[:, None, None] * np.einsum("bij,bjd->bid", w + w.transpose(0, 2, 1), z)
    zt, wt, st = dev_t(z, cuda), dev_t(w, cuda), dev_t(s, cuda)
    fast = ops.gram_bwd(wt, zt, st, unit_rows=True, w_symmetric=True)
    slow = ops.gram_bwd(wt, zt, st)
    mag = s[:, None, None] * np.einsum("bij,bjd->bid", np.abs(w + w.transpose(0, 2, 1)), np.abs(z))
    assert (np.abs(fast.cpu().numpy() - ref) / mag).max() < 4e-7
    assert (np.abs(slow.cpu().numpy() - ref) / mag).max() < 2e-6          # the sequential fp32 chain of the generic kernel is the less accurate one
    assert rel_l2(fast.cpu().numpy(), ref) < 2e-6


def test_duplicate_rows_rank_deficient_gram(cuda):
    g = np.load(os.path.join(GOLD, "degenerate.npz"))
    z = g["z"]
    y = O.one_vs_rest_targets(2, 6)
    o = ops.mll(ops.gram(dev_t(z[None], cuda)), dev_t(y, cuda), dev_t([0.7, 1.1], cuda), dev_t([0.05, -0.02], cuda),
                dev_t([0.1, 0.1], cuda))
    assert int(o["info"].abs().max().item()) == 0
    np.testing.assert_allclose(o["logp"][0].cpu().numpy(), g["logp_dup"], rtol=MLL_RTOL)
    assert rel_l2(o["alpha"][0].cpu().numpy(), g["alpha_dup"]) < 1e-4


# ----------------------------------------------------------------------------------------------
# golden vectors through the fused autograd entry point
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "cfg[1-9]*.npz"))))
def test_golden_training_episode_fused(cuda, path):
    g = np.load(path)
    c, s, q, d = int(g["n_way"]), int(g["n_support"]), int(g["n_query"]), int(g["d"])
    n = c * (s + q)
    z = O.synthetic_features(1, n, d, int(g["seed"]), int(g["correlated"]))[0]
    assert abs(z.sum() - float(g["z_checksum"])) < 1e-9
    zt = dev_t(z[None], cuda).requires_grad_(True)
    raw_s = dev_t(O.inv_softplus(g["outputscale"]), cuda).requires_grad_(True)
    mean = dev_t(g["mean"], cuda).requires_grad_(True)
    noise = dev_t(g["noise"], cuda)
    y = dev_t(O.one_vs_rest_targets(c, s + q), cuda)
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    sv = torch.nn.functional.softplus(raw_s)
    obj, logp, alpha, info, jit, e = ops.episode_loss_linear(zt, y, sv, mean, noise, cw)
    loss = obj.mean()
    loss.backward()
    assert int(info.abs().max().item()) == 0
    assert abs(loss.item() - float(g["loss"])) < MLL_RTOL * abs(float(g["loss"]))
    assert np.abs((logp[0].cpu().numpy() - g["logp"]) / g["logp"]).max() < MLL_RTOL
    assert rel_l2(alpha[0].cpu().numpy(), g["alpha"]) < 1e-4
    dz = zt.grad[0].cpu().numpy().astype(np.float64)
    assert abs(np.linalg.norm(dz) - float(g["dz_fro"])) < GRAD_RTOL * float(g["dz_fro"])
    assert rel_l2(dz[:3], g["dz_rows"]) < GRAD_RTOL
    # chain rule through softplus: d loss / d raw_s = dsv * sigmoid(raw)
    ref_raw = g["dsv"] * O.sigmoid(O.inv_softplus(g["outputscale"]))
    assert rel_l2(raw_s.grad.cpu().numpy(), ref_raw) < GRAD_RTOL
    assert rel_l2(mean.grad.cpu().numpy(), g["dmean"]) < GRAD_RTOL


def test_golden_rbf_episode_autograd(cuda):
    g = np.load(os.path.join(GOLD, "small_3w2s_rbf.npz"))
    c, per = int(g["n_way"]), int(g["n_support"]) + int(g["n_query"])
    z = g["z"]
    zt = dev_t(z[None], cuda).requires_grad_(True)
    ls = dev_t([float(g["lengthscale"])], cuda).requires_grad_(True)
    sv = dev_t(g["outputscale"], cuda).requires_grad_(True)
    mean = dev_t(g["mean"], cuda).requires_grad_(True)
    noise = dev_t(g["noise"], cuda).requires_grad_(True)
    n = c * per
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    e = ops.base_matrix(zt, "rbf", ls)
    obj, logp, alpha, info, jit = ops.mll_objective(e, dev_t(O.one_vs_rest_targets(c, per), cuda), sv, mean, noise, cw)
    obj.mean().backward()
    assert abs(obj.item() - float(g["loss"])) < MLL_RTOL * abs(float(g["loss"]))
    hyp = O.GPHypers(g["outputscale"], g["mean"], g["noise"], float(g["lengthscale"]))
    ref = O.train_episode(z, c, hyp, "rbf")
    assert rel_l2(zt.grad[0].cpu().numpy(), ref["dz"]) < GRAD_RTOL
    assert abs(ls.grad.item() - ref["dlengthscale"]) < GRAD_RTOL * abs(ref["dlengthscale"])
    assert rel_l2(sv.grad.cpu().numpy(), ref["dsv"]) < GRAD_RTOL
    assert rel_l2(noise.grad.cpu().numpy(), ref["dnoise"]) < GRAD_RTOL
    assert rel_l2(mean.grad.cpu().numpy(), ref["dmean"]) < GRAD_RTOL


def test_golden_cfg0_qmul_regression_head(cuda):
    """BASELINE.json configs[0] at its real shape ((19, 2916, 1 GP, RBF) training task and the 5 -> 19 test shape,
    methods/DKT_regression.py:45-97) against the committed fixture tests/golden/cfg0_qmul_regression_rbf.npz (float64 oracle,
    cross-checked against scikit-learn and scipy when it was generated): through the raw ops AND through the DKTRegression head."""
    sys.path.insert(0, GOLD)
    from make_golden import cfg0_features
    g = np.load(os.path.join(GOLD, "cfg0_qmul_regression_rbf.npz"))
    z, labels = cfg0_features(int(g["seed"]), int(g["n"]), int(g["d"]))
    assert abs(z.sum() - float(g["z_checksum"])) < 1e-9 * abs(float(g["z_checksum"])) and np.allclose(labels, g["labels"])
    n = z.shape[0]
    zt = dev_t(z[None], cuda).requires_grad_(True)
    ls = dev_t([float(g["lengthscale"])], cuda).requires_grad_(True)
    sv, mean, noise = (dev_t(g[k], cuda).requires_grad_(True) for k in ("outputscale", "mean", "noise"))
    cw = torch.full((1,), -1.0 / n, device=cuda)
    e = ops.base_matrix(zt, "rbf", ls)
    assert np.abs(e[0, :3].detach().cpu().numpy() - g["e_rows"]).max() < 2e-6
    obj, logp, alpha, info, jit = ops.mll_objective(e, dev_t(labels[None, None], cuda), sv, mean, noise, cw)
    obj.mean().backward()
    assert int(info.abs().max().item()) == 0
    assert abs(obj.item() - float(g["loss"])) < MLL_RTOL * abs(float(g["loss"]))
    assert abs(logp.item() - float(g["logp"][0])) < MLL_RTOL * abs(float(g["logp"][0]))
    assert rel_l2(alpha[0].cpu().numpy(), g["alpha"]) < 5e-4
    assert rel_l2(zt.grad[0, :3].cpu().numpy(), g["dz_rows"]) < GRAD_RTOL
    assert abs(float(zt.grad[0].norm()) - float(g["dz_fro"])) < GRAD_RTOL * float(g["dz_fro"])
    assert abs(ls.grad.item() - float(g["dlengthscale"])) < GRAD_RTOL * abs(float(g["dlengthscale"]))
    assert rel_l2(sv.grad.cpu().numpy(), g["dsv"]) < GRAD_RTOL             # the fixture's gradients are those of the loss -logp / N
    assert rel_l2(noise.grad.cpu().numpy(), g["dnoise"]) < GRAD_RTOL
    assert rel_l2(mean.grad.cpu().numpy(), g["dmean"]) < GRAD_RTOL
    # the DKTRegression head: same numbers through _loss / predict (the backbone replaced by the fixture's features)
    m = dkt_amd.DKTRegression(dkt_amd.backbone.Conv3(), "rbf").to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.fill_(dkt_amd.gp.inv_softplus(float(g["outputscale"][0])))
        m.model.mean_constant.fill_(float(g["mean"][0]))
        m.model.raw_noise.fill_(dkt_amd.gp.inv_softplus(float(g["noise"][0]) - dkt_amd.gp.NOISE_LOWER_BOUND))
        m.model.raw_lengthscale.fill_(dkt_amd.gp.inv_softplus(float(g["lengthscale"])))
    loss, aux = m._loss(dev_t(z, cuda), dev_t(labels, cuda))
    assert abs(loss.item() - float(g["loss"])) < MLL_RTOL * abs(float(g["loss"]))
    sup = g["support"].tolist()
    with torch.no_grad():
        mu, var = m.predict(dev_t(z[sup], cuda), dev_t(labels[sup], cuda), dev_t(z, cuda), with_variance=True)
    assert np.abs(mu.cpu().numpy() - g["pred_mean"]).max() < 1e-4
    assert np.abs(var.cpu().numpy() - g["pred_var"]).max() < 1e-4 * np.abs(g["pred_var"]).max()


@pytest.mark.parametrize("b,n,d", [(2, 5, 12), (2, 85, 512), (3, 105, 64), (2, 105, 1600), (1, 130, 36), (1, 420, 512)])
def test_gram_bwd_vs_oracle(cuda, b, n, d):
    rng = np.random.default_rng(n + d)
    w = rng.standard_normal((b, n, n)).astype(np.float32)       # deliberately NOT symmetric
    z = rng.standard_normal((b, n, d)).astype(np.float32)
    sc = rng.standard_normal(b).astype(np.float32)
    dz = ops.gram_bwd(dev_t(w, cuda), dev_t(z, cuda), dev_t(sc, cuda)).cpu().numpy()
    for i in range(b):
        ref = sc[i] * O.gram_linear_bwd(w[i].astype(np.float64), z[i].astype(np.float64))
        assert rel_l2(dz[i], ref) < 1e-5


# ----------------------------------------------------------------------------------------------
# prediction
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "test_*.npz"))))
def test_golden_test_episode_prediction(cuda, path):
    g = np.load(path)
    c, s, q, d = int(g["n_way"]), int(g["n_support"]), int(g["n_query"]), int(g["d"])
    zall = O.synthetic_features(1, c * (s + q), d, int(g["seed"]), c)[0].reshape(c, s + q, d)
    zs, zq = zall[:, :s].reshape(1, c * s, d), zall[:, s:].reshape(1, c * q, d)
    sv, mean, noise = dev_t(g["outputscale"], cuda), dev_t(g["mean"], cuda), dev_t(g["noise"], cuda)
    zs_t, zq_t = dev_t(zs, cuda), dev_t(zq, cuda)
    out = ops.mll(ops.gram(zs_t), dev_t(O.one_vs_rest_targets(c, s), cuda), sv, mean, noise)
    mu, labels = ops.predict(ops.gram(zq_t, zs_t), out["alpha"], sv, mean)
    assert np.abs(mu[0].cpu().numpy() - g["mu"]).max() < 1e-4
    assert (labels[0].cpu().numpy() == g["labels"]).all(), "predicted labels must be identical"
    assert float((labels[0].cpu().numpy() == np.repeat(np.arange(c), q)).sum()) == float(g["correct"])


def test_batched_test_episodes_single_pass_gram(cuda):
    """bench.py's test-time formulation: the symmetric Gram of [support; query] (episode-resident kernel, B >= 64) holds
    k(s, s) and k(q, s); posterior means / labels equal the two-Gram path and the oracle."""
    b, c, s_, q_, d = 64, 5, 5, 16, 256
    ns, n = c * s_, c * (s_ + q_)
    g = torch.Generator(device=cuda).manual_seed(4)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=cuda) + 2.0 * torch.randn(b, 1, d, generator=g, device=cuda), dim=2)
    hyp = O.perturbed_hypers(c, 5)
    sv, mean, noise = dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda)
    ys = dev_t(O.one_vs_rest_targets(c, s_), cuda)
    e_all = ops.gram(z)
    o1 = ops.mll(e_all[:, :ns, :ns].contiguous(), ys, sv, mean, noise)
    mu1, lab1 = ops.predict(e_all[:, ns:, :ns].contiguous(), o1["alpha"], sv, mean)
    zs, zq = z[:, :ns].contiguous(), z[:, ns:].contiguous()
    o2 = ops.mll(ops.gram(zs), ys, sv, mean, noise)
    mu2, lab2 = ops.predict(ops.gram(zq, zs), o2["alpha"], sv, mean)
    assert (mu1 - mu2).abs().max().item() < 2e-5
    for i in (0, 31, 63):
        ref = O.eval_episode(zs[i].cpu().numpy().astype(np.float64), zq[i].cpu().numpy().astype(np.float64), c, hyp)
        assert np.abs(mu1[i].cpu().numpy() - ref["mu"]).max() < 1e-4
        assert (lab1[i].cpu().numpy() == ref["labels"]).all()


def test_predict_first_max_wins_on_ties(cuda):
    ex = torch.ones(1, 3, 2, device=cuda)
    alpha = torch.zeros(1, 4, 2, device=cuda)
    mean = dev_t([0.5, 0.7, 0.7, 0.1], cuda)
    mu, labels = ops.predict(ex, alpha, torch.ones(4, device=cuda), mean)
    assert labels.cpu().tolist() == [[1, 1, 1]]


def test_predict_variance_vs_oracle(cuda):
    rng = np.random.default_rng(4)
    z = np.abs(rng.standard_normal((19, 60)))
    y = rng.standard_normal(19)
    sup = [1, 3, 8, 12, 17]
    hyp = O.GPHypers(np.array([0.8]), np.array([0.1]), np.array([0.3]), lengthscale=4.0)
    ref = O.regression_predict(z[sup], y[sup], z, hyp)
    ls = dev_t([4.0], cuda)
    zs, za = dev_t(z[sup][None], cuda), dev_t(z[None], cuda)
    sv, mean, noise = dev_t([0.8], cuda), dev_t([0.1], cuda), dev_t([0.3], cuda)
    out = ops.mll(ops.gram(zs, None, ops.KERNEL_RBF, ls), dev_t(y[sup][None, None], cuda), sv, mean, noise, want_chol=True)
    ex = ops.gram(za, zs, ops.KERNEL_RBF, ls)
    mu, _ = ops.predict(ex, out["alpha"], sv, mean, want_labels=False)
    var = ops.predict_var(ex, torch.ones(1, 19, device=cuda), out["chol"], sv, noise)
    assert np.abs(mu[0, 0].cpu().numpy() - ref["mean"]).max() < 1e-4
    assert np.abs(var[0, 0].cpu().numpy() - ref["var"]).max() < 1e-4


# ----------------------------------------------------------------------------------------------
# full-size properties (BASELINE.json cfg2 at bench batch): no oracle pass over 1024 episodes needed
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("unit", [False, True], ids=["bf16_split", "unit_rows_f16_split"])
@pytest.mark.parametrize("cfg", ["cfg2", "cfg1", "cfg3", "cfg1_nxn"])      # cfg1: the feature-space path (D = 64 < N, round 5); cfg1_nxn: its N x N twin (DKT_LOWRANK=0)
def test_full_size_properties_cfg2(cuda, unit, cfg, monkeypatch):
    """BASELINE.json configs[1..3] at (N, D, C) full size and a batch the headline kernels take (>= 64 episodes): cfg2 = (105, 1600, 5),
    cfg1 = (105, 64, 5) Conv4S features, cfg3 = (85, 512, 5) 5-way 1-shot ResNet10 features."""
    # b = 1024 >= DKT_MLL_H2E_MINB: the DEFAULT dispatch takes the wave-per-episode kernels the bench times (mll_h2e_kernel<7> / <6>), not
    # the wave-per-matrix kernel of smaller batches
    monkeypatch.setenv("DKT_LOWRANK", "0" if cfg == "cfg1_nxn" else "force")      # (1024 episodes of 105 rows: below the batch from which the default takes feature space)
    b, n, d, c = {"cfg2": (1024, 105, 1600, 5), "cfg1": (1024, 105, 64, 5), "cfg1_nxn": (1024, 105, 64, 5), "cfg3": (1024, 85, 512, 5)}[cfg]
    gen = torch.Generator(device="cpu").manual_seed(1234)
    zr = torch.randn(b, n, d, generator=gen)
    zr = (zr - zr.mean(1, keepdim=True)) / torch.sqrt(zr.var(1, unbiased=False, keepdim=True) + 1e-5)
    z = torch.nn.functional.normalize(zr, dim=2).to(cuda).requires_grad_(True)
    y = dev_t(O.one_vs_rest_targets(c, n // c), cuda)
    hyp = O.perturbed_hypers(c, 7)
    sv, mean, noise = dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda)
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=unit)
    obj.sum().backward()
    assert int(info.abs().max().item()) == 0 and torch.isfinite(logp).all()
    assert (e is None) == (cfg == "cfg1")                # cfg1 runs in feature space: no N x N matrix exists (the residual check below builds one)
    if e is None:
        e = ops.gram(z.detach(), None, ops.KERNEL_LINEAR_UNIT if unit else ops.KERNEL_LINEAR)
    assert torch.allclose(torch.diagonal(e, dim1=1, dim2=2), torch.ones(b, n, device=cuda), atol=2e-6)
    assert torch.equal(e, e.transpose(1, 2))
    # K alpha = y - m for every episode / class
    k = sv.view(1, c, 1, 1) * e.unsqueeze(1) + noise.view(1, c, 1, 1) * torch.eye(n, device=cuda)
    resid = torch.matmul(k.double(), alpha.double().unsqueeze(-1)).squeeze(-1) - (y.double().unsqueeze(0) - mean.double().view(1, c, 1))
    assert resid.abs().max().item() < 2e-4
    # determinism: a second launch is bitwise identical (no atomics anywhere)
    z2 = z.detach().clone().requires_grad_(True)
    obj2, logp2, *_ = ops.episode_loss_linear(z2, y, sv, mean, noise, cw, unit_rows=unit)
    obj2.sum().backward()
    assert torch.equal(logp, logp2) and torch.equal(z.grad, z2.grad)
    # spot-check three episodes against the oracle at full size
    for i in (0, 100, b - 1):
        ref = O.train_episode(z[i].detach().cpu().numpy().astype(np.float64), c, hyp)
        assert np.abs((logp[i].cpu().numpy() - ref["logp"]) / ref["logp"]).max() < MLL_RTOL
        assert rel_l2(z.grad[i].cpu().numpy(), ref["dz"]) < GRAD_RTOL
    # linearity of the backward in the upstream gradient
    z3 = z.detach().clone().requires_grad_(True)
    obj3, *_ = ops.episode_loss_linear(z3, y, sv, mean, noise, cw, unit_rows=unit)
    (2.5 * obj3.sum()).backward()
    # (the scale is folded into the MFMA A operand, so the two results differ by fp32 rounding amplified by the
    # cancellation between the alpha alpha^T and K^-1 parts of W: compare in relative L2)
    assert float((z3.grad - 2.5 * z.grad).norm() / (2.5 * z.grad).norm()) < 1e-4


LOWRANK_SHAPES = [(3, 5, 21, 64, 0), (2, 5, 21, 64, 5), (2, 20, 21, 64, 0), (2, 20, 21, 64, 20), (3, 5, 16, 32, 0), (2, 3, 37, 60, 3), (1, 17, 5, 64, 0), (2, 32, 3, 8, 0),
                  (2, 1, 97, 64, 0), (1, 5, 100, 64, 0)]


@pytest.mark.parametrize("b,c,per,d,corr", LOWRANK_SHAPES)
def test_episode_in_feature_space_vs_oracle_and_nxn_twin(cuda, b, c, per, d, corr, monkeypatch):
    """Linear kernels with D <= 64 < N (Conv4S / Omniglot: D = 64, N = 105 5-way / 420 20-way; backbone.py:287-310, train.py:132): the episode through
    dkt_lowrank_gram_f32 -> dkt_mll_f32 on the 64 x 64 models K'_c = sv_c Z^T Z + noise_c I -> dkt_lowrank_finish_f32 -> dkt_lowrank_bwd_f32, no N x N
    matrix anywhere.  Against the float64 oracle (log-likelihood 1e-4, gradients 1e-3: the tolerances of the N x N path) on plain and on class-correlated
    features (cond(K) ~ 10^2..10^3), with per-episode targets, class weights of both signs, an upstream gradient per episode, D < 64 (zero-padded), C > 16
    (two class tiles), N not a multiple of 4 or 16; and against its N x N twin (DKT_LOWRANK=0: dkt_gram_f32 / dkt_mll_f32 / dkt_gram_bwd_f32)."""
    n = c * per
    rng = np.random.default_rng(100 * c + per + d)
    z64 = O.synthetic_features(b, n, d, 700 + c + d, corr)
    hyp = O.perturbed_hypers(c, 31 + c)
    cw = np.full(c, -1.0 / (c * n))
    if c > 1:
        cw[c // 2] *= -1.5
    gup = rng.uniform(0.5, 2.0, b)
    y = O.one_vs_rest_targets(c, per)

    def run():
        z = dev_t(z64, cuda).requires_grad_(True)
        sv, mean, noise = (dev_t(v, cuda).requires_grad_(True) for v in (hyp.outputscale, hyp.mean, hyp.noise))
        out = ops.episode_loss_linear(z, dev_t(y, cuda), sv, mean, noise, dev_t(cw, cuda), unit_rows=True)
        (out[0] * dev_t(gup, cuda)).sum().backward()
        return out, z.grad, sv.grad, mean.grad, noise.grad

    monkeypatch.setenv("DKT_LOWRANK", "force")             # small batches: the default dispatch takes feature space from 3072 episodes of <= 128 rows, always above 128 rows
    assert ops.lowrank_applies(n, d, c, b) and ops.lowrank_supported(n, d, c)
    (obj, logp, alpha, info, jit, e), dz, gsv, gmean, gnoise = run()
    assert e is None and int(info.abs().max().item()) == 0 and float(jit.abs().max()) == 0.0
    monkeypatch.setenv("DKT_LOWRANK", "0")
    (obj_t, logp_t, alpha_t, info_t, _, e_t), dz_t, gsv_t, gmean_t, gnoise_t = run()
    assert e_t is not None and tuple(e_t.shape) == (b, n, n)
    ref_gsv, ref_gmean, ref_gnoise = np.zeros(c), np.zeros(c), np.zeros(c)
    for i in range(b):
        zi = z64[i].astype(np.float32).astype(np.float64)
        e64 = zi @ zi.T
        res = O.mll_terms(e64, y, hyp.outputscale, hyp.mean, hyp.noise)
        assert np.abs((logp[i].cpu().numpy() - res.logp) / res.logp).max() < MLL_RTOL
        assert rel_l2(alpha[i].cpu().numpy(), res.alpha) < 5e-4
        w_ref, dsv, dmean, dnoise = O.mll_grads(e64, res, hyp.outputscale, hyp.noise, cw)
        assert rel_l2(dz[i].cpu().numpy(), gup[i] * O.gram_linear_bwd(w_ref, zi)) < GRAD_RTOL
        ref_gsv += gup[i] * dsv
        ref_gmean += gup[i] * dmean
        ref_gnoise += gup[i] * dnoise
        assert abs(obj[i].item() - float((cw * res.logp).sum())) < MLL_RTOL * np.abs(cw * res.logp).sum()
    assert rel_l2(gsv.cpu().numpy(), ref_gsv) < GRAD_RTOL and rel_l2(gnoise.cpu().numpy(), ref_gnoise) < GRAD_RTOL
    assert np.abs(gmean.cpu().numpy() - ref_gmean).max() < GRAD_RTOL * np.abs(ref_gmean).max() + 1e-5
    # the N x N twin: same tolerances to each other as each has to the oracle
    assert rel_l2(logp.cpu().numpy(), logp_t.cpu().numpy()) < 2e-5 and rel_l2(alpha.cpu().numpy(), alpha_t.cpu().numpy()) < 5e-4
    assert rel_l2(dz.cpu().numpy(), dz_t.cpu().numpy()) < GRAD_RTOL and rel_l2(gsv.cpu().numpy(), gsv_t.cpu().numpy()) < GRAD_RTOL
    monkeypatch.setenv("DKT_LOWRANK", "force")
    # bitwise repeatable (fixed reduction orders, no atomics)
    (_, logp2, alpha2, *_), dz2, *_ = run()
    assert torch.equal(logp, logp2) and torch.equal(alpha, alpha2) and torch.equal(dz, dz2)
    # the default dispatch: feature space for every batch of episodes with more than 128 rows, for shorter ones from LOWRANK_MIN_B episodes per call
    monkeypatch.delenv("DKT_LOWRANK")
    assert ops.lowrank_applies(n, d, c, b) == (n > 128) and ops.lowrank_applies(n, d, c, ops.LOWRANK_MIN_B) and ops.lowrank_applies(n, d, c, 8192, front_end=True) == (n > 128)


def test_episode_in_feature_space_jitter_and_failure(cuda, monkeypatch):
    """The jitter ladder of the feature-space path is the D x D call's: psd_safe_cholesky's total jitter 1e-6 * 10^i lands on noise_c, as it does on
    the diagonal of K_c in the reference.  noise = 0 makes K_c = sv Z Z^T singular (rank D < N): attempt 0 fails, 1e-6 succeeds -- per class, per
    episode; a NaN feature poisons only its own episode (info != 0 / NaN outputs), per-episode targets are honoured."""
    monkeypatch.setenv("DKT_LOWRANK", "force")
    b, c, per, d = 3, 5, 21, 64
    n = c * per
    z64 = O.synthetic_features(b, n, d, 41, 0)
    y = np.stack([O.one_vs_rest_targets(c, per) * (1.0 + 0.1 * i) for i in range(b)])          # [B, C, N]: per-episode targets
    sv, mean = np.linspace(0.5, 1.5, c), np.linspace(-0.1, 0.1, c)
    noise = np.array([0.1, 0.0, 0.2, 0.0, 0.05])
    cw = np.full(c, -1.0 / (c * n))
    z = dev_t(z64, cuda).requires_grad_(True)
    obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise, cuda), dev_t(cw, cuda), unit_rows=True)
    obj.sum().backward()
    assert e is None and int(info.abs().max().item()) == 0
    jit_np = jit.cpu().numpy()
    assert (jit_np[:, [0, 2, 4]] == 0.0).all() and np.allclose(jit_np[:, [1, 3]], 1e-6, rtol=1e-5)
    for i in range(b):
        zi = z64[i].astype(np.float32).astype(np.float64)
        for k in (0, 2, 4):                                  # the healthy classes against the oracle (the jittered ones are cond ~ 10^8: no fp32 reference)
            res = O.mll_terms(zi @ zi.T, y[i, k:k + 1], sv[k:k + 1], mean[k:k + 1], noise[k:k + 1])
            assert abs((logp[i, k].item() - res.logp[0]) / res.logp[0]) < MLL_RTOL
            assert rel_l2(alpha[i, k].cpu().numpy(), res.alpha[0]) < 5e-4
    assert torch.isfinite(logp).all() and torch.isfinite(z.grad).all()
    zb = dev_t(z64, cuda)
    zb[1, 7, 3] = float("nan")
    o2 = ops.episode_loss_linear(zb.requires_grad_(True), dev_t(y, cuda), dev_t(sv, cuda), dev_t(mean, cuda), dev_t(noise + 0.1, cuda), dev_t(cw, cuda), unit_rows=True)
    assert (o2[3][1] != 0).all() and torch.isnan(o2[1][1]).all() and torch.isnan(o2[2][1]).all()
    assert int(o2[3][[0, 2]].abs().max().item()) == 0 and torch.isfinite(o2[1][[0, 2]]).all()


def test_dkt_omniglot_shape_train_step_runs_in_feature_space(cuda, capsys, monkeypatch):
    """The drop-in class at the Omniglot shape (Conv4S trunk: D = 64, backbone.py:287-310; 5-way 5-shot + 16 queries: N = 105; cossim = the un-fused
    front end): the train step takes the feature-space path (aux['e'] is None) and its loss / parameter gradients match the N x N twin; the in-loop
    evaluation at a print point (which conditions on the stale train features, DKT.py:170-192) builds its own Gram."""
    monkeypatch.setenv("DKT_LOWRANK", "force")             # (one 105-row episode: the default dispatch keeps it on the N x N kernels -- both steps are launch-bound there)
    torch.manual_seed(3)
    model = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5, kernel_type="cossim").to(cuda)
    model.train()
    x = torch.rand(5, 21, 3, 28, 28, generator=torch.Generator().manual_seed(4))
    y_t = model._targets(5, 21, cuda)
    loss, aux = model._episode_loss(model._embed(x.view(105, 3, 28, 28).to(cuda)), y_t)
    assert aux["e"] is None and int(aux["info"].abs().max().item()) == 0
    params = [p_ for p_ in model.feature_extractor.parameters() if p_.requires_grad]
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    monkeypatch.setenv("DKT_LOWRANK", "0")
    loss2, aux2 = model._episode_loss(model._embed(x.view(105, 3, 28, 28).to(cuda)), y_t)
    grads2 = torch.autograd.grad(loss2, params, allow_unused=True)
    monkeypatch.setenv("DKT_LOWRANK", "force")
    assert aux2["e"] is not None and abs(loss.item() - loss2.item()) < 1e-5 * abs(loss2.item())
    for g1, g2 in zip(grads, grads2):
        assert (g1 is None) == (g2 is None)
        if g1 is not None:
            assert float((g1 - g2).norm()) <= GRAD_RTOL * float(g2.norm()) + 2e-5        # (conv biases in front of a train-mode BN have an exactly-zero gradient: absolute floor)
    model.train_loop(0, [(x, None)] * 2, None, print_freq=1)
    assert "Epoch [0] [0/2]" in capsys.readouterr().out


def test_bench_batch_cfg0_regression_head_vs_oracle(cuda):
    """BASELINE.json configs[0] (QMUL regression: one GP per task on (19, 2916) Conv3 features, RBF kernel, learned noise) at a batch the
    bench's dispatch takes (1024 tasks: gram_small_kernel / gram_small_bwd_kernel + mll_h2e_kernel<2> + rbf_bwd_kernel), through the calls
    bench.py makes (ops.base_matrix -> ops.mll_objective), against O.regression_episode on sampled tasks; bitwise determinism."""
    b, n, d = 1024, 19, 2916
    g = torch.Generator(device="cpu").manual_seed(99)
    z = (torch.randn(b, n, d, generator=g).abs() * 0.35).to(cuda).requires_grad_(True)      # the fixture's feature scale (make_golden.cfg0_features)
    yb = (torch.rand(b, 1, n, generator=g) * 2.0 - 1.0).to(cuda)
    sv = torch.tensor([0.8], device=cuda, requires_grad=True)
    mean = torch.tensor([0.05], device=cuda, requires_grad=True)
    ls = torch.tensor([11.0], device=cuda, requires_grad=True)                               # off-diagonal kernel values ~ 0.3 at this feature scale
    nz = torch.tensor([0.6932], device=cuda, requires_grad=True)
    cw = torch.full((1,), -1.0 / n, device=cuda)

    def run(zz):
        e = ops.base_matrix(zz, "rbf", ls)
        obj, logp, alpha, info, jit = ops.mll_objective(e, yb, sv, mean, nz, cw)
        obj.sum().backward()
        return obj, logp, info
    obj, logp, info = run(z)
    assert int(info.abs().max().item()) == 0 and torch.isfinite(logp).all()
    dls = ls.grad.clone()
    z2 = z.detach().clone().requires_grad_(True)
    ls.grad = None
    obj2, logp2, _ = run(z2)
    assert torch.equal(logp, logp2) and torch.equal(z.grad, z2.grad)
    hyp = O.GPHypers(np.array([0.8]), np.array([0.05]), np.array([0.6932]), lengthscale=11.0)
    for i in (0, 511, 1023):
        ref = O.regression_episode(z[i].detach().cpu().numpy().astype(np.float64), yb[i, 0].cpu().numpy().astype(np.float64), hyp, "rbf")
        assert abs((logp[i, 0].item() - ref["logp"][0]) / ref["logp"][0]) < MLL_RTOL
        assert abs(obj[i].item() - float(np.ravel(ref["loss"])[0])) < MLL_RTOL * abs(float(np.ravel(ref["loss"])[0]))
        assert rel_l2(z.grad[i].cpu().numpy(), ref["dz"]) < GRAD_RTOL
    assert torch.isfinite(dls).all()


def test_bench_batch_cfg4_full_chunk_and_ragged_tail(cuda):
    """BASELINE.json configs[4] at the batch the bench runs: one FULL 1024-episode chunk of the tile-array marginal likelihood plus a
    ragged second chunk of 6 (N = 420, C = 20, D = 512), default dispatch: residual K alpha = y - m on the first / last episodes of the
    full chunk and on the tail, symmetric W, oracle spot checks in both chunks, bitwise determinism of a second call."""
    b, d, c, per = 1030, 512, 20, 21
    n = c * per
    gen = torch.Generator(device=cuda).manual_seed(77)
    zr = torch.randn(b, n, d, generator=gen, device=cuda)
    zr = (zr - zr.mean(1, keepdim=True)) / torch.sqrt(zr.var(1, unbiased=False, keepdim=True) + 1e-5)
    z = torch.nn.functional.normalize(zr, dim=2).contiguous().requires_grad_(True)
    del zr
    y = dev_t(O.one_vs_rest_targets(c, per), cuda)
    hyp = O.perturbed_hypers(c, 9)
    sv, mean, noise = dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda)
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=True)
    obj.sum().backward()
    assert int(info.abs().max().item()) == 0 and torch.isfinite(logp).all() and float(jit.abs().max()) == 0.0
    for lo, hi in ((0, 24), (1000, 1030)):
        k = sv.view(1, c, 1, 1) * e[lo:hi].unsqueeze(1) + noise.view(1, c, 1, 1) * torch.eye(n, device=cuda)
        r = torch.matmul(k.double(), alpha[lo:hi].double().unsqueeze(-1)).squeeze(-1) - (y.double().unsqueeze(0) - mean.double().view(1, c, 1))
        assert r.abs().max().item() < 5e-4
        del k, r
    z2 = z.detach().clone().requires_grad_(True)
    obj2, logp2, *_ = ops.episode_loss_linear(z2, y, sv, mean, noise, cw, unit_rows=True)
    obj2.sum().backward()
    assert torch.equal(logp, logp2) and torch.equal(z.grad, z2.grad)
    for i in (3, 1023, 1029):                      # inside the full chunk, its last episode, the ragged tail
        ref = O.train_episode(z[i].detach().cpu().numpy().astype(np.float64), c, hyp)
        assert np.abs((logp[i].cpu().numpy() - ref["logp"]) / ref["logp"]).max() < MLL_RTOL
        assert rel_l2(z.grad[i].cpu().numpy(), ref["dz"]) < GRAD_RTOL


@pytest.mark.parametrize("n_way,per", [(20, 21), (20, 16)], ids=["cfg4_n420", "cfg4_n320"])
def test_full_size_properties_cfg4(cuda, n_way, per):
    """BASELINE.json configs[4] (20-way, ResNet18 features D = 512; N = 420 as train_loop builds it and the 320 x 320 Gram the
    config quotes) at a full chunk of the blocked large-N path plus a ragged second chunk: residual K alpha = y - m for every
    episode and class, symmetric unit-diagonal Gram, bitwise determinism, linearity of the backward, oracle spot checks --
    through the asynchronous (no host read-back) call."""
    b, d, c = 132, 512, n_way                      # 128 = one workspace chunk, + 4
    n = c * per
    gen = torch.Generator(device="cpu").manual_seed(4321)
    zr = torch.randn(b, n, d, generator=gen)
    zr = (zr - zr.mean(1, keepdim=True)) / torch.sqrt(zr.var(1, unbiased=False, keepdim=True) + 1e-5)
    z = torch.nn.functional.normalize(zr, dim=2).to(cuda).requires_grad_(True)
    y = dev_t(O.one_vs_rest_targets(c, per), cuda)
    hyp = O.perturbed_hypers(c, 9)
    sv, mean, noise = dev_t(hyp.outputscale, cuda), dev_t(hyp.mean, cuda), dev_t(hyp.noise, cuda)
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    obj, logp, alpha, info, jit, e = ops.episode_loss_linear(z, y, sv, mean, noise, cw, unit_rows=True)
    obj.sum().backward()
    assert int(info.abs().max().item()) == 0 and torch.isfinite(logp).all() and float(jit.abs().max()) == 0.0
    assert torch.allclose(torch.diagonal(e, dim1=1, dim2=2), torch.ones(b, n, device=cuda), atol=2e-6)
    assert torch.equal(e, e.transpose(1, 2))
    worst = 0.0
    for lo in range(0, b, 33):                     # K alpha = y - m, in slices (20 x 420 x 420 doubles per episode)
        hi = min(b, lo + 33)
        k = sv.view(1, c, 1, 1) * e[lo:hi].unsqueeze(1) + noise.view(1, c, 1, 1) * torch.eye(n, device=cuda)
        r = torch.matmul(k.double(), alpha[lo:hi].double().unsqueeze(-1)).squeeze(-1) - (y.double().unsqueeze(0) - mean.double().view(1, c, 1))
        worst = max(worst, r.abs().max().item())
    assert worst < 5e-4
    z2 = z.detach().clone().requires_grad_(True)
    obj2, logp2, *_ = ops.episode_loss_linear(z2, y, sv, mean, noise, cw, unit_rows=True)
    obj2.sum().backward()
    assert torch.equal(logp, logp2) and torch.equal(z.grad, z2.grad)
    for i in (0, 127, 131):                        # first chunk, its last episode, the ragged tail
        ref = O.train_episode(z[i].detach().cpu().numpy().astype(np.float64), c, hyp)
        assert np.abs((logp[i].cpu().numpy() - ref["logp"]) / ref["logp"]).max() < MLL_RTOL
        assert rel_l2(z.grad[i].cpu().numpy(), ref["dz"]) < GRAD_RTOL
    z3 = z.detach().clone().requires_grad_(True)
    obj3, *_ = ops.episode_loss_linear(z3, y, sv, mean, noise, cw, unit_rows=True)
    (2.5 * obj3.sum()).backward()
    assert float((z3.grad - 2.5 * z.grad).norm() / (2.5 * z.grad).norm()) < 1e-4


def test_mll_bitwise_stable_next_to_split_gram_kernels(cuda):
    """dkt_mll_f32 on one stream while split Gram kernels run on another (co-resident workgroups on the same CUs): the result
    must be bitwise the single-stream result.  Regression test for the packed row-factor forms."""
    b, n, d, c = 8192, 105, 1600, 5
    g = torch.Generator(device=cuda).manual_seed(21)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, generator=g, device=cuda), dim=2).contiguous()
    y = dev_t(O.one_vs_rest_targets(c, n // c), cuda)
    sv = torch.full((c,), 0.69, device=cuda) + 0.01 * torch.arange(c, device=cuda)
    mean, noise = torch.zeros(c, device=cuda), torch.full((c,), 0.1, device=cuda)
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
    torch.cuda.synchronize()
    ref = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(4):
        with torch.cuda.stream(s2):
            for _ in range(3):
                ops.gram(z, None, ops.KERNEL_LINEAR if rep % 2 else ops.KERNEL_LINEAR_UNIT)
        with torch.cuda.stream(s1):
            a = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        torch.cuda.synchronize()
        for k in ("logp", "alpha", "w", "dsv", "dmean"):
            assert torch.equal(a[k], ref[k]), (rep, k, int((a[k] != ref[k]).sum()))


def test_gram_kernels_full_occupancy_are_race_free(cuda):
    """2048 episodes of the headline shape: every LDS stage buffer hand-off of the streaming kernels under full occupancy.
    Bitwise repeatable, and sampled episodes equal to float64."""
    b, n, d = 2048, 105, 1600
    g = torch.Generator(device=cuda).manual_seed(9)
    x = torch.randn(b, n, d, generator=g, device=cuda).abs() + 0.5
    w = torch.randn(b, n, n, generator=g, device=cuda)
    sc = torch.rand(b, generator=g, device=cuda) + 0.5
    gamma = torch.rand(d, generator=g, device=cuda) + 0.5
    beta = torch.randn(d, generator=g, device=cuda) * 0.1
    z = torch.nn.functional.normalize(x - x.mean(1, keepdim=True), dim=2).contiguous()
    samples = (0, 777, 2047)
    e = [ops.gram(z) for _ in range(3)]
    assert torch.equal(e[0], e[1]) and torch.equal(e[0], e[2])
    dz = [ops.gram_bwd(w, z, sc) for _ in range(3)]
    assert torch.equal(dz[0], dz[1]) and torch.equal(dz[0], dz[2])
    eu = [ops.gram(z, None, ops.KERNEL_LINEAR_UNIT) for _ in range(3)]           # the scaled-f16 kernels (unit-norm rows)
    assert torch.equal(eu[0], eu[1]) and torch.equal(eu[0], eu[2])
    du = [ops.gram_bwd(w, z, sc, unit_rows=True) for _ in range(3)]
    assert torch.equal(du[0], du[1]) and torch.equal(du[0], du[2])
    st = ops.bn_stats(x, gamma, beta)
    eb = [ops.gram_bn(x, st["a"], st["s"]) for _ in range(3)]
    assert torch.equal(eb[0][0], eb[1][0]) and torch.equal(eb[0][0], eb[2][0]) and torch.equal(eb[0][1], eb[2][1])
    db = [ops.gram_bn_bwd(w, eb[0][0], x, st["a"], st["s"], eb[0][1], st["mean"], st["rstd"], sc) for _ in range(3)]
    for k in range(3):
        assert torch.equal(db[0][k], db[1][k]) and torch.equal(db[0][k], db[2][k])
    for i in samples:
        zi = z[i].double()
        assert (e[0][i].double() - zi @ zi.T).abs().max().item() < 2e-6
        a = (w[i] + w[i].T).double() * sc[i].double()
        assert float((dz[0][i].double() - a @ zi).norm() / (a @ zi).norm()) < 1e-5
        assert (eu[0][i].double() - zi @ zi.T).abs().max().item() < 2e-6
        assert float((du[0][i].double() - a @ zi).norm() / (a @ zi).norm()) < 1e-5
        # fused front end against float64 autograd of bn_out(train) -> normalize -> <W, E>
        xi = x[i].double().detach().requires_grad_(True)
        g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        yi = torch.nn.functional.batch_norm(xi, None, None, g64, b64, True, 0.1, 1e-5)
        zn = torch.nn.functional.normalize(yi, dim=1)
        ei = zn @ zn.T
        assert (eb[0][0][i].double() - ei).abs().max().item() < 2e-6
        (sc[i].double() * (ei * w[i].double()).sum()).backward()
        assert float((db[0][0][i].double() - xi.grad).norm() / xi.grad.norm()) < GRAD_RTOL
        assert float((db[0][1][i].double() - g64.grad).norm() / g64.grad.norm()) < GRAD_RTOL
        assert float((db[0][2][i].double() - b64.grad).norm() / b64.grad.norm()) < GRAD_RTOL


# ----------------------------------------------------------------------------------------------
# the method surface end to end (DKT class on Conv4S / synthetic Omniglot-shaped images)
# ----------------------------------------------------------------------------------------------
class _Loader:
    def __init__(self, n_ep, n_way, per, hw, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = [torch.rand(n_way, per, 3, hw, hw, generator=g) for _ in range(n_ep)]

    def __len__(self):
        return len(self.x)

    def __iter__(self):
        return iter((x, None) for x in self.x)


def test_dkt_train_step_matches_float64_autograd(cuda):
    """One optimizer-free training episode through DKT._episode_loss vs the torch float64 restatement
    run on a CPU copy of the same backbone (gradients w.r.t. every backbone / bn_out / GP parameter)."""
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5).to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.0, 0.5]))
        m.model.mean_constant.copy_(torch.tensor([0.05, -0.1, 0.0, 0.02, 0.1]))
    import copy
    ref = copy.deepcopy(m).cpu().double()
    x = torch.rand(5, 21, 3, 28, 28, generator=torch.Generator().manual_seed(1))
    x_all = x.view(105, 3, 28, 28)
    m.train()
    z = m._embed(x_all.to(cuda))
    y = m._targets(5, 21, cuda)
    loss, aux = m._episode_loss(z, y)
    loss.backward()
    ref.train()
    zr = ref._embed(x_all.double())
    loss_r, logp_r, _ = T.classification_loss(zr, 5, ref.model.outputscale, ref.model.mean, ref.model.noise)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    checked = 0
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None
            continue
        # conv biases in front of a train-mode BN have an exactly-zero gradient: absolute floor
        diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
        assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
        checked += 1
    assert checked >= 18


@pytest.mark.parametrize("kernel", ["bncossim", "cossim"])
def test_dkt_train_step_fused_front_end_matches_float64_autograd(cuda, kernel):
    """The drop-in training step with bn_out + F.normalize inside the Gram kernels (DKT._episode_loss_from_trunk):
    loss, every backbone / bn_out / GP gradient and the running statistics vs the float64 torch restatement."""
    import copy
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5, kernel_type=kernel).to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.0, 0.5]))
        m.model.mean_constant.copy_(torch.tensor([0.05, -0.1, 0.0, 0.02, 0.1]))
        if kernel == "bncossim":
            m.feature_extractor.trunk.bn_out.weight.uniform_(0.5, 1.5)
            m.feature_extractor.trunk.bn_out.bias.normal_(0.0, 0.1)
    ref = copy.deepcopy(m).cpu().double()
    x = torch.rand(5, 21, 3, 28, 28, generator=torch.Generator().manual_seed(1))
    x_all = x.view(105, 3, 28, 28)
    m.train()
    x_feat = m._trunk_features(x_all.to(cuda))
    assert m._fused_front_end(x_feat.shape[0], x_feat.shape[1])
    y = m._targets(5, 21, cuda)
    loss, aux, z_train = m._episode_loss_from_trunk(x_feat, y)
    loss.backward()
    ref.train()
    zr = ref._embed(x_all.double())
    loss_r, _, _ = T.classification_loss(zr, 5, ref.model.outputscale, ref.model.mean, ref.model.noise)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    assert rel_l2(z_train.cpu().numpy(), zr.detach().numpy()) < 1e-5
    checked = 0
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, name
            continue
        diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
        assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
        checked += 1
    assert checked >= 16
    if kernel == "bncossim":
        bn, bnr = m.feature_extractor.trunk.bn_out, ref.feature_extractor.trunk.bn_out
        assert rel_l2(bn.running_mean.cpu().numpy(), bnr.running_mean.numpy()) < 1e-5
        assert rel_l2(bn.running_var.cpu().numpy(), bnr.running_var.numpy()) < 1e-4
        assert int(bn.num_batches_tracked.item()) == int(bnr.num_batches_tracked.item()) == 1


@pytest.mark.parametrize("kernel", ["bncossim", "rbf"])
def test_dkt_meta_batch_step_is_the_mean_of_the_single_episode_losses(cuda, kernel, capsys):
    """train.py --meta_batch B (opt-in; 1 = the reference's one Adam step per episode, DKT.py:160-164): one backbone pass over the
    B x N images, bn_out and the C GPs per episode through the batched [B, N, D] hot-path entries.  The step's loss and every
    gradient must equal the float64 restatement of mean_b loss(episode b), and bn_out's running estimates what nn.BatchNorm1d
    holds after seeing the B episodes one after the other."""
    import copy
    torch.manual_seed(0)
    nb, n_way, per = 3, 5, 21
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=n_way, n_support=5, kernel_type=kernel).to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.0, 0.5]))
        m.model.mean_constant.copy_(torch.tensor([0.05, -0.1, 0.0, 0.02, 0.1]))
        if m.model.raw_lengthscale is not None:
            m.model.raw_lengthscale.copy_(torch.tensor([8.0, 6.5, 9.0, 7.0, 10.0]))
    ref = copy.deepcopy(m).cpu().double()
    x = torch.rand(nb, n_way, per, 3, 28, 28, generator=torch.Generator().manual_seed(1))
    n = n_way * per
    x_all = x.view(nb * n, 3, 28, 28)
    m.train()
    y = m._targets(n_way, per, cuda)
    x_feat = m._trunk_features(x_all.to(cuda))
    if kernel == "bncossim":
        assert m._fused_front_end(n, x_feat.shape[1])
        loss, aux, _ = m._episode_loss_from_trunk(x_feat.view(nb, n, -1), y)
    else:
        loss, aux = m._episode_loss(x_feat.view(nb, n, -1), y)
    loss.backward()
    assert aux["logp"].shape == (nb, n_way) and int(aux["info"].abs().max().item()) == 0
    ref.train()
    xr = ref._trunk_features(x_all.double())                       # the backbone's BatchNorm2d layers see all nb * N images
    bnr = getattr(ref.feature_extractor.trunk, "bn_out", None) if kernel == "bncossim" else None
    losses = []
    for b in range(nb):
        zb = xr[b * n:(b + 1) * n]
        if bnr is not None:
            zb = bnr(zb)                                            # per-episode batch statistics, running estimates episode by episode
        if ref.normalize:
            zb = torch.nn.functional.normalize(zb, p=2, dim=1)
        lb, _, _ = T.classification_loss(zb, n_way, ref.model.outputscale, ref.model.mean, ref.model.noise, kernel, ref.model.lengthscale)
        losses.append(lb)
    loss_r = torch.stack(losses).mean()
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, name
            continue
        diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
        assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
    if bnr is not None:
        bn = m.feature_extractor.trunk.bn_out
        assert rel_l2(bn.running_mean.cpu().numpy(), bnr.running_mean.numpy()) < 1e-5
        assert rel_l2(bn.running_var.cpu().numpy(), bnr.running_var.numpy()) < 1e-4
        assert int(bn.num_batches_tracked.item()) == int(bnr.num_batches_tracked.item()) == nb
    # the driver-facing switch: train_loop with meta_batch = 2 takes len(loader) // 2 Adam steps
    m.meta_batch = 2
    before = m.model.raw_outputscale.detach().clone()
    m.train_loop(0, _Loader(5, n_way, per, 28, 3), None)
    out = capsys.readouterr().out
    assert "Epoch [0] [0/2]" in out and not torch.equal(before, m.model.raw_outputscale.detach())


# ----------------------------------------------------------------------------------------------
# BNCosSim front half fused into the Gram build (bn_out + F.normalize + LinearKernel, DKT.py:48,141-142,375-378)
# ----------------------------------------------------------------------------------------------
def _relu_like(rng, b, n, d):
    """Trunk outputs: non-negative, a large common offset per feature (what BatchNorm has to remove)."""
    return (np.abs(rng.standard_normal((b, n, d))) * rng.uniform(0.2, 3.0, (1, 1, d)) + rng.uniform(0.0, 5.0, (1, 1, d))).astype(np.float32)


@pytest.mark.parametrize("b,n,d", [(2, 5, 12), (3, 25, 64), (2, 85, 512), (2, 105, 1600), (1, 128, 100), (2, 19, 2916)])
def test_bn_stats_and_fused_gram_train_mode(cuda, b, n, d):
    rng = np.random.default_rng(n * 31 + d)
    x = _relu_like(rng, b, n, d)
    gamma = rng.uniform(0.5, 1.5, d).astype(np.float32)
    beta = rng.normal(0.0, 0.2, d).astype(np.float32)
    st = ops.bn_stats(dev_t(x, cuda), dev_t(gamma, cuda), dev_t(beta, cuda), 1e-5)
    e, rnorm = ops.gram_bn(dev_t(x, cuda), st["a"], st["s"])
    for i in range(b):
        y, mu, var_u = O.batchnorm1d_train(x[i].astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
        assert np.abs(st["mean"][i].cpu().numpy() - mu).max() < 1e-5 * (1.0 + np.abs(mu).max())
        assert rel_l2(st["var_unbiased"][i].cpu().numpy(), var_u) < 2e-5
        a64 = gamma / np.sqrt(x[i].astype(np.float64).var(0) + 1e-5)
        assert rel_l2(st["a"][i].cpu().numpy(), a64) < 2e-5
        zn = O.l2_normalize(y)
        ref = zn @ zn.T
        assert np.abs(e[i].cpu().numpy() - ref).max() < 2e-5, np.abs(e[i].cpu().numpy() - ref).max()
        assert rel_l2(rnorm[i].cpu().numpy(), 1.0 / np.linalg.norm(y, axis=1)) < 2e-5
    assert torch.equal(e, e.transpose(1, 2))


@pytest.mark.parametrize("affine", [True, False])
@pytest.mark.parametrize("b,n,d", [(2, 5, 12), (3, 25, 64), (2, 85, 512), (2, 105, 1600), (1, 128, 100), (2, 19, 2916), (5, 1, 36), (2, 33, 8)])
def test_one_pass_statistics_and_gram_train_mode(cuda, b, n, d, affine):
    """dkt_gram_bn_train_f32: the batch statistics taken inside the Gram kernel's staging path (X read once) against the float64 oracle of
    bn_out(train) -> F.normalize -> Gram, and against the two-launch route it replaces (dkt_bn_stats_f32 + dkt_gram_bn_f32)."""
    rng = np.random.default_rng(n * 37 + d)
    x = _relu_like(rng, b, n, d)
    gamma = rng.uniform(0.5, 1.5, d).astype(np.float32) if affine else None
    beta = rng.normal(0.0, 0.2, d).astype(np.float32) if affine else None
    gd = None if gamma is None else dev_t(gamma, cuda)
    bd = None if beta is None else dev_t(beta, cuda)
    e, rnorm, st = ops.gram_bn_train(dev_t(x, cuda), gd, bd, 1e-5)
    st2 = ops.bn_stats(dev_t(x, cuda), gd, bd, 1e-5)
    e2, rnorm2 = ops.gram_bn(dev_t(x, cuda), st2["a"], st2["s"])
    g64 = np.ones(d) if gamma is None else gamma.astype(np.float64)
    b64 = np.zeros(d) if beta is None else beta.astype(np.float64)
    for i in range(b):
        y, mu, var_u = O.batchnorm1d_train(x[i].astype(np.float64), g64, b64)
        assert np.abs(st["mean"][i].cpu().numpy() - mu).max() < 1e-5 * (1.0 + np.abs(mu).max())
        if n > 1:
            assert rel_l2(st["var_unbiased"][i].cpu().numpy(), var_u) < 2e-5
        a64 = g64 / np.sqrt(x[i].astype(np.float64).var(0) + 1e-5)
        assert rel_l2(st["a"][i].cpu().numpy(), a64) < 2e-5
        assert rel_l2(st["rstd"][i].cpu().numpy(), 1.0 / np.sqrt(x[i].astype(np.float64).var(0) + 1e-5)) < 2e-5
        assert np.abs(st["s"][i].cpu().numpy() - (b64 - mu * a64)).max() < 2e-5 * (1.0 + np.abs(mu * a64).max())
        if n > 1:                                      # a batch of one: y = beta exactly, its direction is rounding noise unless beta != 0
            zn = O.l2_normalize(y)
            ref = zn @ zn.T
            assert np.abs(e[i].cpu().numpy() - ref).max() < 2e-5, np.abs(e[i].cpu().numpy() - ref).max()
            assert rel_l2(rnorm[i].cpu().numpy(), 1.0 / np.linalg.norm(y, axis=1)) < 2e-5
    assert torch.equal(e, e.transpose(1, 2))
    if n > 1:
        assert (e - e2).abs().max().item() < 2e-5 and rel_l2(rnorm.cpu().numpy(), rnorm2.cpu().numpy()) < 2e-5
    for k in ("mean", "rstd", "a", "s", "var_unbiased"):
        assert rel_l2(st[k].cpu().numpy(), st2[k].cpu().numpy()) < 1e-5, k
    e3, _, st3 = ops.gram_bn_train(dev_t(x, cuda), gd, bd, 1e-5)          # fixed reduction order: bitwise reproducible
    assert torch.equal(e, e3) and torch.equal(st["a"], st3["a"])


@pytest.mark.parametrize("n,d", [(105, 64), (20, 64), (85, 100)])
def test_fused_train_forward_optional_outputs_at_the_c_abi(cuda, n, d):
    """dkt_gram_bn_train_f32 called at the C ABI with var_unbiased = NULL and gamma = beta = NULL (empty descriptors inside the f16 kernel): same E / rnorm / statistics
    as the full call; nothing is written through the NULL pointer."""
    L = dkt_amd._lib.load()
    rng = np.random.default_rng(n + d)
    x = dev_t(_relu_like(rng, 3, n, d), cuda)
    p = lambda t: 0 if t is None else t.data_ptr()
    full = ops.gram_bn_train(x, None, None, 1e-5)
    outs = {k: torch.full((3, d), float("nan"), device=cuda) for k in ("mean", "rstd", "a", "s")}
    e = torch.empty(3, n, n, device=cuda)
    rn = torch.empty(3, n, device=cuda)
    rc = L.dkt_gram_bn_train_f32(p(x), 0, 0, 1e-5, p(outs["mean"]), p(outs["rstd"]), p(outs["a"]), p(outs["s"]), 0, p(e), p(rn), 3, n, d, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(e, full[0]) and torch.equal(rn, full[1])
    for k in outs:
        assert torch.equal(outs[k], full[2][k]), k


@pytest.mark.parametrize("b,c,n,cmap,power", [(3, 5, 105, "rbf", 1), (2, 5, 80, "matern", 1), (4, 3, 33, "poly", 2), (1, 8, 128, "rbf", 1), (2, 2, 7, "poly", 1), (70, 5, 25, "rbf", 1)])
def test_class_kernel_backward_row_kernel_twin(cuda, monkeypatch, b, c, n, cmap, power):
    """dkt_class_kernel_bwd_f32 at N <= 128, C <= 8 (round 5: all loads of a row in flight, per-class constants and partials in registers) against the
    element-by-element kernel of round 3 (DKT_CLASS_BWD_N128=0) and float64."""
    rng = np.random.default_rng(n * 3 + c)
    base = rng.uniform(0.0, 2.0, (b, n, n))
    base = (0.5 * (base + base.transpose(0, 2, 1))).astype(np.float32)
    w = rng.standard_normal((b, c, n, n)) * 0.05
    w = (w + w.transpose(0, 1, 3, 2)).astype(np.float32)
    param = np.linspace(0.7, 1.5, c).astype(np.float32)
    km = {"rbf": ops.CLASSMAP_RBF, "matern": ops.CLASSMAP_MATERN25, "poly": ops.CLASSMAP_POLY}[cmap]
    out = {}
    for v in ("0", "1"):
        monkeypatch.setenv("DKT_CLASS_BWD_N128", v)
        out[v] = ops.class_kernel_bwd(dev_t(w, cuda), dev_t(base, cuda), km, power, dev_t(param, cuda))
    monkeypatch.delenv("DKT_CLASS_BWD_N128")
    # float64: E_c = f_c(base), obj = sum_c <W_c, E_c>;  Wp and dparam are the gradients w.r.t. the (squared-distance / Gram) base and the parameter
    bt = torch.tensor(base, dtype=torch.float64, requires_grad=True)
    pt = torch.tensor(param, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w, dtype=torch.float64)
    if cmap == "poly":
        e = (bt.unsqueeze(1) + pt.view(1, c, 1, 1)) ** power
    else:
        u = bt.unsqueeze(1) / pt.view(1, c, 1, 1) ** 2
        if cmap == "rbf":
            e = torch.exp(-0.5 * u)
        else:
            r = torch.sqrt(5.0 * u.clamp_min(1e-30))
            e = (1.0 + r + r * r / 3.0) * torch.exp(-r)
    (wt * e).sum().backward()
    for v in ("0", "1"):
        wp, dparam = out[v]
        if cmap == "poly":
            ref_wp = bt.grad.numpy()
        else:                                                     # distance kinds: Wp = diag(A 1) - A with A = 2 d obj / d d2 (what dkt_gram_bwd_f32 turns into dZ)
            a = 2.0 * bt.grad.numpy()
            ref_wp = -a
            idx = np.arange(n)
            ref_wp[:, idx, idx] = a.sum(2) - a[:, idx, idx]
        assert rel_l2(wp.cpu().numpy(), ref_wp) < 2e-6, (v, rel_l2(wp.cpu().numpy(), ref_wp))
        assert rel_l2(dparam.sum(0).cpu().numpy(), pt.grad.numpy()) < 2e-5, v
    assert rel_l2(out["1"][0].cpu().numpy(), out["0"][0].cpu().numpy()) < 1e-6


@pytest.mark.parametrize("b,n,d,kind", [(5, 19, 2916, "rbf"), (3, 25, 1600, "linear"), (6, 32, 2052, "linear"), (2, 17, 4096, "rbf"), (7, 10, 2916, "rbf"),
                                        (4, 20, 2916, "rbf"), (9, 18, 512, "linear"), (3, 19, 2916, "linear"), (5, 20, 64, "rbf")])
def test_small_gram_workgroup_per_task_twin(cuda, monkeypatch, b, n, d, kind):
    """N <= 32 with long rows (round 5): a workgroup per task (DKT_GRAM_SMALL_WG=1; the default from 2048 / 1024 features) against the wave-per-task kernels
    (=0) and float64 -- the forward sums its four partial Grams in a fixed order (equal to rounding, bitwise reproducible), the backward is bitwise equal.
    Round 6: 17 <= N <= 20 forms the rows beyond sixteen on the VALU (the default; DKT_GRAM_SMALL_XR=0 = three MFMA tiles): both forms, both launch shapes."""
    rng = np.random.default_rng(n * 7 + d)
    z = (rng.standard_normal((b, n, d)) * 0.05).astype(np.float32)
    w = (rng.standard_normal((b, n, n)) * 0.1).astype(np.float32)
    zd, wd = dev_t(z, cuda), dev_t(w, cuda)
    ls = torch.tensor([1.1], device=cuda)
    k = ops.KERNEL_RBF if kind == "rbf" else ops.KERNEL_LINEAR
    out = {}
    variants = [("0", "1"), ("1", "1"), ("0", "0"), ("1", "0")]            # (workgroup per task, extra rows on the VALU)
    for v in variants:
        monkeypatch.setenv("DKT_GRAM_SMALL_WG", v[0])
        monkeypatch.setenv("DKT_GRAM_SMALL_XR", v[1])
        out[v] = (ops.gram(zd, None, k, ls if kind == "rbf" else None), ops.gram_bwd(wd, zd))
        e2 = ops.gram(zd, None, k, ls if kind == "rbf" else None)
        assert torch.equal(e2, out[v][0])
    monkeypatch.delenv("DKT_GRAM_SMALL_WG")
    monkeypatch.delenv("DKT_GRAM_SMALL_XR")
    for v in variants[1:]:
        assert torch.equal(out[variants[0]][1], out[v][1])
    if not 16 < n <= 20:                                                   # outside 17 .. 20 rows the switch selects nothing
        assert torch.equal(out[("0", "1")][0], out[("0", "0")][0]) and torch.equal(out[("1", "1")][0], out[("1", "0")][0])
    z64 = z.astype(np.float64)
    for i in range(b):
        g = z64[i] @ z64[i].T
        if kind == "rbf":
            d2 = np.maximum(np.diag(g)[:, None] + np.diag(g)[None, :] - 2 * g, 0.0)
            g = np.exp(-0.5 * d2 / 1.1 ** 2)
        for v in variants:
            # (4e-6: the wave-per-task kernel sums all D products of an element in ONE fp32 chain -- 2.0e-6 on the diagonal of a 19 x 2916 linear Gram)
            assert np.abs(out[v][0][i].cpu().numpy() - g).max() < 4e-6 * max(1.0, np.abs(g).max()), (v, i)
        dz = (w[i].astype(np.float64) + w[i].astype(np.float64).T) @ z64[i]
        assert rel_l2(out[("1", "1")][1][i].cpu().numpy(), dz) < 2e-6
    for v in variants:
        assert torch.equal(out[v][0], out[v][0].transpose(1, 2)), v
    # the product library's default dispatch is one of the twins, bitwise
    monkeypatch.setenv("DKT_TWINS", "0")
    e_prod = ops.gram(zd, None, k, ls if kind == "rbf" else None)
    assert any(torch.equal(e_prod, out[v][0]) for v in variants)
    assert torch.equal(ops.gram_bwd(wd, zd), out[variants[0]][1])


@pytest.mark.parametrize("n,d", [(105, 1600), (85, 512), (50, 96), (128, 64), (40, 2916)])
def test_fused_train_forward_f16_split_twins_and_fixup(cuda, monkeypatch, n, d):
    """dkt_gram_bn_train_f32 at N > 32 (round 5): the scaled 2-way f16 split under train-mode BatchNorm's a-priori element bound in the pipelined kernel
    (default) against the 3-way bf16 kernel it replaces (DKT_GRAM_BN_F16=0) and float64; and the a-posteriori row check:
    an episode with a row whose norm is far below the element bound (a sample that sits on the batch mean; one feature with an enormous gamma) is
    redone by the bf16 kernel in the fix-up launch -- its outputs are then bitwise those of DKT_GRAM_BN_F16=0, its neighbours in the batch untouched."""
    rng = np.random.default_rng(n * 11 + d)
    b = 4
    x = _relu_like(rng, b, n, d)
    x[1, 3] = (x[1].astype(np.float64).sum(0) - x[1, 3]) / (n - 1)          # episode 1: row 3 on the batch mean -> |y_3| ~ rounding noise (beta = 0 below)
    gamma = rng.uniform(0.5, 1.5, d).astype(np.float32)
    xd, gd = dev_t(x, cuda), dev_t(gamma, cuda)
    out = {}
    for v in ("0", "1"):
        monkeypatch.setenv("DKT_GRAM_BN_F16", v)
        out[v] = ops.gram_bn_train(xd, gd, None, 1e-5)
    monkeypatch.delenv("DKT_GRAM_BN_F16")
    e_p, rn_p, st_p = ops.gram_bn_train(xd, gd, None, 1e-5)                  # the product library's default
    assert torch.equal(e_p, out["1"][0]) and torch.equal(rn_p, out["1"][1]) and torch.equal(st_p["a"], out["1"][2]["a"])
    e0, rn0, st0 = out["0"]
    for v in ("1",):
        e, rn, st = out[v]
        assert torch.isfinite(e).all() and torch.isfinite(rn).all()
        for k in ("mean", "rstd", "a", "s", "var_unbiased"):
            assert rel_l2(st[k].cpu().numpy(), st0[k].cpu().numpy()) < 1e-6, k      # the statistics path is the same arithmetic in all three
        assert torch.equal(e[1], e0[1]) and torch.equal(rn[1], rn0[1])      # flagged -> the bf16 kernel's result, bit for bit
        for i in (0, 2, 3):
            assert not torch.equal(e[i], e0[i])                             # (the check discriminates: an f16 result differs in the last bits)
            y, _, _ = O.batchnorm1d_train(x[i].astype(np.float64), gamma.astype(np.float64), np.zeros(d))
            zn = O.l2_normalize(y)
            err, err0 = np.abs(e[i].cpu().numpy() - zn @ zn.T).max(), np.abs(e0[i].cpu().numpy() - zn @ zn.T).max()
            assert err < 6e-6 and err < 3.0 * err0 + 2e-6, (v, i, err, err0)
            assert rel_l2(rn[i].cpu().numpy(), 1.0 / np.linalg.norm(y, axis=1)) < 2e-6
        assert torch.equal(e, e.transpose(1, 2))
        monkeypatch.setenv("DKT_GRAM_BN_F16", v)
        e_a, rn_a, _ = ops.gram_bn_train(xd[2:3].contiguous(), gd, None, 1e-5)          # an episode's result does not depend on its neighbours
        assert torch.equal(e_a[0], e[2]) and torch.equal(rn_a[0], rn[2])
        # one feature's gamma dwarfs the rest: the element bound pushes every ordinary row under the flush threshold -> all episodes redone
        g2 = gamma.copy()
        g2[0] = 3.0e6
        e_g, rn_g, _ = ops.gram_bn_train(xd, dev_t(g2, cuda), None, 1e-5)
        monkeypatch.setenv("DKT_GRAM_BN_F16", "0")
        e_g0, rn_g0, _ = ops.gram_bn_train(xd, dev_t(g2, cuda), None, 1e-5)
        assert torch.equal(e_g, e_g0) and torch.equal(rn_g, rn_g0)
    monkeypatch.delenv("DKT_GRAM_BN_F16")


@pytest.mark.parametrize("b,n,d", [(3, 25, 64), (2, 105, 1600), (2, 75, 512)])
def test_fused_gram_eval_mode_and_plain_cossim(cuda, b, n, d):
    rng = np.random.default_rng(n + d)
    x = _relu_like(rng, b, n, d)
    gamma, beta = rng.uniform(0.5, 1.5, d), rng.normal(0.0, 0.2, d)
    rm, rv = rng.uniform(0.0, 5.0, d), rng.uniform(0.1, 4.0, d)
    a = gamma / np.sqrt(rv + 1e-5)
    s = beta - rm * a
    e, _ = ops.gram_bn(dev_t(x, cuda), dev_t(a, cuda), dev_t(s, cuda))                    # [D] affine map: running statistics
    e1, _ = ops.gram_bn(dev_t(x, cuda), torch.ones(d, device=cuda), torch.zeros(d, device=cuda))    # cossim: no bn_out
    for i in range(b):
        zn = O.l2_normalize(O.batchnorm1d_eval(x[i].astype(np.float64), rm, rv, gamma, beta))
        assert np.abs(e[i].cpu().numpy() - zn @ zn.T).max() < 2e-5
        z1 = O.l2_normalize(x[i].astype(np.float64))
        assert np.abs(e1[i].cpu().numpy() - z1 @ z1.T).max() < 2e-5


@pytest.mark.parametrize("b,c,per,d", [(2, 5, 5, 64), (2, 5, 21, 1600), (3, 5, 17, 512), (2, 3, 6, 40), (1, 2, 64, 128), (2, 5, 30, 64), (2, 20, 21, 128), (1, 3, 67, 36),
                                       (2, 5, 21, 64), (1, 20, 21, 64)])       # the last four with D <= 64 < N: the feature-space path behind dkt_affine_normalize_f32 (round 5)
def test_episode_loss_from_trunk_features_matches_float64_autograd(cuda, b, c, per, d):
    """bn_out(train) + F.normalize + Gram + MLL and the whole backward (dX, dgamma, dbeta, hyper-parameters) in the fused
    kernels vs float64 torch autograd of the reference formulation.  N <= 128: the episode-resident kernels (the normalised features never leave the chip);
    N = 150, 201, 420 (the 20-way shape): dkt_bn_stats_f32 -> dkt_affine_normalize_f32 -> the large-N Gram / marginal-likelihood kernels and back through
    dkt_gram_bwd_f32 -> dkt_normalize_bn_bwd_f32."""
    n = c * per
    rng = np.random.default_rng(b * 100 + n + d)
    x = _relu_like(rng, b, n, d)
    gamma = rng.uniform(0.5, 1.5, d).astype(np.float32)
    beta = rng.normal(0.0, 0.2, d).astype(np.float32)
    raw_s = rng.normal(0.0, 0.5, c).astype(np.float32)
    mean = rng.normal(0.0, 0.1, c).astype(np.float32)
    xt = dev_t(x, cuda).requires_grad_(True)
    gt, bt = dev_t(gamma, cuda).requires_grad_(True), dev_t(beta, cuda).requires_grad_(True)
    rst, mt = dev_t(raw_s, cuda).requires_grad_(True), dev_t(mean, cuda).requires_grad_(True)
    noise = torch.full((c,), 0.1, device=cuda)
    cls = torch.arange(c, device=cuda).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=cuda).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    obj, logp, alpha, info, jit, e, bmean, bvar = ops.episode_loss_bn(xt, gt, bt, y, torch.nn.functional.softplus(rst), mt, noise, cw)
    assert int(info.abs().max().item()) == 0
    assert (e is None) == ops.lowrank_applies(n, d, c, b, front_end=True)      # D <= 64 and more than 128 rows: the episode ran in feature space, no N x N matrix exists
    w_ep = torch.linspace(0.5, 1.5, b, device=cuda)                # non-uniform upstream gradient per episode
    (obj * w_ep).sum().backward()
    # float64 reference
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    g64, b64 = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
    rs64, m64 = torch.tensor(raw_s, dtype=torch.float64, requires_grad=True), torch.tensor(mean, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for i in range(b):
        zi = torch.nn.functional.batch_norm(x64[i], None, None, g64, b64, True, 0.1, 1e-5)
        loss_i, logp_i, _ = T.classification_loss(zi, c, torch.nn.functional.softplus(rs64), m64, torch.full((c,), 0.1, dtype=torch.float64),
                                                  normalize=True)
        assert abs(obj[i].item() - loss_i.item()) < MLL_RTOL * abs(loss_i.item())
        total = total + float(w_ep[i].item()) * loss_i
    total.backward()
    assert rel_l2(xt.grad.cpu().numpy(), x64.grad.numpy()) < GRAD_RTOL
    assert rel_l2(gt.grad.cpu().numpy(), g64.grad.numpy()) < GRAD_RTOL
    assert rel_l2(bt.grad.cpu().numpy(), b64.grad.numpy()) < GRAD_RTOL
    assert rel_l2(rst.grad.cpu().numpy(), rs64.grad.numpy()) < GRAD_RTOL
    assert rel_l2(mt.grad.cpu().numpy(), m64.grad.numpy()) < GRAD_RTOL
    # the batch statistics returned for the running-estimate update
    assert rel_l2(bmean.cpu().numpy(), x.astype(np.float64).mean(1)) < 1e-5
    assert rel_l2(bvar.cpu().numpy(), x.astype(np.float64).var(1, ddof=1)) < 5e-5
    # the pieces of the large-N front end on their own: Zn, rnorm and the backward with constant statistics (eval mode / no bn_out) vs float64
    if n > 128:
        a_ev = torch.linspace(0.5, 1.5, d, device=cuda)
        s_ev = torch.linspace(-0.2, 0.3, d, device=cuda)
        zn, rn = ops.affine_normalize(xt.detach(), a_ev, s_ev)
        y64 = torch.tensor(x, dtype=torch.float64) * a_ev.double().cpu() + s_ev.double().cpu()
        assert rel_l2(zn.cpu().numpy(), torch.nn.functional.normalize(y64, p=2, dim=2).numpy()) < 2e-6
        assert rel_l2(rn.cpu().numpy(), (1.0 / y64.norm(dim=2)).numpy()) < 2e-6
        gz = torch.randn(b, n, d, generator=torch.Generator().manual_seed(5))
        xe = torch.tensor(x, dtype=torch.float64, requires_grad=True)
        (torch.nn.functional.normalize(xe * a_ev.double().cpu() + s_ev.double().cpu(), p=2, dim=2) * gz.double()).sum().backward()
        dx_ev, dg_ev, db_ev = ops.normalize_bn_bwd(gz.to(cuda), zn, xt.detach(), a_ev, rn)
        assert dg_ev is None and db_ev is None and rel_l2(dx_ev.cpu().numpy(), xe.grad.numpy()) < 5e-6


@pytest.mark.parametrize("b,c,per,d", [(2, 5, 30, 64), (1, 20, 21, 36), (2, 5, 12, 64)])
def test_episode_loss_plain_cossim_from_features_matches_float64_autograd(cuda, b, c, per, d):
    """The cossim kernel (no bn_out: F.normalize + Gram + MLL) through the same entry, small and large episodes: obj and d obj / d x vs float64 autograd."""
    n = c * per
    rng = np.random.default_rng(n + d)
    x = _relu_like(rng, b, n, d)
    xt = dev_t(x, cuda).requires_grad_(True)
    raw_s = rng.normal(0.0, 0.5, c).astype(np.float32)
    mean = rng.normal(0.0, 0.1, c).astype(np.float32)
    rst, mt = dev_t(raw_s, cuda).requires_grad_(True), dev_t(mean, cuda).requires_grad_(True)
    noise = torch.full((c,), 0.1, device=cuda)
    cls = torch.arange(c, device=cuda).repeat_interleave(per)
    y = torch.where(cls.unsqueeze(0) == torch.arange(c, device=cuda).unsqueeze(1), 1.0, -1.0).contiguous()
    cw = torch.full((c,), -1.0 / (c * n), device=cuda)
    out = ops.episode_loss_bn(xt, None, None, y, torch.nn.functional.softplus(rst), mt, noise, cw, use_bn=False)
    obj, info = out[0], out[3]
    assert int(info.abs().max().item()) == 0
    obj.sum().backward()
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    rs64, m64 = torch.tensor(raw_s, dtype=torch.float64, requires_grad=True), torch.tensor(mean, dtype=torch.float64, requires_grad=True)
    total = 0.0
    for i in range(b):
        loss_i, _, _ = T.classification_loss(x64[i], c, torch.nn.functional.softplus(rs64), m64, torch.full((c,), 0.1, dtype=torch.float64), normalize=True)
        assert abs(obj[i].item() - loss_i.item()) < MLL_RTOL * abs(loss_i.item())
        total = total + loss_i
    total.backward()
    assert rel_l2(xt.grad.cpu().numpy(), x64.grad.numpy()) < GRAD_RTOL
    assert rel_l2(rst.grad.cpu().numpy(), rs64.grad.numpy()) < GRAD_RTOL and rel_l2(mt.grad.cpu().numpy(), m64.grad.numpy()) < GRAD_RTOL


@pytest.mark.parametrize("kernel", ["rbf", "matern", "poli1", "poli2", "linear", "cossim"])
def test_dkt_every_kernel_type_matches_float64_autograd(cuda, kernel):
    """configs.kernel_type values of the reference's ExactGPLayer (DKT.py:352-370): loss, gradients w.r.t. every
    parameter (backbone, outputscale, mean, lengthscale / offset / variance) and predictions vs the float64 restatement."""
    _every_kernel_type_check(cuda, kernel, per=12, n_support=5)


@pytest.mark.parametrize("kernel", ["rbf", "matern", "poli2"])
def test_dkt_per_class_kernels_large_episode_one_launch(cuda, kernel):
    """The same check on an episode of 150 rows (training) / a support set of 135 rows (prediction): the class models' own base matrices go
    through the tile-array pipeline of dkt_mll_f32 in ONE call (DKT_MLL_E_PER_CLASS beyond N = 111) -- no per-class loop on the host."""
    calls = []
    orig = ops.mll

    def spy(e, *a, **k):
        calls.append(tuple(e.shape))
        return orig(e, *a, **k)

    ops.mll = spy
    try:
        _every_kernel_type_check(cuda, kernel, per=30, n_support=27)
    finally:
        ops.mll = orig
    assert calls == [(1, 5, 150, 150), (1, 5, 135, 135)], calls


def _every_kernel_type_check(cuda, kernel, per, n_support):
    import copy
    torch.manual_seed(1)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=n_support, kernel_type=kernel).to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.0, 0.5]))
        m.model.mean_constant.copy_(torch.tensor([0.05, -0.1, 0.0, 0.02, 0.1]))
        # every class model owns its base-kernel parameters (one ExactGPLayer per class, DKT.py:63-66): distinct values per class
        if m.model.raw_lengthscale is not None:
            m.model.raw_lengthscale.copy_(torch.tensor([8.0, 6.5, 9.0, 7.0, 10.0]))      # un-normalised Conv4S features: distances ~ 8
        if m.model.raw_offset is not None:
            m.model.raw_offset.copy_(torch.tensor([0.3, -0.2, 0.6, 0.0, 1.0]))
        if kernel == "linear":
            m.model.raw_variance.copy_(torch.tensor([-0.4, 0.1, -0.8, 0.3, 0.0]))
    assert all(getattr(m.model, nm) is None or getattr(m.model, nm).shape == (5,) for nm in ("raw_lengthscale", "raw_offset", "raw_variance"))
    ref = copy.deepcopy(m).cpu().double()
    x = torch.rand(5, per, 3, 28, 28, generator=torch.Generator().manual_seed(2))
    x_all = x.view(5 * per, 3, 28, 28)
    m.train()
    z = m._embed(x_all.to(cuda))
    loss, aux = m._episode_loss(z, m._targets(5, per, cuda))
    loss.backward()
    assert int(aux["info"].abs().max().item()) == 0
    ref.train()
    zr = ref._embed(x_all.double())
    extra = ref.model.lengthscale if ref.model.raw_lengthscale is not None else ref.model.offset
    var = ref.model.variance if ref.model.raw_variance is not None else 1.0
    loss_r, _, alpha_r = T.classification_loss(zr, 5, ref.model.outputscale, ref.model.mean, ref.model.noise,
                                               kernel if kernel != "cossim" else "linear", extra, variance=var)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, name
            continue
        diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
        assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
    # prediction through the same kernel (cross matrix, per-class parameters) vs the float64 restatement
    m.eval()
    ref.eval()
    m.n_query = per - n_support
    logits = m.get_logits(x)
    assert logits.shape == (5 * (per - n_support), 5) and torch.isfinite(logits).all()
    with torch.no_grad():
        zs = ref._embed(x[:, :n_support].reshape(5 * n_support, 3, 28, 28).double())
        zq = ref._embed(x[:, n_support:].reshape(5 * (per - n_support), 3, 28, 28).double())
        kn = kernel if kernel != "cossim" else "linear"
        _, _, alpha_s = T.classification_loss(zs, 5, ref.model.outputscale, ref.model.mean, ref.model.noise, kn, extra, variance=var)
        mu_r = T.predict_mean(zs, zq, alpha_s, ref.model.outputscale, ref.model.mean, kn, extra, variance=var)
    assert np.abs(logits.t().cpu().numpy() - mu_r.numpy()).max() < 2e-3 * max(1.0, np.abs(mu_r.numpy()).max())


@pytest.mark.parametrize("kernel", ["rbf", "poli2"])
def test_dkt_per_class_kernels_more_than_32_classes(cuda, kernel):
    """40 class models with their own lengthscale / offset (methods/DKT.py:63-66, 352-365 at n_way = 40): dkt_class_kernel_bwd_f32 takes 32 class maps per launch,
    so the host runs the one-launch path per GROUP of 32 classes (two marginal-likelihood calls for training, two for the prediction) instead of 40 single-model
    calls.  Loss, every parameter gradient and the posterior means against the float64 restatement."""
    import copy
    n_way, per, n_support = 40, 3, 2
    calls = []
    orig = ops.mll

    def spy(e, *a, **k):
        calls.append(tuple(e.shape))
        return orig(e, *a, **k)

    torch.manual_seed(4)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=n_way, n_support=n_support, kernel_type=kernel).to(cuda)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.rand(n_way, generator=g) - 0.4)
        m.model.mean_constant.copy_(0.2 * torch.rand(n_way, generator=g) - 0.1)
        if m.model.raw_lengthscale is not None:
            m.model.raw_lengthscale.copy_(6.5 + 3.5 * torch.rand(n_way, generator=g))
        if m.model.raw_offset is not None:
            m.model.raw_offset.copy_(1.2 * torch.rand(n_way, generator=g) - 0.2)
    ref = copy.deepcopy(m).cpu().double()
    x = torch.rand(n_way, per, 3, 28, 28, generator=torch.Generator().manual_seed(6))
    x_all = x.view(n_way * per, 3, 28, 28)
    ops.mll = spy
    try:
        m.train()
        loss, aux = m._episode_loss(m._embed(x_all.to(cuda)), m._targets(n_way, per, cuda))
        loss.backward()
        m.eval()
        m.n_query = per - n_support
        logits = m.get_logits(x)
    finally:
        ops.mll = orig
    nt, ns = n_way * per, n_way * n_support
    assert calls == [(1, 32, nt, nt), (1, 8, nt, nt), (1, 32, ns, ns), (1, 8, ns, ns)], calls
    assert int(aux["info"].abs().max().item()) == 0 and aux["logp"].shape == (1, n_way) and aux["alpha"].shape == (1, n_way, nt) and aux["e"].shape == (1, n_way, nt, nt)
    ref.train()
    extra = ref.model.lengthscale if ref.model.raw_lengthscale is not None else ref.model.offset
    loss_r, _, _ = T.classification_loss(ref._embed(x_all.double()), n_way, ref.model.outputscale, ref.model.mean, ref.model.noise, kernel, extra)
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is None:
            assert p.grad is None, name
            continue
        diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
        assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
    ref.eval()
    with torch.no_grad():
        zs = ref._embed(x[:, :n_support].reshape(ns, 3, 28, 28).double())
        zq = ref._embed(x[:, n_support:].reshape(n_way * (per - n_support), 3, 28, 28).double())
        _, _, alpha_s = T.classification_loss(zs, n_way, ref.model.outputscale, ref.model.mean, ref.model.noise, kernel, extra)
        mu_r = T.predict_mean(zs, zq, alpha_s, ref.model.outputscale, ref.model.mean, kernel, extra)
    assert logits.shape == (n_way * (per - n_support), n_way)
    assert np.abs(logits.t().cpu().numpy() - mu_r.numpy()).max() < 2e-3 * max(1.0, np.abs(mu_r.numpy()).max())


@pytest.mark.parametrize("kernel", ["bncossim", "rbf"])
def test_correct_with_test_time_adaptation_matches_float64_adam(cuda, kernel):
    """`correct(x, N > 0)` (methods/DKT.py:242-272): N Adam steps (lr 1e-3) on the GP hyper-parameters only, on the support set,
    then the posterior-mean labels of the queries.  Against a float64 restatement run on a CPU copy: same average loss, the same
    adapted hyper-parameters to Adam's step size, identical query labels and correct-count."""
    import copy
    torch.manual_seed(3)
    n_way, n_support, n_query, n_steps = 5, 5, 15, 3
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=n_way, n_support=n_support, kernel_type=kernel).to(cuda)
    with torch.no_grad():
        m.model.raw_outputscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.0, 0.5]))
        m.model.mean_constant.copy_(torch.tensor([0.05, -0.1, 0.0, 0.02, 0.1]))
        if m.model.raw_lengthscale is not None:
            m.model.raw_lengthscale.copy_(torch.tensor([8.0, 6.5, 9.0, 7.0, 10.0]))
    m.train()                                             # a few train-mode passes so that the BatchNorm running estimates are not the init values
    with torch.no_grad():
        for k in range(3):
            m.feature_extractor(torch.rand(50, 3, 28, 28, generator=torch.Generator().manual_seed(10 + k)).to(cuda))
    ref = copy.deepcopy(m).cpu().double()
    x = _Loader(1, n_way, n_support + n_query, 28, 5).x[0]
    m.eval()
    m.n_query = n_query
    top1, count, avg_loss = m.correct(x, N=n_steps)
    # float64 restatement
    ref.eval()
    with torch.no_grad():
        zs = ref._embed(x[:, :n_support].reshape(n_way * n_support, 3, 28, 28).double())
        zq = ref._embed(x[:, n_support:].reshape(n_way * n_query, 3, 28, 28).double())
    kn = kernel if kernel != "cossim" else "linear"
    opt = torch.optim.Adam([{'params': ref.model.parameters()}], lr=1e-3)
    tot = 0.0
    for _ in range(n_steps):
        opt.zero_grad()
        loss_r, _, _ = T.classification_loss(zs, n_way, ref.model.outputscale, ref.model.mean, ref.model.noise, kn, ref.model.lengthscale)
        loss_r.backward()
        opt.step()
        tot += loss_r.item()
    with torch.no_grad():
        _, _, alpha_r = T.classification_loss(zs, n_way, ref.model.outputscale, ref.model.mean, ref.model.noise, kn, ref.model.lengthscale)
        mu_r = T.predict_mean(zs, zq, alpha_r, ref.model.outputscale, ref.model.mean, kn, ref.model.lengthscale)
    labels_r = torch.sigmoid(mu_r).argmax(0).numpy()
    y_q = np.repeat(np.arange(n_way), n_query)
    assert count == n_way * n_query
    assert abs(avg_loss - tot / n_steps) < 1e-4 * abs(tot / n_steps)
    assert top1 == float((labels_r == y_q).sum())
    for name in ("raw_outputscale", "mean_constant", "raw_lengthscale"):
        p, pr = getattr(m.model, name), getattr(ref.model, name)
        if p is not None:
            assert np.abs(p.detach().cpu().numpy() - pr.detach().numpy()).max() < 2e-4, name      # 3 steps of 1e-3 each
    m.n_query = n_query
    logits = m.get_logits(x)                               # with the adapted hyper-parameters
    assert (logits.argmax(1).cpu().numpy() == labels_r).all()


def test_dkt_train_loop_and_test_loop_run(cuda, capsys):
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5).to(cuda)
    before = m.model.raw_outputscale.detach().clone()
    m.train()
    m.train_loop(0, _Loader(3, 5, 21, 28, 0), None)
    assert "Epoch [0] [0/3]" in capsys.readouterr().out
    assert not torch.equal(before, m.model.raw_outputscale.detach())
    assert torch.isfinite(m._last["loss"]) and 0.0 <= m._last["acc_query"].item() <= 100.0
    m.eval()
    acc, std = m.test_loop(_Loader(4, 5, 20, 28, 1), return_std=True)
    assert 0.0 <= acc <= 100.0 and std >= 0.0
    # correct() / get_logits() against the oracle on the features the model itself produced
    x = _Loader(1, 5, 20, 28, 2).x[0]
    m.n_query = 15
    top1, count, avg_loss = m.correct(x)
    logits = m.get_logits(x)
    assert count == 75 and logits.shape == (75, 5) and avg_loss == 0.0
    with torch.no_grad():
        xs, xq = m._split(x)
        zs, zq = m._embed(xs).cpu().numpy().astype(np.float64), m._embed(xq).cpu().numpy().astype(np.float64)
    hyp = O.GPHypers(m.model.outputscale.detach().cpu().numpy().astype(np.float64),
                     m.model.mean.detach().cpu().numpy().astype(np.float64), np.full(5, 0.1))
    ref = O.eval_episode(zs, zq, 5, hyp)
    assert np.abs(logits.cpu().numpy() - ref["logits"]).max() < 1e-3
    margin = np.sort(ref["mu"], axis=0)
    decided = (margin[-1] - margin[-2]) > 1e-3        # ignore numerically tied queries
    assert (logits.argmax(1).cpu().numpy()[decided] == ref["labels"][decided]).all()
    # test-time adaptation path (N > 0) runs and returns a finite loss
    top1b, countb, avg = m.correct(x, N=2)
    assert countb == 75 and np.isfinite(avg) and avg != 0.0


@pytest.mark.parametrize("way", [5, 20])
def test_dkt_train_loop_captured_in_a_hip_graph_matches_eager(cuda, capsys, monkeypatch, way):
    """DKT_TRAIN_GRAPH=1: the whole training step (backbone, GP kernels, capturable Adam) as one hipGraph launch -- same losses as the eager loop after four steps,
    at the 5-way shape (N = 105: fused front end + N x N kernels) and at the 20-way shape (N = 420, Conv4S features: streaming front end + feature-space episode)."""
    losses = {}
    for graph in ("0", "1"):
        monkeypatch.setenv("DKT_TRAIN_GRAPH", graph)
        torch.manual_seed(0)
        m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=way, n_support=5).to(cuda)
        m.train()
        g = torch.Generator().manual_seed(1)
        loader = [(torch.rand(way, 21, 3, 28, 28, generator=g), None) for _ in range(4)]
        m.train_loop(0, loader, None, print_freq=2)
        losses[graph] = float(m._last["loss"])
        assert np.isfinite(losses[graph])
    capsys.readouterr()
    assert abs(losses["1"] - losses["0"]) < 1e-4 * abs(losses["0"])


def test_correct_laplace_branch(cuda):
    """`correct(x, laplace=True)` (reference DKT.py:207-222: sklearn's Laplace-approximation GP classifier on the embedded support set, "not the method used in the
    paper"): the surface and the return convention (top1_correct, count, 0.0)."""
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5).to(cuda)
    m.eval()
    m.n_query = 16
    x = torch.rand(5, 21, 3, 28, 28, generator=torch.Generator().manual_seed(2))
    top1, count, avg_loss = m.correct(x, laplace=True)
    assert count == 80 and 0.0 <= top1 <= 80.0 and avg_loss == 0.0 and isinstance(top1, float)


def test_dkt_failed_step_is_skipped_on_the_device_and_raised_at_the_next_print(cuda, capsys):
    """An episode whose factorisation fails (here: NaN pixels -> NaN Gram -> info != 0 after every jitter retry) must not reach the
    weights: the fused Adam step takes the failure flag as `found_inf` and leaves parameters and moments untouched; the error
    surfaces at the next print point (GPyTorch raises NotPSDError before the step, reference methods/DKT.py:161-164)."""
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5, kernel_type="cossim").to(cuda)
    m.train()
    ld = _Loader(3, 5, 21, 28, 0)
    ld.x[1] = torch.full_like(ld.x[1], float("nan"))
    with pytest.raises(RuntimeError, match="not positive definite"):
        m.train_loop(0, ld, None, print_freq=2)            # prints at i = 0 and i = 2; the NaN episode is i = 1
    capsys.readouterr()
    for name, p in list(m.model.named_parameters()) + list(m.feature_extractor.named_parameters()):
        assert torch.isfinite(p).all(), name


@pytest.mark.parametrize("kernel", ["bncossim", "cossim", "bncossim-large"])
def test_test_time_fused_front_end_matches_unfused(cuda, monkeypatch, kernel):
    """correct() / get_logits() with bn_out (running statistics) + F.normalize folded into one Gram launch over the stacked
    [support; query] trunk features, against torch's BatchNorm1d / F.normalize in front of the same GP kernels.  `-large`: 150 rows per test episode --
    one normalisation kernel (dkt_affine_normalize_f32) in front of the large-N Gram kernel."""
    torch.manual_seed(1)
    large = kernel.endswith("-large")
    kernel = kernel.split("-")[0]
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5, kernel_type=kernel).to(cuda)
    m.train()
    m.train_loop(0, _Loader(3, 5, 21, 28, 0), None)          # non-trivial running statistics and hyper-parameters
    m.eval()
    x = _Loader(1, 5, 30 if large else 20, 28, 5).x[0]
    m.n_query = 25 if large else 15
    used = []
    orig = ops.affine_normalize if large else ops.gram_bn
    monkeypatch.setattr(ops, "affine_normalize" if large else "gram_bn", lambda *a, **k: (used.append(1), orig(*a, **k))[1])
    logits_f = m.get_logits(x)
    top_f = m.correct(x)
    assert len(used) == 2, "the fused eval path must have run"
    monkeypatch.setenv("DKT_FUSED_FRONTEND", "0")
    logits_u = m.get_logits(x)
    top_u = m.correct(x)
    assert len(used) == 2
    assert (logits_f - logits_u).abs().max().item() < 5e-5 * max(1.0, logits_u.abs().max().item())
    assert top_f == top_u
    # in train mode (batch statistics) the fused eval path must not be taken
    monkeypatch.delenv("DKT_FUSED_FRONTEND")
    m.train()
    m.get_logits(x)
    assert len(used) == 2


@pytest.mark.parametrize("path", ["feature_space", "nxn"])
def test_dkt_20way_train_and_test_use_the_blocked_large_n_path(cuda, capsys, path, monkeypatch):
    """The 20-way shape through the drop-in class: 20-way 5-shot, 16 queries -> N = 420 in train_loop, 100 support / 300 query at test time; the loss of one episode
    against the float64 restatement.  Conv4S features (D = 64: the Omniglot trunk) take the feature-space episode since round 5 (`feature_space`: dkt_lowrank_*
    behind the streaming front end of dkt_frontend_big.hip); `nxn` (DKT_LOWRANK=0) keeps the N x N kernels of the cfg4 shape on the same data: tile-array MLL path."""
    if path == "nxn":
        monkeypatch.setenv("DKT_LOWRANK", "0")
    torch.manual_seed(0)
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=20, n_support=5).to(cuda)
    m.train()
    m.train_loop(0, _Loader(2, 20, 21, 28, 0), None)
    assert "Epoch [0] [0/2]" in capsys.readouterr().out
    assert torch.isfinite(m._last["loss"]) and 0.0 <= m._last["acc_query"].item() <= 100.0
    x = _Loader(1, 20, 21, 28, 3).x[0]
    z = m._embed(x.view(420, 3, 28, 28).to(cuda))
    y = m._targets(20, 21, cuda)
    loss, aux = m._episode_loss(z, y)
    assert (aux["e"] is None) == (path == "feature_space")
    hyp = O.GPHypers(m.model.outputscale.detach().cpu().numpy().astype(np.float64),
                     m.model.mean.detach().cpu().numpy().astype(np.float64), np.full(20, 0.1))
    ref = O.train_episode(z.detach().cpu().numpy().astype(np.float64), 20, hyp)
    assert abs(loss.item() - ref["loss"]) < MLL_RTOL * abs(ref["loss"])
    assert int(aux["info"].abs().max()) == 0
    m.eval()
    m.n_query = 15
    top1, count, _ = m.correct(_Loader(1, 20, 20, 28, 4).x[0])
    assert count == 300 and 0 <= top1 <= 300


def test_dkt_regression_surface(cuda):
    torch.manual_seed(0)
    bb = dkt_amd.backbone.Conv3()
    m = dkt_amd.DKTRegression(bb, "rbf").to(cuda)
    opt = torch.optim.Adam([{'params': m.model.parameters(), 'lr': 1e-3}, {'params': m.feature_extractor.parameters(), 'lr': 1e-3}])
    g = torch.Generator().manual_seed(0)
    batch = torch.rand(3, 19, 3, 100, 100, generator=g)
    labels = torch.rand(3, 19, generator=g) * 2 - 1
    # one task against the float64 restatement
    z = m.feature_extractor(batch[0].to(cuda))
    loss, aux = m._loss(z, labels[0].to(cuda))
    loss.backward()
    import copy
    ref = copy.deepcopy(m).cpu().double()
    for p in ref.parameters():
        p.grad = None
    zr = ref.feature_extractor(batch[0].double())
    loss_r, _, _ = T.regression_loss(zr, labels[0].double(), ref.model.outputscale[0], ref.model.mean[0], ref.model.noise[0],
                                     ref.model.lengthscale[0])
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    for (name, p), (_, pr) in zip(m.named_parameters(), ref.named_parameters()):
        if pr.grad is not None:
            diff = np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - pr.grad.numpy())
            assert diff <= 5e-3 * np.linalg.norm(pr.grad.numpy()) + 2e-5, (name, diff)
    m.train_loop(0, opt, batch, labels)
    mse = m.test_loop(5, inputs=batch, targets=labels)
    assert mse.dim() == 0 and torch.isfinite(mse)


# ----------------------------------------------------------------------------------------------
# spectral-mixture kernel of the regression head (DKT_regression.py:121-122)
# ----------------------------------------------------------------------------------------------
def _smk_case(rng, b, m, n, d, q, spread):
    a = (rng.standard_normal((b, m, d)) * spread).astype(np.float32)
    c = (rng.standard_normal((b, n, d)) * spread).astype(np.float32)
    w = (rng.random(q) + 0.2).astype(np.float32)
    mu = (rng.random((q, d)) * 0.8 + 0.05).astype(np.float32)
    sg = (rng.random((q, d)) * 0.8 + 0.05).astype(np.float32)
    return a, c, w, mu, sg


@pytest.mark.parametrize("b,m,n,d,q,spread", [(3, 19, 19, 2916, 4, 0.01), (2, 19, 5, 2916, 4, 0.01), (2, 7, 7, 70, 3, 0.2),
                                               (1, 33, 9, 257, 1, 0.1), (2, 12, 12, 64, 8, 0.3), (1, 19, 19, 2916, 4, 1.0)])
def test_spectral_mixture_forward_matches_oracle(cuda, b, m, n, d, q, spread):
    """spread = 1.0 is the regime of raw backbone features: every off-diagonal product of 2916 cosines leaves the fp32
    range, the matrix is sum(w) on the diagonal and (numerically) zero elsewhere -- in the oracle and on the GPU."""
    rng = np.random.default_rng(b * 100 + d + q)
    a, c, w, mu, sg = _smk_case(rng, b, m, n, d, q, spread)
    sym = m == n
    e, eq = ops.smk(dev_t(a, cuda), None if sym else dev_t(c, cuda), dev_t(w, cuda), dev_t(mu, cuda), dev_t(sg, cuda),
                    want_terms=True)
    e, eq = e.cpu().numpy(), eq.cpu().numpy()
    for i in range(b):
        ref, refq = O.gram_spectral_mixture(a[i], None if sym else c[i], w, mu, sg, terms=True)
        assert np.abs(e[i] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-30, np.abs(e[i] - ref).max()
        assert np.abs(eq[i] - refq).max() <= 2e-5
        if sym:
            assert (e[i] == e[i].T).all()
            np.testing.assert_allclose(np.diag(e[i]), np.full(m, w.astype(np.float64).sum()), rtol=1e-6)


@pytest.mark.parametrize("b,n,d,q,spread", [(2, 9, 70, 3, 0.2), (2, 19, 2916, 4, 0.01), (1, 30, 129, 2, 0.1)])
def test_spectral_mixture_backward_matches_float64_autograd(cuda, b, n, d, q, spread):
    rng = np.random.default_rng(n * 10 + q)
    a, _, w, mu, sg = _smk_case(rng, b, n, n, d, q, spread)
    ge = rng.standard_normal((b, n, n)).astype(np.float32)          # not symmetric on purpose
    leaves = [dev_t(v, cuda).requires_grad_(True) for v in (a, w, mu, sg)]
    e = ops.spectral_mixture_matrix(*leaves)
    (e * dev_t(ge, cuda)).sum().backward()
    ref = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (a, w, mu, sg)]
    tot = 0.0
    for i in range(b):
        tot = tot + (T.spectral_mixture(ref[0][i], None, ref[1], ref[2], ref[3]) * torch.tensor(ge[i], dtype=torch.float64)).sum()
    tot.backward()
    for name, g, r in zip(("dz", "dweights", "dmeans", "dscales"), leaves, ref):
        assert rel_l2(g.grad.cpu().numpy(), r.grad.numpy()) <= GRAD_RTOL, (name, rel_l2(g.grad.cpu().numpy(), r.grad.numpy()))


def test_dkt_regression_spectral_kernel(cuda):
    """DKT(backbone, 'spectral'): the loss and every gradient (mixture weights / means / scales, noise, mean, features)
    against float64 autograd of the restated formulation, then the train / test loops run."""
    torch.manual_seed(0)
    d = 96
    m = dkt_amd.DKTRegression(torch.nn.Identity(), "spectral", ard_num_dims=d).to(cuda)
    assert sorted(n for n, _ in m.model.named_parameters()) == ["mean_constant", "raw_mixture_means", "raw_mixture_scales",
                                                                "raw_mixture_weights", "raw_noise"]
    assert tuple(m.model.raw_mixture_means.shape) == (4, 1, d) and tuple(m.model.raw_mixture_weights.shape) == (4,)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        m.model.raw_mixture_means.copy_(torch.randn(4, 1, d, generator=g) * 0.5 - 1.0)
        m.model.raw_mixture_scales.copy_(torch.randn(4, 1, d, generator=g) * 0.5 - 1.0)
        m.model.raw_mixture_weights.copy_(torch.randn(4, generator=g) * 0.3)
        m.model.mean_constant.fill_(0.1)
    z = (torch.randn(19, d, generator=g) * 0.15).to(cuda).requires_grad_(True)
    labels = torch.rand(19, generator=g) * 2 - 1
    loss, aux = m._loss(z, labels.to(cuda))
    loss.backward()
    import copy
    ref = copy.deepcopy(m.model).cpu().double()
    for p in ref.parameters():
        p.grad = None
    zr = z.detach().cpu().double().requires_grad_(True)
    e = T.spectral_mixture(zr, None, ref.mixture_weights, ref.mixture_means, ref.mixture_scales)
    lp, _ = T.gp_logp(e, labels.double(), torch.ones((), dtype=torch.float64), ref.mean[0], ref.noise[0])
    loss_r = -lp / 19
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) < MLL_RTOL * abs(loss_r.item())
    assert rel_l2(z.grad.cpu().numpy(), zr.grad.numpy()) <= GRAD_RTOL
    for (name, p), (_, pr) in zip(m.model.named_parameters(), ref.named_parameters()):
        assert rel_l2(p.grad.cpu().numpy(), pr.grad.numpy()) <= GRAD_RTOL, name
    # prediction against the oracle
    hyp = O.GPHypers(np.ones(1), ref.mean.detach().numpy(), ref.noise.detach().numpy(),
                     mixture=(ref.mixture_weights.detach().numpy(), ref.mixture_means.detach().numpy().reshape(4, d),
                              ref.mixture_scales.detach().numpy().reshape(4, d)))
    zn, yn = z.detach().cpu().numpy().astype(np.float64), labels.numpy().astype(np.float64)
    sup = [0, 3, 7, 11, 18]
    mu, var = m.predict(z.detach()[sup], labels.to(cuda)[sup], z.detach(), with_variance=True)
    pr = O.regression_predict(zn[sup], yn[sup], zn, hyp, kernel="spectral")
    np.testing.assert_allclose(mu.cpu().numpy(), pr["mean"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(var.cpu().numpy(), pr["var"], rtol=2e-4, atol=2e-5)
    # full surface with the real backbone (raw Conv3 features: the matrix degenerates to sum(w) I, as in the reference)
    bb = dkt_amd.backbone.Conv3()
    m2 = dkt_amd.DKTRegression(bb, "spectral").to(cuda)
    opt = torch.optim.Adam([{'params': m2.model.parameters(), 'lr': 1e-3}, {'params': m2.feature_extractor.parameters(), 'lr': 1e-3}])
    batch = torch.rand(2, 19, 3, 100, 100, generator=g)
    lab = torch.rand(2, 19, generator=g) * 2 - 1
    m2.train_loop(0, opt, batch, lab)
    mse = m2.test_loop(5, inputs=batch, targets=lab)
    assert mse.dim() == 0 and torch.isfinite(mse)
    assert all(torch.isfinite(p).all() for p in m2.parameters())


def test_golden_spectral_regression_episode(cuda):
    """The committed spectral-mixture fixture through the product path: loss, alpha, every gradient, the prediction."""
    g = np.load(os.path.join(GOLD, "regression_spectral_q4.npz"))
    d = g["z"].shape[1]
    m = dkt_amd.DKTRegression(torch.nn.Identity(), "spectral", ard_num_dims=d).to(cuda)
    inv = lambda v: torch.log(torch.expm1(torch.as_tensor(v, dtype=torch.float64))).float()      # softplus^-1
    with torch.no_grad():
        m.model.raw_mixture_weights.copy_(inv(g["weights"]))
        m.model.raw_mixture_means.copy_(inv(g["means"]).reshape(4, 1, d))
        m.model.raw_mixture_scales.copy_(inv(g["scales"]).reshape(4, 1, d))
        m.model.mean_constant.copy_(torch.as_tensor(g["mean"]).float())
        m.model.raw_noise.copy_(inv(g["noise"] - 1e-4))
    z = dev_t(g["z"], cuda).requires_grad_(True)
    hy = m.model
    w, mu, sg = [v.detach().requires_grad_(True) for v in (hy.mixture_weights, hy.mixture_means, hy.mixture_scales)]
    e = ops.spectral_mixture_matrix(z.unsqueeze(0), w, mu, sg)
    np.testing.assert_allclose(e[0].detach().cpu().numpy(), g["e"], rtol=2e-5, atol=2e-6)
    cw = torch.full((1,), -1.0 / 19, device=cuda)
    obj, logp, alpha, info, jit = ops.mll_objective(e, dev_t(g["labels"], cuda).reshape(1, 1, -1), hy.scale_times_variance(),
                                                    hy.mean.detach(), hy.noise.detach(), cw)
    obj.sum().backward()
    assert int(info.abs().sum()) == 0 and float(jit.abs().sum()) == 0.0
    assert abs(obj.item() - float(g["loss"])) < MLL_RTOL * abs(float(g["loss"]))
    assert rel_l2(alpha[0].cpu().numpy(), g["alpha"]) < 1e-4
    for name, got, ref in (("dz", z.grad, g["dz"]), ("dweights", w.grad, g["dweights"]), ("dmeans", mu.grad.reshape(4, d), g["dmeans"]),
                           ("dscales", sg.grad.reshape(4, d), g["dscales"])):
        assert rel_l2(got.cpu().numpy(), ref) <= GRAD_RTOL, name
    sup = g["support"].tolist()
    pm, pv = m.predict(z.detach()[sup], dev_t(g["labels"], cuda)[sup], z.detach(), with_variance=True)
    np.testing.assert_allclose(pm.cpu().numpy(), g["pred_mean"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pv.cpu().numpy(), g["pred_var"], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("flags", [[], ["--spectral"]])
def test_regression_drivers_end_to_end(cuda, tmp_path, monkeypatch, capsys, flags):
    """train_regression.py -> checkpoint file -> test_regression.py report (reference train_regression.py / test_regression.py)."""
    import importlib
    monkeypatch.chdir(tmp_path)
    tr = importlib.import_module("train_regression")
    te = importlib.import_module("test_regression")
    model = tr.main(["--seed", "2", "--stop_epoch", "2"] + flags)
    ckpt = tmp_path / "save" / "checkpoints" / "synthetic" / "Conv3_DKT"
    assert ckpt.is_file()
    state = torch.load(ckpt, map_location="cpu")
    assert set(state) == {"gp", "likelihood", "net"} and "layer1.weight" in state["net"]
    assert ("raw_mixture_means" in state["gp"]) == bool(flags)
    mse = te.main(["--seed", "2", "--n_test_epochs", "3", "--n_support", "5"] + flags)
    assert len(mse) == 3 and all(np.isfinite(v) and v >= 0.0 for v in mse)
    out = capsys.readouterr().out
    assert "[0] - Loss:" in out and "Average MSE: " in out and " +- " in out


def test_drivers_end_to_end(cuda, tmp_path, monkeypatch, capsys):
    """train.py -> checkpoint -> test.py (results line) -> test_uncertainty.py on synthetic episodes."""
    import importlib
    monkeypatch.chdir(tmp_path)
    train = importlib.import_module("train")
    test = importlib.import_module("test")
    tu = importlib.import_module("test_uncertainty")
    common = ["--model", "Conv4S", "--image_size", "28", "--n_episode", "6", "--seed", "3"]
    train.main(common + ["--stop_epoch", "2", "--save_freq", "1"])
    ckpt = tmp_path / "save" / "checkpoints" / "synthetic" / "Conv4S_DKT_5way_5shot"
    assert (ckpt / "best_model.tar").exists() and (ckpt / "1.tar").exists()
    state = torch.load(ckpt / "1.tar", map_location="cpu")
    assert state["epoch"] == 1 and "feature.trunk.0.C.weight" in state["state"] and "model.raw_outputscale" in state["state"]
    train.main(common + ["--stop_epoch", "3", "--save_freq", "1", "--resume"])       # resumes at epoch 2
    assert (ckpt / "2.tar").exists()
    accs = test.main(common + ["--repeat", "2"])
    assert len(accs) == 2 and all(0.0 <= a <= 100.0 for a in accs)
    line = (tmp_path / "record" / "results.txt").read_text().strip().splitlines()[-1]
    assert "Setting: synthetic-novel-Conv4S-DKT 5shot 5way_train 5way_test" in line and "Test Acc" in line
    eces, temperature = tu.main(common + ["--repeat", "2"])
    assert len(eces) == 2 and all(0.0 <= e <= 1.0 for e in eces) and temperature > 0.0
    out = capsys.readouterr().out
    assert "Epoch [0] [0/6]" in out and "Overall Test Acc" in out and "Overall ECE" in out


def test_bench_line_is_last_on_stdout_with_rccl_initialised(cuda):
    """The graded command's contract on hardware: the LAST line of stdout parses as the JSON record even when RCCL (whose version banner
    goes through C stdio and is flushed at exit) was initialised in the process -- here by the opt-in world-1 launch-plumbing check of the
    gradient bucket (19.6 / 44.7 MB through GradBucket.allreduce_mean: views of the flat buffer, no pack copies, the failure flag in the same
    collective).  The line carries the contract's keys and stays under 8 KB."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-other-configs", "--no-cpu-baseline",
                          "--no-test-time", "--rccl-selftest"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = res.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 8000
    out = json.loads(lines[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline"):
        assert key in out
    assert out["valid"] and out["steps"] == 2 and out["warmup"] == 1 and out["config"]["workload"].startswith("cfg2")
    rc = out["rccl_selftest"]
    assert rc["valid"] and all(b["pack_copies"] == 0 and b["grads_are_views"] for b in rc["buckets"].values())
