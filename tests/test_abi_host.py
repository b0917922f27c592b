"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/dkt_abi.h declares; host logic (constraints, targets, module surface); and the product path
fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch

import dkt_amd
from oracle import dkt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols(lib):
    header = open(os.path.join(ROOT, "include", "dkt_abi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(dkt_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), "libdkt_hip.so lacks %s" % name
        assert name in dkt_amd._lib.SIGNATURES, "no ctypes signature for %s" % name
    assert sorted(dkt_amd._lib.SIGNATURES) == declared
    assert lib.dkt_abi_version() == 7
    # pure host queries (no GPU needed)
    assert lib.dkt_mll_workspace_bytes(8, 5, 105) == 0                 # register resident
    # N > 127: blocked path, per (episode, class) four N x N matrices + two vectors + bookkeeping
    assert lib.dkt_mll_workspace_bytes(2, 20, 420) == (2 * 20 * (4 * 420 * 420 + 2 * 420 + 4) + 16) * 4
    assert lib.dkt_mll_workspace_bytes(2, 20, 420) >= 2 * 421 * 421 * 4  # also covers the generic kernel's global matrices
    assert lib.dkt_mll_workspace_bytes(0, 5, 105) == 0


def test_argument_errors_do_not_launch(lib):
    # NULL pointers / bad sizes are rejected on the host before any launch
    assert lib.dkt_gram_f32(None, None, None, 1, 4, 4, 4, 0, None, None) == -1
    assert lib.dkt_gram_bwd_f32(None, None, None, 1, 4, 4, None, 0, None) == -1
    assert lib.dkt_predict_f32(None, None, None, None, None, None, 1, 1, 1, 1, None) == -1
    # round 4: the front end of large episodes, the retired twin flag, the workspace of the per-class tile-array path
    assert lib.dkt_affine_normalize_f32(None, None, None, 0, None, None, 1, 4, 4, None) == -1
    assert lib.dkt_normalize_bn_bwd_f32(None, None, None, None, 0, None, None, None, None, None, None, None, 1, 4, 4, None) == -1
    assert lib.dkt_mll_workspace_bytes(1, 20, 105) == 0 and lib.dkt_mll_workspace_bytes(1, 20, 420) > 0
    # per-class base matrices need as many tiles again for E: a small batch reserves up to twice the shared-matrix workspace, a full chunk of 1024 episodes the same
    assert lib.dkt_mll_workspace_bytes(2048, 20, 420) == lib.dkt_mll_workspace_bytes(1024, 20, 420)
    assert lib.dkt_mll_workspace_bytes(1, 20, 420) >= 2 * 20 * (27 * 28 // 2) * 1024
    # the per-call query: never more than the flag-less one; a default shared-matrix call takes the tile arrays only (no per-class layout, no blocked-path matrices)
    for (b, c, n) in ((1, 20, 420), (64, 20, 420), (1024, 20, 320), (3, 5, 150), (2, 5, 500)):
        for flags in (0, 1, 2, 4, 16, 64, 65, 1 | 16):
            assert lib.dkt_mll_workspace_bytes_for(b, c, n, flags) <= lib.dkt_mll_workspace_bytes(b, c, n), (b, c, n, flags)
    assert lib.dkt_mll_workspace_bytes_for(1, 20, 420, 1) < lib.dkt_mll_workspace_bytes(1, 20, 420) // 4
    assert lib.dkt_mll_workspace_bytes_for(1, 20, 420, 1 | 64) >= 2 * 20 * (27 * 28 // 2) * 1024
    assert lib.dkt_mll_workspace_bytes_for(8, 5, 105, 1) == 0 and lib.dkt_mll_workspace_bytes_for(8, 5, 105, 2) == 0


def test_twins_library_is_the_same_abi_and_the_product_has_no_variant_switches():
    """libdkt_twins.so = the same sources with -DDKT_TWINS: every symbol of the header, the same ABI version.  The product library must not read a variant
    switch: its binary contains none of their names (they are compiled out), only the four documented product switches."""
    twins = dkt_amd._lib.load_twins()
    assert twins.dkt_abi_version() == dkt_amd._lib.abi_version_of_header()
    for name in dkt_amd._lib.SIGNATURES:
        assert hasattr(twins, name), name
    prod = open(dkt_amd._lib.LIB_PATH, "rb").read()
    twin = open(dkt_amd._lib.TWINS_LIB_PATH, "rb").read()
    for name in dkt_amd.ops._VARIANT_SWITCHES:
        assert name.encode() + b"\0" not in prod, name
    for name in ("DKT_GRAM_UNIT_VAR", "DKT_MLL_TILED_WRES", "DKT_GRAM_SPLIT"):
        assert name.encode() + b"\0" in twin, name
    for name in dkt_amd.ops._PRODUCT_SWITCHES:
        assert name.encode() + b"\0" in prod, name
    upath = os.path.join(os.path.dirname(dkt_amd._lib.LIB_PATH), "build", "libdkt_hip.so.resource_usage.json")       # written by the build in this checkout
    if os.path.exists(upath):
        usage = __import__("json").load(open(upath))
        # (250: + the four RBF instances of the N <= 32 Gram forward that form rows 17 .. 20 on the VALU, end of round 6)
        assert len(usage) <= 250 and max(u.get("vgpr_spill", 0) for u in usage.values()) <= 28
        assert dkt_amd._lib.check_resources(usage) == []


def test_no_wide_buffer_store_is_followed_by_a_write_of_its_data_registers():
    """A 16-byte buffer store with an SGPR soffset gets no wait state from hipcc before a VALU instruction that rewrites its data registers, and gfx950
    needs one (round 5; _lib.unprotected_wide_buffer_stores).  Audit of the device code of both libraries: no such pair; and the scanner finds the pair in
    a code object that is known to contain it (a probe compiled here)."""
    import subprocess, tempfile, textwrap
    L = dkt_amd._lib
    assert len(L.device_code_objects(L.LIB_PATH)) >= len(L.SOURCES)
    assert L.unprotected_wide_buffer_stores(L.LIB_PATH) == []
    # positive control: the product library's 16-byte buffer stores were seen AND matched the operand syntax the audit expects
    assert L.AUDIT_STATS["wide_stores_any"] >= 100 and L.AUDIT_STATS["wide_stores_parsed"] == L.AUDIT_STATS["wide_stores_any"], L.AUDIT_STATS
    assert L.unprotected_wide_buffer_stores(L.TWINS_LIB_PATH) == []
    probe = textwrap.dedent("""
        #include <hip/hip_runtime.h>
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        __global__ void probe(float* p, int n, int so, float s) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p, 0, n, 0x00020000);
            float a = s * threadIdx.x, b = a + 1.f, c = a + 2.f, d = a + 3.f;
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
                __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)threadIdx.x * 16, so * (i + 1), 0);
                a = a * s + b; b = b * s + c; c = c * s + d; d = d * s + a;
            }
        }""")
    with tempfile.TemporaryDirectory() as td:
        src, obj = os.path.join(td, "probe.hip"), os.path.join(td, "probe.o")
        open(src, "w").write(probe)
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-c", src, "-o", obj], check=True, capture_output=True)
        assert len(L.device_code_objects(obj)) == 1
        found = L.unprotected_wide_buffer_stores(obj)
    # hipcc MAY schedule something harmless in between on another day; what must hold is that the scanner parses this form of the store at all
    assert all("buffer_store_dwordx4" in h[1] for h in found)


def test_spill_reloads_stay_out_of_the_streaming_loops():
    """A scratch reload waits `vmcnt(0)`; inside a loop that prefetches it drains the prefetch on every trip (round 5: 1.89 instead of 1.36 ms for the unit-row Gram
    forward at 112 < N <= 128).  Audit of the product library's disassembly: only the three kernels whose reloads were looked at and accepted -- the wave-per-matrix
    marginal likelihood at NT = 7 (class loop; batches under 1024 episodes) and the tile-array factor / invert at 3 workgroups per CU (around the diagonal-tile sweeps,
    not in the K loop; 3 vs 2 workgroups re-measured in profiles/r05/v14_tiled_wgs_ab.log) -- reload a spilled register inside a loop that also loads from memory."""
    import re
    found = dkt_amd._lib.spill_reloads_in_streaming_loops(dkt_amd._lib.LIB_PATH)
    # round 6: + the two-sided kernels of the band reduction (256 VGPRs: 27 partial accumulators + the operands of the fused pass).  Their reloads sit in the panel
    # loop -- one per Householder column of the forward kernel, a few per panel in both -- and none in the unrolled tile loop of the pass (the ISA was read:
    # profiles/r06/INDEX.md); the panel loop "also loads from memory", which is what this audit keys on.
    allowed = (r"mll_h2_kernelILi7E", r"tiled_factor_kernelILi7ELb1ELi3E", r"tiled_invert_kernelILi7ELb1ELb1ELi3E", r"band_sym_kernelILb[01]E")
    # positive controls: the scanner really parsed branches, backward branches and kernels, and it does find the kernels that are known to reload
    st = dkt_amd._lib.AUDIT_STATS
    assert st["kernels"] >= 200 and st["branches"] >= 1000 and st["backward_branches"] >= 200, st
    assert any(re.search(r"mll_h2_kernelILi7E", k) for k in found) and any(re.search(r"band_sym_kernelILb0E", k) for k in found), sorted(found)
    for k in found:
        assert any(re.search(a, k) for a in allowed), (k, found[k])
    for hot in ("gram_sym_ep_split_kernel", "gram_bwd_ep_f16x2_kernel", "mll_h2e_kernel", "gram_bn_train_f16_kernel", "gram_bn_bwd_ep_kernel", "lowrank_", "gram_small"):
        assert not any(hot in k for k in found), hot


def test_library_selection_product_unless_twins_are_asked_for(monkeypatch):
    """ops._lib_now(): the product library for every call, the twins library only with DKT_TWINS=1 AND (a variant switch set | the call names a twin)."""
    ops, L = dkt_amd.ops, dkt_amd._lib
    for k in ops._VARIANT_SWITCHES + ("DKT_MLL_F32MFMA",):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("DKT_TWINS", "1")
    assert ops._lib_now()._name == L.LIB_PATH
    assert ops._lib_now(want_twin=True)._name == L.TWINS_LIB_PATH
    monkeypatch.setenv("DKT_GRAM_UNIT_VAR", "2223")
    assert ops._lib_now()._name == L.TWINS_LIB_PATH
    monkeypatch.setenv("DKT_TWINS", "0")
    assert ops._lib_now()._name == L.LIB_PATH and ops._lib_now(want_twin=True)._name == L.LIB_PATH      # no opt-in: the variant switches have no effect
    monkeypatch.delenv("DKT_GRAM_UNIT_VAR")
    monkeypatch.setenv("DKT_TWINS", "1")
    monkeypatch.setenv("DKT_MLL_H2E_MINB", "1")                  # a product switch: stays on the product library
    assert ops._lib_now()._name == L.LIB_PATH


def test_per_class_path_sizes():
    """Which (N, C) the one-launch per-class path serves (ops.mll_per_class_supported mirrors dkt_mll_f32 with DKT_MLL_E_PER_CLASS and dkt_class_kernel_bwd_f32)."""
    ok = dkt_amd.ops.mll_per_class_supported
    assert ok(105, 5) and ok(111, 5) and ok(112, 5) and ok(127, 5)
    assert ok(128, 5) and ok(420, 20) and ok(447, 20) and not ok(448, 20)
    assert ok(105, 32) and not ok(105, 33)
    assert dkt_amd.ops.FUSED_EP_MAX_N == 128


def test_feature_space_dispatch_rule(monkeypatch):
    """ops.lowrank_applies: D <= 64, D % 4 == 0, C <= 32, N >= 80; by default every batch of episodes with more than 128 rows and batches >= LOWRANK_MIN_B of shorter
    ones (never behind the fused front end at N <= 128); DKT_LOWRANK=0 never, =force wherever supported."""
    ap, ops = dkt_amd.ops.lowrank_applies, dkt_amd.ops
    monkeypatch.delenv("DKT_LOWRANK", raising=False)
    assert ap(105, 64, 5, 8192) and ap(105, 64, 5, ops.LOWRANK_MIN_B) and not ap(105, 64, 5, ops.LOWRANK_MIN_B - 1) and not ap(105, 64, 5, 1)
    assert ap(420, 64, 20, 1) and ap(129, 32, 5, 1) and ap(420, 64, 20, 1, front_end=True) and not ap(105, 64, 5, 8192, front_end=True)
    assert not ap(105, 1600, 5, 8192) and not ap(105, 62, 5, 8192) and not ap(79, 64, 5, 8192) and not ap(420, 64, 33, 8192) and not ap(25, 64, 5, 8192)
    monkeypatch.setenv("DKT_LOWRANK", "0")
    assert not ap(420, 64, 20, 8192) and not ap(105, 64, 5, 8192)
    monkeypatch.setenv("DKT_LOWRANK", "force")
    assert ap(105, 64, 5, 1) and ap(105, 64, 5, 1, front_end=True) and not ap(105, 1600, 5, 1)


def test_product_path_fails_loudly_on_cpu_tensors():
    with pytest.raises(RuntimeError, match="HIP-only"):
        dkt_amd.ops.gram(torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match="HIP-only"):
        dkt_amd.ops.episode_loss_linear(torch.zeros(1, 4, 8), torch.zeros(2, 4), torch.ones(2), torch.zeros(2),
                                        torch.ones(2), torch.ones(2))
    # the product package never imports the oracle
    import sys
    for name, mod in list(sys.modules.items()):
        if name.startswith("dkt_amd"):
            src = getattr(mod, "__file__", None)
            if src and src.endswith(".py"):
                assert "oracle" not in open(src).read().replace("the oracle", ""), name


def test_hyper_parameter_constraints_match_oracle():
    h = dkt_amd.gp.ExactGPHypers(5, "bncossim", fixed_noise=0.1)
    assert torch.allclose(h.outputscale, torch.full((5,), float(np.log(2.0))), atol=1e-7)
    assert torch.allclose(h.noise, torch.full((5,), 0.1), atol=1e-7)
    assert torch.allclose(h.variance, torch.ones(1), atol=1e-6)
    assert not h.raw_noise.requires_grad and not h.raw_variance.requires_grad
    assert [n for n, p in h.named_parameters() if p.requires_grad] == ["mean_constant", "raw_outputscale"]
    r = dkt_amd.gp.ExactGPHypers(1, "rbf", fixed_noise=None)
    assert abs(r.noise.item() - (np.log(2.0) + 1e-4)) < 1e-6      # GaussianLikelihood default
    assert abs(r.lengthscale.item() - np.log(2.0)) < 1e-6
    assert r.raw_noise.requires_grad and r.raw_lengthscale.requires_grad
    with torch.no_grad():
        h.raw_outputscale.copy_(torch.tensor([-1.0, 0.0, 0.5, 2.0, 40.0]))
    np.testing.assert_allclose(h.outputscale.detach().numpy(), O.softplus([-1.0, 0.0, 0.5, 2.0, 40.0]), rtol=1e-6)
    with pytest.raises(ValueError):
        dkt_amd.gp.ExactGPHypers(5, "nope")
    # model views expose the attribute paths the reference's logging reads (DKT.py:148-154)
    mv = h.models[2]
    assert mv.covar_module.base_kernel.lengthscale is None
    assert abs(mv.likelihood.noise.item() - 0.1) < 1e-6


def test_dkt_surface_and_state_dict_names():
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=5, n_support=5)
    for name in ("train_loop", "test_loop", "correct", "get_logits", "set_forward", "set_forward_loss",
                 "init_summary", "get_model_likelihood_mll", "parse_feature"):
        assert callable(getattr(m, name))
    assert m.feature is m.feature_extractor and m.normalize and m.feat_dim == 64
    assert isinstance(m.feature_extractor.trunk.bn_out, torch.nn.BatchNorm1d)
    keys = set(m.state_dict().keys())
    assert "feature.trunk.0.C.weight" in keys and "feature_extractor.trunk.0.trunk.0.weight" in keys
    assert "feature.trunk.bn_out.running_mean" in keys and "model.raw_outputscale" in keys
    assert m.set_forward(None) is None and m.set_forward_loss(None) is None
    y = m._targets(3, 2, torch.device("cpu"))
    np.testing.assert_array_equal(y.numpy(), O.one_vs_rest_targets(3, 2))
    # optimizer groups exactly as DKT.py:114-115
    n_gp = sum(p.numel() for p in m.model.parameters() if p.requires_grad)
    assert n_gp == 10
    reg = dkt_amd.DKTRegression(dkt_amd.backbone.Conv3(), "rbf")
    assert sum(p.numel() for p in reg.model.parameters() if p.requires_grad) == 4   # mean, outputscale, lengthscale, noise
    with pytest.raises(ValueError):
        dkt_amd.DKTRegression(dkt_amd.backbone.Conv3(), "bncossim")


def test_reference_checkpoint_key_loader():
    h = dkt_amd.gp.ExactGPHypers(2, "bncossim")
    state = {"model.models.0.mean_module.constant": torch.tensor([0.3]),
             "model.models.1.covar_module.raw_outputscale": torch.tensor(-0.7),
             "model.models.1.likelihood.noise_covar.raw_noise": torch.tensor([0.2])}
    assert h.load_reference_state_dict(state) == 3
    assert abs(h.mean_constant[0].item() - 0.3) < 1e-7 and abs(h.raw_outputscale[1].item() + 0.7) < 1e-7


def test_reference_format_checkpoints_load_through_the_drivers_entry_points(tmp_path):
    """A state dict with the reference's key names (GPyTorch module tree, methods/DKT.py:58-71; every class model owns its
    base-kernel parameters) loads through DKT.load_state_dict -- what test.py / train.py --resume / test_uncertainty.py call --
    and the reference's regression checkpoint layout through DKTRegression.load_checkpoint."""
    m = dkt_amd.DKT(dkt_amd.backbone.Conv4S, n_way=3, n_support=2, kernel_type="rbf")
    own = m.state_dict()
    state = {("feature." + k[len("feature_extractor."):]): v.clone() + 0.25 for k, v in own.items() if k.startswith("feature_extractor.")}
    state.update({k: v for k, v in own.items() if k.startswith("feature_extractor.")})     # the reference saves both aliases
    for c in range(3):
        b = "model.models.%d." % c
        state[b + "mean_module.constant"] = torch.tensor([0.1 * (c + 1)])
        state[b + "covar_module.raw_outputscale"] = torch.tensor(-0.5 * c)
        state[b + "covar_module.base_kernel.raw_lengthscale"] = torch.tensor([[1.0 + c]])
        state[b + "likelihood.noise_covar.raw_noise"] = torch.tensor([-2.0])
        state["likelihood.likelihoods.%d.noise_covar.raw_noise" % c] = torch.tensor([-2.0])
        state["mll.model.models.%d.mean_module.constant" % c] = torch.tensor([0.1 * (c + 1)])
    m.load_state_dict(state)
    assert torch.allclose(m.model.mean_constant, torch.tensor([0.1, 0.2, 0.3]))
    assert torch.allclose(m.model.raw_outputscale, torch.tensor([0.0, -0.5, -1.0]))
    assert torch.allclose(m.model.raw_lengthscale, torch.tensor([1.0, 2.0, 3.0]))          # per class, none dropped
    assert torch.allclose(m.model.raw_noise, torch.full((3,), -2.0))
    with pytest.raises(RuntimeError):
        m.load_state_dict({"model.models.0.mean_module.constant": torch.zeros(1)})          # no backbone tensors
    reg = dkt_amd.DKTRegression(dkt_amd.backbone.Conv3(), "rbf")
    path = str(tmp_path / "ref_regression_ckpt")
    torch.save({"gp": {"likelihood.noise_covar.raw_noise": torch.tensor([0.4]), "mean_module.constant": torch.tensor([0.7]),
                       "covar_module.raw_outputscale": torch.tensor(0.2), "covar_module.base_kernel.raw_lengthscale": torch.tensor([[1.5]])},
                "likelihood": {"noise_covar.raw_noise": torch.tensor([0.4])}, "net": reg.feature_extractor.state_dict()}, path)
    reg.load_checkpoint(path)
    assert abs(reg.model.mean_constant.item() - 0.7) < 1e-7 and abs(reg.model.raw_lengthscale.item() - 1.5) < 1e-7
    assert abs(reg.model.raw_noise.item() - 0.4) < 1e-7 and abs(reg.model.raw_outputscale.item() - 0.2) < 1e-7
    own_path = str(tmp_path / "own_ckpt")
    reg.save_checkpoint(own_path)
    reg.load_checkpoint(own_path)
    # the spectral-mixture head (DKT_regression.py:121-122: SpectralMixtureKernel(num_mixtures=4, ard_num_dims=2916), no ScaleKernel):
    # its three raw tensors load too; a GP tensor nothing consumes raises instead of being dropped
    sp = dkt_amd.DKTRegression(dkt_amd.backbone.Conv3(), "spectral")
    q, dd = sp.model.raw_mixture_weights.shape[0], sp.model.raw_mixture_means.shape[-1]
    gp = {"likelihood.noise_covar.raw_noise": torch.tensor([0.3]), "mean_module.constant": torch.tensor([-0.2]),
          "covar_module.raw_mixture_weights": torch.linspace(-1.0, 1.0, q), "covar_module.raw_mixture_means": torch.full((q, 1, dd), 0.25),
          "covar_module.raw_mixture_scales": torch.full((q, 1, dd), -0.75)}
    sp_path = str(tmp_path / "ref_spectral_ckpt")
    torch.save({"gp": gp, "likelihood": {"noise_covar.raw_noise": torch.tensor([0.3])}, "net": sp.feature_extractor.state_dict()}, sp_path)
    sp.load_checkpoint(sp_path)
    assert torch.allclose(sp.model.raw_mixture_weights, torch.linspace(-1.0, 1.0, q))
    assert torch.allclose(sp.model.raw_mixture_means, torch.full((q, 1, dd), 0.25)) and torch.allclose(sp.model.raw_mixture_scales, torch.full((q, 1, dd), -0.75))
    assert abs(sp.model.mean_constant.item() + 0.2) < 1e-7 and abs(sp.model.raw_noise.item() - 0.3) < 1e-7
    with pytest.raises(RuntimeError):
        reg.load_checkpoint(sp_path)                       # an RBF head must not swallow a spectral checkpoint silently


def test_backbone_shapes():
    bb = dkt_amd.backbone
    x = torch.randn(2, 3, 28, 28)
    assert bb.Conv4S()(x).shape == (2, 64)
    assert bb.Conv4()(torch.randn(2, 3, 84, 84)).shape == (2, 1600)
    assert bb.Conv3()(torch.randn(2, 3, 100, 100)).shape == (2, 2916)
    assert sum(p.numel() for p in bb.ResNet10().parameters()) == 4905792
