"""GPU diagnostics: per-kernel numeric errors vs the oracle + per-kernel timings. Prints, never asserts.
(test/measurement tooling; may import the oracle)"""
import os as _os; _os.environ.setdefault("DKT_TWINS", "1")   # the variant switches this tool flips live in libdkt_twins.so (ops._lib_now)
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402
from dkt_amd import ops  # noqa: E402
from oracle import dkt_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
print("device:", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName,
      "CUs", dkt_amd._lib.load().dkt_device_cu_count(), flush=True)


def t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)


def section(name, fn):
    print("==== " + name, flush=True)
    try:
        fn()
    except Exception:  # noqa: BLE001
        traceback.print_exc()
    sys.stdout.flush()


def gram_checks():
    for (b, n, d) in [(1, 1, 4), (2, 5, 7), (2, 19, 2916), (2, 64, 32), (2, 65, 36), (2, 105, 64), (2, 105, 1600), (1, 420, 512)]:
        z = np.random.default_rng(n + d).standard_normal((b, n, d)).astype(np.float32)
        e = ops.gram(t(z)).cpu().numpy()
        ref = np.einsum("bnd,bmd->bnm", z.astype(np.float64), z.astype(np.float64))
        print("gram sym", (b, n, d), "max abs err %.3e" % np.abs(e - ref).max(), "ref max %.2f" % np.abs(ref).max(),
              "symmetric", bool((e == e.transpose(0, 2, 1)).all()))
    for (b, m, n, d) in [(2, 75, 25, 64), (1, 300, 100, 512), (1, 7, 3, 5)]:
        rng = np.random.default_rng(m)
        a, bm = rng.standard_normal((b, m, d)).astype(np.float32), rng.standard_normal((b, n, d)).astype(np.float32)
        e = ops.gram(t(a), t(bm)).cpu().numpy()
        ref = np.einsum("bmd,bnd->bmn", a.astype(np.float64), bm.astype(np.float64))
        print("gram cross", (b, m, n, d), "max abs err %.3e" % np.abs(e - ref).max())
    for (n, d, ls, sh) in [(19, 2916, 30.0, 0.4), (105, 64, 1.3, 0.0), (70, 33, 0.9, 5.0)]:
        z = (np.abs(np.random.default_rng(n).standard_normal((1, n, d))) * 0.5 + sh).astype(np.float32)
        e = ops.gram(t(z), None, ops.KERNEL_RBF, t([ls])).cpu().numpy()
        print("gram rbf", (n, d, ls, sh), "max abs err %.3e" % np.abs(e[0] - O.gram_rbf(z[0].astype(np.float64), None, ls)).max())


def mll_checks():
    for (c, per, d, corr) in [(1, 1, 8, 0), (5, 5, 64, 0), (5, 21, 64, 0), (5, 21, 1600, 5), (5, 38, 32, 0), (20, 21, 512, 20)]:
        n = c * per
        z = O.synthetic_features(1, n, d, 5, corr)
        hyp = O.perturbed_hypers(c, 6)
        y = O.one_vs_rest_targets(c, per)
        cw = np.full(c, -1.0 / (c * n))
        out = ops.mll(ops.gram(t(z)), t(y), t(hyp.outputscale), t(hyp.mean), t(hyp.noise), want_grad=True, want_chol=True,
                      cls_weight=t(cw))
        torch.cuda.synchronize()
        e = O.gram_linear(z[0])
        res = O.mll_terms(e, y, hyp.outputscale, hyp.mean, hyp.noise)
        w_ref, dsv, dmean, dnoise = O.mll_grads(e, res, hyp.outputscale, hyp.noise, np.ones(c))
        w_ref2 = O.mll_grads(e, res, hyp.outputscale, hyp.noise, cw)[0]
        rl = lambda a, b_: np.linalg.norm(np.asarray(a, np.float64) - b_) / max(np.linalg.norm(b_), 1e-30)  # noqa: E731
        print("mll", (c, per, d, corr), "info", out["info"].cpu().tolist()[0][:3], "logp rel %.2e" %
              np.abs((out["logp"][0].cpu().numpy() - res.logp) / res.logp).max(),
              "alpha %.2e chol %.2e W %.2e dsv %.2e dmean %.2e dnoise %.2e" % (
                  rl(out["alpha"][0].cpu().numpy(), res.alpha), rl(out["chol"][0].cpu().numpy(), res.chol),
                  rl(out["w"][0].cpu().numpy(), w_ref2), rl(out["dsv"][0].cpu().numpy(), dsv),
                  rl(out["dmean"][0].cpu().numpy(), dmean), rl(out["dnoise"][0].cpu().numpy(), dnoise)))


def bwd_checks():
    for (b, n, d) in [(2, 5, 12), (2, 105, 64), (2, 105, 1600), (1, 420, 512)]:
        rng = np.random.default_rng(n)
        w, z = rng.standard_normal((b, n, n)).astype(np.float32), rng.standard_normal((b, n, d)).astype(np.float32)
        dz = ops.gram_bwd(t(w), t(z)).cpu().numpy()
        ref = np.stack([O.gram_linear_bwd(w[i].astype(np.float64), z[i].astype(np.float64)) for i in range(b)])
        print("gram_bwd", (b, n, d), "rel %.3e" % (np.linalg.norm(dz - ref) / np.linalg.norm(ref)))


def timings():
    c, n, d = 5, 105, 1600
    for b in (1, 64, 1024, 4096):
        z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev), dim=2)
        y = t(O.one_vs_rest_targets(c, n // c))
        sv, mean, noise = t(np.full(c, 0.7)), t(np.zeros(c)), t(np.full(c, 0.1))
        cw = t(np.full(c, -1.0 / (c * n)))
        e = ops.gram(z)
        out = ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw)
        w = out["w"]

        def tm(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            bb.record()
            torch.cuda.synchronize()
            return a.elapsed_time(bb) / reps
        tg = tm(lambda: ops.gram(z))
        tmf = tm(lambda: ops.mll(e, y, sv, mean, noise))
        tmg = tm(lambda: ops.mll(e, y, sv, mean, noise, want_grad=True, cls_weight=cw))
        tb = tm(lambda: ops.gram_bwd(w, z))
        tot = tg + tmg + tb
        print("B=%5d  gram %.3f ms (%.0f GB/s, %.1f TF)  mll fwd %.3f ms  mll fwd+grad %.3f ms  gram_bwd %.3f ms (%.0f GB/s)  "
              "sum %.3f ms -> %.0f eps/s" % (b, tg, 4 * (n * d + n * n) * b / tg / 1e6, 2 * n * n * d * b / tg / 1e9, tmf, tmg, tb,
                                            4 * (2 * n * d + n * n) * b / tb / 1e6, tot, b / tot * 1e3), flush=True)
    # test-time shape
    b, ns, mq, d = 600, 25, 75, 1600
    zs = torch.nn.functional.normalize(torch.randn(b, ns, d, device=dev), dim=2)
    zq = torch.nn.functional.normalize(torch.randn(b, mq, d, device=dev), dim=2)
    y = t(O.one_vs_rest_targets(5, 5))
    sv, mean, noise = t(np.full(5, 0.7)), t(np.zeros(5)), t(np.full(5, 0.1))

    def test_ep():
        o = ops.mll(ops.gram(zs), y, sv, mean, noise)
        return ops.predict(ops.gram(zq, zs), o["alpha"], sv, mean)
    test_ep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        test_ep()
    torch.cuda.synchronize()
    print("test episodes (25->75, D=1600): %.0f eps/s" % (600 * 10 / (time.perf_counter() - t0)))


def ep_vs_v0():
    b, n, d = 256, 105, 1600
    g = torch.Generator(device=dev).manual_seed(3)
    z = torch.nn.functional.normalize(torch.randn(b, n, d, device=dev, generator=g), dim=2)
    w = torch.randn(b, n, n, device=dev, generator=g) * 0.01
    sc = torch.full((b,), 2.5, device=dev)
    for name, ww in (("asym", w), ("sym", 0.5 * (w + w.transpose(1, 2)))):
        os.environ["DKT_GRAM_EP"] = "1"
        a1 = ops.gram_bwd(ww, z)
        a2 = ops.gram_bwd(ww, z, sc)
        e1 = ops.gram(z)
        os.environ["DKT_GRAM_EP"] = "0"
        b1 = ops.gram_bwd(ww, z)
        b2 = ops.gram_bwd(ww, z, sc)
        e0 = ops.gram(z)
        os.environ["DKT_GRAM_EP"] = "1"
        ref = torch.matmul((ww + ww.transpose(1, 2)).double(), z.double())
        def mx(x, y):
            dd = (x.double() - y.double()).abs()
            i = int(dd.argmax())
            return "%.3e at %s (vals %.6e %.6e)" % (dd.max().item(), tuple(int(v) for v in np.unravel_index(i, dd.shape)),
                                                   x.flatten()[i].item(), y.flatten()[i].item())
        print(name, "ep vs f64:", mx(a1, ref), "| v0 vs f64:", mx(b1, ref))
        print(name, "ep scaled vs 2.5*ep:", mx(a2, 2.5 * a1), "| v0 scaled vs 2.5*v0:", mx(b2, 2.5 * b1))
        print(name, "gram ep vs v0:", mx(e1, e0))


section("ep_vs_v0", ep_vs_v0)
section("gram", gram_checks)
section("mll", mll_checks)
section("gram_bwd", bwd_checks)
section("timings", timings)
print("diag done")
