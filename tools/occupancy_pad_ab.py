"""Occupancy A/B of the streaming kernels around the Gram / marginal-likelihood kernels (twins library): every launch of the feature-space kernels (dkt_lowrank.hip), the
large-episode front end (dkt_frontend_big.hip) and the class maps (dkt_classkernel.hip) reserves DKT_PAD_* bytes of dynamic LDS it never touches, which caps the
workgroups per CU.  One process per level (the library reads the switches per launch, but a fresh process keeps the allocator state equal).

    python tools/occupancy_pad_ab.py            # driver: levels 0 / 10 / 20 / 40 / 80 KB
    python tools/occupancy_pad_ab.py one        # one level (environment already set): prints the kernel times of the steps
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PADS = ("DKT_PAD_LR_GRAM", "DKT_PAD_LR_FIN", "DKT_PAD_LR_BWD", "DKT_PAD_AFFNORM", "DKT_PAD_ROWDOT", "DKT_PAD_NBB", "DKT_PAD_CK_FWD", "DKT_PAD_CK_BWD")

if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from dkt_amd import ops
    dev = torch.device("cuda", 0)
    for cfg, b in (("cfg1", 8192), ("cfg1_20way", 2048)):
        step, _ = bench._workload(cfg, b, dev, 0, True)
        for _ in range(3):
            step()
        ops.kernel_timing(True)
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        print("  %s (%d): %s" % (cfg, b, {k: round(v[1], 4) for k, v in ops.kernel_timing_results().items()}), flush=True)
        ops.kernel_timing(False)
        del step
    for cfg, b1, b2 in (("cfg4", 512, 64), ("cfg2", 2048, 2048)):
        r = bench._aux_paths(dev, cfg, b1, b2)
        for path in ("from_trunk_features", "rbf_per_class_lengthscales"):
            if path in r and not (cfg == "cfg2" and path == "from_trunk_features"):
                print("  %s %s: %s" % (cfg, path, r[path]["kernels_ms"]), flush=True)
else:
    for level in (0, 10000, 20000, 40000, 80000):
        env = dict(os.environ, DKT_TWINS="force")
        for p in PADS:
            env[p] = str(level)
        print("dynamic LDS per workgroup: %d bytes (at most %s workgroups per CU)" % (level, "all" if level == 0 else str(163840 // level)), flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, check=False)
