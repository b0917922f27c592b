"""Wall time per episode of the drop-in DKT.train_loop / test_loop (level-2 context of SURVEY.md 8d: synthetic images through
the backbone + the GP hot path, one Adam step per episode as the reference does).  Measurement tooling; prints only."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd  # noqa: E402


class Loader:
    def __init__(self, n_ep, n_way, per, hw, seed):
        # DKT_DEVICE_DATA=1: the episodes already live on the GPU (what an input pipeline that decodes on the device would hand over)
        d = torch.device("cuda", 0) if os.environ.get("DKT_DEVICE_DATA", "0") == "1" else torch.device("cpu")
        g = torch.Generator(device=d).manual_seed(seed)
        self.x = [torch.rand(n_way, per, 3, hw, hw, generator=g, device=d) for _ in range(min(n_ep, 256))]
        self.n = n_ep

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter((self.x[i % len(self.x)], None) for i in range(self.n))


dev = torch.device("cuda", 0)
for name, hw in (("Conv4S", 28), ("Conv4", 84), ("ResNet10", 224)):
    m = dkt_amd.DKT(getattr(dkt_amd.backbone, name), n_way=5, n_support=5).to(dev)
    mb = int(os.environ.get("DKT_META_BATCH", "1"))          # episodes per Adam step (train.py --meta_batch)
    m.meta_batch = mb
    n_ep = (40 if name != "ResNet10" else 12) * (mb if mb > 1 else 1) // (4 if mb >= 16 else 1)
    if mb > 1 and name == "ResNet10":
        continue
    ld = Loader(n_ep, 5, 21, hw, 0)
    m.train()
    sys.stdout = open(os.devnull, "w")
    m.train_loop(0, Loader(3 * mb, 5, 21, hw, 1), None, print_freq=1000)      # warm-up (MIOpen find, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.train_loop(1, ld, None, print_freq=1000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_ep
    m.eval()
    lt = Loader(n_ep, 5, 20, hw, 2)
    m.test_loop(Loader(3, 5, 20, hw, 3))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    m.test_loop(lt)
    torch.cuda.synchronize()
    dte = (time.perf_counter() - t1) / n_ep
    sys.stdout = sys.__stdout__
    print("device_data=%s meta_batch=%d graph=%s fused_adam=%s %-9s %3dx%-3d  train_loop %.2f ms / episode (%.0f episodes/s)   test_loop %.2f ms / episode" % (os.environ.get("DKT_DEVICE_DATA", "0"), mb, os.environ.get("DKT_TRAIN_GRAPH", "0"), os.environ.get("DKT_FUSED_ADAM", "1"), name, hw, hw, 1e3 * dt, 1 / dt, 1e3 * dte), flush=True)
