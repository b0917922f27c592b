import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
for (n, d) in [(129, 64), (190, 512), (320, 512), (420, 512), (257, 100)]:
    rng = np.random.default_rng(n + d)
    for heavy in (0.0, 1.5):
        z = rng.standard_normal((6, n, d)) * np.exp(heavy * rng.standard_normal((6, n, d)))
        z /= np.linalg.norm(z, axis=2, keepdims=True)
        zt = torch.tensor(z, dtype=torch.float32, device=dev)
        z32 = zt.double().cpu().numpy()
        ref = np.einsum("bnd,bmd->bnm", z32, z32)
        mag = np.einsum("bnd,bmd->bnm", np.abs(z32), np.abs(z32))
        fast = ops.gram(zt, kind=ops.KERNEL_LINEAR_UNIT).cpu().numpy()
        slow = ops.gram(zt).cpu().numpy()
        print(n, d, heavy, "fast err/mag %.2e abs %.2e | slow err/mag %.2e abs %.2e | sym %s diag %.2e" % (
            (np.abs(fast - ref) / mag).max(), np.abs(fast - ref).max(), (np.abs(slow - ref) / mag).max(), np.abs(slow - ref).max(),
            bool((fast == fast.transpose(0, 2, 1)).all()), np.abs(np.diagonal(fast, axis1=1, axis2=2) - 1).max()))
