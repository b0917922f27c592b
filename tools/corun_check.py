"""dkt_mll_f32 on one stream while bf16-split Gram kernels run on another: number of episodes whose alpha differs from the
single-stream result (0 expected).  DKT_AMD_LIB selects a variant library.  Measurement tooling."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_amd
from dkt_amd import ops
dev = torch.device("cuda", 0)
B, N, D, C = 8192, 105, 1600, 5
z = torch.nn.functional.normalize(torch.randn(B, N, D, device=dev), dim=2).contiguous()
cls = torch.arange(C, device=dev).repeat_interleave(N // C)
y = torch.where(cls.unsqueeze(0) == torch.arange(C, device=dev).unsqueeze(1), 1.0, -1.0).contiguous()
sv = torch.full((C,), 0.69, device=dev) + 0.01 * torch.arange(C, device=dev)
mean = torch.zeros(C, device=dev); noise = torch.full((C,), 0.1, device=dev)
e = ops.gram(z, None, ops.KERNEL_LINEAR_UNIT)
torch.cuda.synchronize()
ref = ops.mll(e, y, sv, mean, noise)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
worst = 0
for rep in range(6):
    with torch.cuda.stream(s2):
        for _ in range(3):
            ops.gram(z, None, ops.KERNEL_LINEAR)
    with torch.cuda.stream(s1):
        a = ops.mll(e, y, sv, mean, noise)
    torch.cuda.synchronize()
    worst = max(worst, int((a["alpha"] != ref["alpha"]).flatten(1).any(1).sum()))
print(os.path.basename(os.environ.get("DKT_AMD_LIB", "default")), ": episodes with different alpha under co-run (max of 6):", worst)
