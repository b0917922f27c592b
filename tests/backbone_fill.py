"""Deterministic parameter / input fill shared by tests/golden/make_backbone_golden.py (which runs the REFERENCE's backbone.py)
and tests/test_backbone_reference.py (which runs this repo's backbones): every tensor of a state dict is overwritten in key
order from one seeded generator, so no weights have to be stored with the fixture."""
import torch

CASES = [  # name, image size, batch, input channels
    ("Conv4", 84, 4, 3), ("Conv4S", 28, 4, 3), ("Conv6", 84, 3, 3), ("ResNet10", 224, 2, 3), ("ResNet18", 224, 2, 3),
    ("Conv3", 100, 3, 3),
]


def fill_state(module, seed):
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            v.zero_()
        elif k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.75)
        elif k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
        elif v.dim() == 1 and (".BN." in k or "bn" in k.lower()) and k.endswith("weight"):
            v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.75)
        elif v.dim() == 1:
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
        else:
            fan_in = v[0].numel()
            v.copy_(torch.randn(v.shape, generator=g) * (1.5 / fan_in) ** 0.5)
    module.load_state_dict(sd)
    return module


def make_input(size, batch, channels, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    return torch.rand(batch, channels, size, size, generator=g)


def run_case(module, x):
    """Train-mode forward + backward of sum(out^2)/numel, then eval-mode forward.  Returns the arrays the fixture pins."""
    module.train()
    out = module(x)
    loss = (out.double() ** 2).mean()
    module.zero_grad()
    loss.backward()
    first_w = next(p for p in module.parameters() if p.dim() == 4)
    sd = module.state_dict()
    rms = [k for k in sd if k.endswith("running_mean")]
    res = {
        "train_out_head": out.detach()[:, :24].clone(), "train_out_sum": out.detach().double().sum(), "train_out_abs": out.detach().double().abs().sum(),
        "loss": loss.detach(), "grad_first_conv_sum": first_w.grad.double().sum(), "grad_first_conv_abs": first_w.grad.double().abs().sum(),
        "running_mean_last": sd[rms[-1]].clone() if rms else torch.zeros(1),      # Conv3 has no BatchNorm
    }
    module.eval()
    with torch.no_grad():
        oe = module(x)
    res["eval_out_head"] = oe[:, :24].clone()
    res["eval_out_sum"] = oe.double().sum()
    res["shape"] = torch.tensor(list(out.shape))
    return {k: v.detach().cpu().numpy() for k, v in res.items()}
