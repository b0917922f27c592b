"""CPU checks of the driver-side pieces that mirror the reference's io_utils / data contracts."""
import os

import numpy as np
import torch

import dkt_amd
from dkt_amd.data import SyntheticEpisodeLoader, SyntheticHeadPoseSampler, get_episode_loader
from dkt_amd.io_utils import (checkpoint_dir_for, default_image_size, get_assigned_file, get_best_file,
                              get_resume_file, parse_args, parse_args_regression)


def test_parse_args_defaults_match_reference_flags():
    p = parse_args('train', [])
    assert (p.seed, p.method, p.train_n_way, p.test_n_way, p.n_shot, p.save_freq, p.start_epoch, p.stop_epoch) == \
        (0, 'DKT', 5, 5, 5, 50, 0, -1)
    assert not p.resume and not p.warmup and not p.train_aug
    t = parse_args('test', ['--repeat', '2', '--split', 'val'])
    assert (t.repeat, t.split, t.save_iter, t.adaptation) == (2, 'val', -1, False)
    assert default_image_size('Conv4', 'CUB') == 84 and default_image_size('Conv4S', 'omniglot') == 28
    assert default_image_size('ResNet10', 'miniImagenet') == 224


def test_regression_flags_and_head_pose_sampler_contract():
    """Reference io_utils.py:48-63 flags; data/qmul_loader.py:41-59 batch contract ([P,19,3,100,100], targets in [-1,1]
    on the dataset's 10-degree pitch grid, one shared trajectory per call, 24 train / 5 test people)."""
    p = parse_args_regression('train_regression', [])
    assert (p.seed, p.model, p.method, p.spectral, p.start_epoch, p.stop_epoch, p.resume) == (0, 'Conv3', 'DKT', False, 0, 100, False)
    t = parse_args_regression('test_regression', ['--spectral', '--n_support', '7'])
    assert (t.spectral, t.n_support, t.n_test_epochs) == (True, 7, 10)
    sampler = SyntheticHeadPoseSampler(seed=3)
    x, y = sampler("train")
    assert x.shape == (24, 19, 3, 100, 100) and x.dtype == torch.float32 and y.shape == (24, 19)
    xt, yt = sampler("test")
    assert xt.shape == (5, 19, 3, 100, 100) and yt.shape == (5, 19)
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    assert (y == y[:1]).all(), "every person follows the same trajectory within a batch"
    grid = (y + 1.0) * 3.0                                   # pitch 60..120 in steps of 10 -> 0, 1/3, ..., 2 -> 0..6
    assert torch.allclose(grid, grid.round(), atol=1e-5) and float(y.min()) >= -1.0 and float(y.max()) <= 1.0
    x2, y2 = SyntheticHeadPoseSampler(seed=3)("train")
    assert torch.equal(x, x2) and torch.equal(y, y2)
    # the pose is visible in the frame: frames of one person differ along the trajectory
    assert (x[0, 0] - x[0, 9]).abs().mean() > 0.01


def test_checkpoint_path_helpers(tmp_path):
    p = parse_args('train', ['--model', 'Conv4', '--train_aug'])
    d = checkpoint_dir_for(p, str(tmp_path))
    assert d.endswith('checkpoints/synthetic/Conv4_DKT_aug_5way_5shot')
    os.makedirs(d)
    assert get_resume_file(d) is None and get_best_file(d) is None
    for e in (0, 50, 7):
        torch.save({'epoch': e, 'state': {}}, get_assigned_file(d, e))
    assert get_resume_file(d).endswith('50.tar')
    assert get_best_file(d).endswith('50.tar')             # falls back to the latest epoch
    torch.save({'epoch': 3, 'state': {}}, os.path.join(d, 'best_model.tar'))
    assert get_best_file(d).endswith('best_model.tar') and get_resume_file(d).endswith('50.tar')


def test_synthetic_episode_loader_contract():
    ld = SyntheticEpisodeLoader(5, 5, 16, n_episode=3, image_size=28, n_classes=20, seed=1)
    assert len(ld) == 3
    eps = list(ld)
    assert len(eps) == 3
    x, y = eps[0]
    assert x.shape == (5, 21, 3, 28, 28) and x.dtype == torch.float32 and y.shape == (5, 21)
    assert len(set(y[:, 0].tolist())) == 5 and (y == y[:, :1]).all()       # class-major rows, distinct classes
    # same class -> same prototype: within-class distance < between-class distance
    m = x.mean(1)
    within = (x[0] - m[0]).pow(2).mean()
    between = (m[0] - m[1]).pow(2).mean() + within
    assert between > within
    # deterministic per seed, disjoint class pools per split
    x2, y2 = next(iter(SyntheticEpisodeLoader(5, 5, 16, n_episode=3, image_size=28, n_classes=20, seed=1)))
    assert torch.equal(x, x2) and torch.equal(y, y2)
    p = parse_args('train', [])
    yb = next(iter(get_episode_loader(p, 'base', 5, 1, 1, 1, 28)))[1]
    yn = next(iter(get_episode_loader(p, 'novel', 5, 1, 1, 1, 28)))[1]
    assert yb.max() < 64 <= 96 <= yn.min()


def test_ece_loss_on_known_distribution():
    import test_uncertainty as tu
    ece = tu.ECELoss(n_bins=10)
    # perfectly confident and always right -> ECE 0; perfectly confident and always wrong -> ECE 1
    logits = torch.tensor([[50.0, 0.0], [0.0, 50.0]]).repeat(10, 1)
    labels = torch.tensor([0, 1]).repeat(10)
    assert ece(logits, labels).item() < 1e-6
    assert abs(ece(logits, 1 - labels).item() - 1.0) < 1e-6
    # temperature calibration (reference test_uncertainty.py:62-74: logits / T, raw T, LBFGS lr 0.01 x 300 iterations from T = 1)
    # known answer: labels drawn from softmax(z0), logits handed over as 4 z0 (over-confident by a factor 4) -> the NLL of
    # logits / T is minimised at T = 4 up to sampling noise
    g = torch.Generator().manual_seed(0)
    z0 = torch.randn(20000, 5, generator=g) * 1.5
    y = torch.multinomial(torch.softmax(z0, 1), 1, generator=g).squeeze(1)
    z = 4.0 * z0
    t = ece.calibrate(z, y, iterations=300, lr=0.01)
    nll = torch.nn.CrossEntropyLoss()
    assert nll(z / t, y) < nll(z, y)
    # lr 0.01 x 300 LBFGS iterations do not converge from T = 1 (the reference's protocol does not either): the result must be
    # what a float64 re-run of the same protocol gives, and lie between the start and the optimum
    t64 = torch.ones(1, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.LBFGS([t64], lr=0.01, max_iter=300)

    def closure():
        opt.zero_grad()
        loss = nll(z.double() / t64, y)
        loss.backward()
        return loss
    opt.step(closure)
    assert abs(t.item() - t64.item()) < 1e-3 * t64.item() and 1.0 < t.item() <= 4.2
    # dividing by the temperature: a larger T flattens the distribution, so over-confident logits calibrate better at T = 4
    assert ece(z, y, 4.0).item() < ece(z, y, 1.0).item()


def test_cli_flags_and_checkpoint_helpers_match_reference_fixture(tmp_path):
    """tests/golden/cli_reference.json was produced by the REFERENCE's io_utils.py / configs.py (make_cli_golden.py): every
    flag of the four drivers exists here with the same default, except the two documented deviations (no datasets and only
    the DKT method exist in this build), and the checkpoint-file helpers pick the same files."""
    import json
    from dkt_amd import configs
    from dkt_amd.io_utils import model_dict
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_reference.json")))
    deviations = {"dataset": "synthetic", "method": "DKT"}
    for script, fn in (("train", parse_args), ("test", parse_args), ("train_regression", parse_args_regression),
                       ("test_regression", parse_args_regression)):
        mine = vars(fn(script, []))
        for flag, default in ref["flags"][script].items():
            assert flag in mine, (script, flag)
            assert mine[flag] == deviations.get(flag, default), (script, flag, mine[flag], default)
    assert configs.kernel_type == ref["configs"]["kernel_type"] and configs.save_dir == ref["configs"]["save_dir"]
    assert set(ref["model_dict_keys"]) - set(model_dict) <= {"ResNet50", "ResNet101"}      # bottleneck ResNets: not used by DKT configs
    d = str(tmp_path)
    h = ref["checkpoint_helpers"]
    assert get_resume_file(d) is h["empty_resume"] and get_best_file(d) is h["empty_best"]
    for e in (0, 50, 7):
        open(os.path.join(d, "%d.tar" % e), "w").close()
    assert os.path.basename(get_resume_file(d)) == h["resume"] and os.path.basename(get_best_file(d)) == h["best_without_best_model"]
    open(os.path.join(d, "best_model.tar"), "w").close()
    assert os.path.basename(get_best_file(d)) == h["best_with_best_model"]
    assert os.path.basename(get_resume_file(d)) == h["resume_with_best_model"]
    assert os.path.basename(get_assigned_file(d, 12)) == h["assigned_12"]
