"""Episodic loaders.  The reference's SetDataManager (data/datamgr.py:68-84) reads JSON file lists of image
datasets through torchvision; neither the datasets nor torchvision exist here (data/ is
out of scope), so the drivers run on a SYNTHETIC episodic source with the same output contract:
an iterable of length n_episode yielding (x, y) with x: float32 [n_way, n_support + n_query, 3, H, W] on the CPU
and y: [n_way, n_support + n_query] global class ids, classes drawn by torch.randperm(n_classes)[:n_way]
(EpisodicBatchSampler, data/dataset.py:76-87)."""
from __future__ import annotations

import torch


class SyntheticEpisodeLoader:
    """Class-structured random images: every class owns a fixed low-frequency prototype; a sample is the
    prototype under a random gain plus pixel noise, so a backbone can actually learn to separate classes.
    `class_offset` shifts the class pool so base / val / novel splits are disjoint."""

    def __init__(self, n_way, n_support, n_query, n_episode=100, image_size=84, n_classes=64, class_offset=0,
                 seed=0, noise=0.6):
        self.n_way, self.per = n_way, n_support + n_query
        self.n_episode, self.hw = n_episode, image_size
        self.n_classes, self.class_offset, self.noise = n_classes, class_offset, noise
        self.gen = torch.Generator().manual_seed(1000 + seed)
        self._proto_cache = {}

    def __len__(self):
        return self.n_episode

    def _prototype(self, cls):
        p = self._proto_cache.get(cls)
        if p is None:
            g = torch.Generator().manual_seed(7919 * (cls + self.class_offset) + 13)
            low = torch.rand(3, 7, 7, generator=g)
            p = torch.nn.functional.interpolate(low[None], size=(self.hw, self.hw), mode='bilinear', align_corners=False)[0]
            self._proto_cache[cls] = p
        return p

    def __iter__(self):
        for _ in range(self.n_episode):
            classes = torch.randperm(self.n_classes, generator=self.gen)[:self.n_way]
            xs, ys = [], []
            for c in classes.tolist():
                proto = self._prototype(c)
                gain = 0.7 + 0.6 * torch.rand(self.per, 1, 1, 1, generator=self.gen)
                x = gain * proto[None] + self.noise * torch.randn(self.per, 3, self.hw, self.hw, generator=self.gen)
                xs.append(x.clamp(0.0, 1.5))
                ys.append(torch.full((self.per,), c + self.class_offset, dtype=torch.long))
            yield torch.stack(xs), torch.stack(ys)


def get_episode_loader(params, split, n_way, n_support, n_query, n_episode, image_size, seed=0):
    """split in {'base', 'val', 'novel'}.  Real datasets need the reference's filelists + torchvision."""
    if params.dataset != 'synthetic':
        raise NotImplementedError(
            "dataset '%s': the image datasets of the reference (filelists/, torchvision transforms) are not part of "
            "this build; use --dataset synthetic" % params.dataset)
    offsets = {'base': 0, 'val': 64, 'novel': 96}
    sizes = {'base': 64, 'val': 32, 'novel': 40}
    return SyntheticEpisodeLoader(n_way, n_support, n_query, n_episode, image_size, n_classes=max(sizes[split], n_way),
                                  class_offset=offsets[split], seed=seed + {'base': 0, 'val': 1, 'novel': 2}[split])


class SyntheticHeadPoseSampler:
    """Stand-in for the QMUL head-pose sampler (reference data/qmul_loader.py:41-59 `get_batch`): a call returns
    (inputs [P, 19, 3, 100, 100], targets [P, 19]) -- one trajectory of 19 frames per person, the target being the
    normalised pitch in [-1, 1] that follows amp * sin(phase + t), amp ~ U(-3, 3), phase ~ U(-5, 5), quantised to the
    dataset's 10-degree grid exactly as the reference maps the curve to image files.  The frame of (person, pitch, angle)
    is that person's fixed low-frequency "face" shifted vertically with the pitch and horizontally with the yaw angle,
    plus pixel noise, so a backbone can learn the pose.  24 train / 5 test people like the dataset's split."""

    def __init__(self, seed=0, image_size=100, num_samples=19, n_train=24, n_test=5, noise=0.05):
        self.hw, self.num_samples, self.noise = image_size, num_samples, noise
        self.people = {"train": list(range(n_train)), "test": list(range(n_train, n_train + n_test))}
        self.rng = __import__("numpy").random.RandomState(1234 + seed)
        self.gen = torch.Generator().manual_seed(4321 + seed)
        self._faces = {}

    def _face(self, person):
        f = self._faces.get(person)
        if f is None:
            g = torch.Generator().manual_seed(104729 * (person + 1))
            low = torch.rand(3, 6, 6, generator=g)
            f = torch.nn.functional.interpolate(low[None], size=(self.hw + 40, self.hw + 40), mode='bilinear', align_corners=False)[0]
            self._faces[person] = f
        return f

    def __call__(self, split="train"):
        np = __import__("numpy")
        amp, phase = self.rng.uniform(-3, 3), self.rng.uniform(-5, 5)
        wave = [amp * np.sin(phase + x) for x in range(self.num_samples)]
        angles = [10 * x for x in range(self.num_samples)]
        pitches = [int(round((y + 3) * 10 + 60, -1)) for y in wave]
        inputs, targets = [], []
        for person in self.people[split]:
            face = self._face(person)
            frames = []
            for pitch, angle in zip(pitches, angles):
                dy = 20 + int(round((pitch - 90) / 30.0 * 18))          # pitch 60..120 -> vertical shift
                dx = 20 + int(round((angle - 90) / 90.0 * 18))          # yaw 0..180 -> horizontal shift
                frames.append(face[:, dy:dy + self.hw, dx:dx + self.hw])
            x = torch.stack(frames)
            x = (x + self.noise * torch.randn(x.shape, generator=self.gen)).clamp(0.0, 1.0)
            inputs.append(x)
            targets.append(torch.tensor([2.0 * ((p - 60) / 60.0) - 1.0 for p in pitches], dtype=torch.float32))
        return torch.stack(inputs), torch.stack(targets)
