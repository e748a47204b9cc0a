"""Minimal input side for train.py: `get_dataset(data, pipeline) -> DataLoader` (signature of
/root/reference/nkb_classification/dataset.py:541-629).

The reference's five annotation-file dataset types and the geometric / colour augmentations of its albumentations
stack stay CPU-side input code outside the accelerated path and are not rebuilt.  What IS here:
  * `DeviceLoader`: the loader hands over pinned **uint8** HWC batches; the H2D copy runs on a copy stream one batch
    ahead of the train step, and the pad -> flip -> normalise -> CHW tail of the pipeline (configs/singletask_config.py:
    162-219) runs on the GPU (`nkb_image_prep`), so engine.py:40's `img.to(device)` finds the batch already resident;
  * `ImbalancedDatasetSampler` (dataset.py:27-86: weights 1 / count[label], `torch.multinomial` with replacement) and
    `ShardedSampler` (one process per GPU: rank r takes indices r::W of a seed-shared permutation);
  * `get_dataset` honours `shuffle`, `weighted_sampling`, `drop_last`, `num_workers` (dataset.py:606-628).
Two sources are provided so the config-driven train.py runs end to end:
  * type "synthetic": seeded randn images / randint labels held in memory (the benchmark's input);
  * default (no/unknown "type" with a "root"): an ImageFolder walk (root/<class>/<image>) decoded with PIL,
    resized, scaled to [0,1] and normalised — `loader.dataset.classes` is the sorted class-folder list, as
    train.py:94 expects.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Sampler

_EXT = {".png", ".jpg", ".jpeg", ".bmp", ".webp"}


class SyntheticDataset(Dataset):
    def __init__(self, n_images, classes, size=224, seed=1234):
        g = torch.Generator().manual_seed(seed)
        self.classes = classes
        self.images = torch.randn(n_images, 3, size, size, generator=g)
        if isinstance(classes, dict):
            self.labels = {t: torch.randint(0, len(c), (n_images,), generator=g) for t, c in classes.items()}
        else:
            self.labels = torch.randint(0, len(classes), (n_images,), generator=g)

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        if isinstance(self.labels, dict):
            return self.images[i], {t: int(v[i]) for t, v in self.labels.items()}
        return self.images[i], int(self.labels[i])


class FolderDataset(Dataset):
    """raw=True: items are (uint8 HWC image resized so that its longest side is `size` and placed top-left in a
    size x size canvas, (h, w), label) for DeviceLoader; raw=False: normalised float CHW, squashed to size x size."""

    def __init__(self, root, size=224, classes=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), raw=False):
        self.raw = raw
        root = Path(root)
        self.classes = classes or sorted(d.name for d in root.iterdir() if d.is_dir())
        self.items = [(p, i) for i, c in enumerate(self.classes) for p in sorted((root / c).iterdir())
                      if p.suffix.lower() in _EXT]
        self.size = size
        self.mean = np.asarray(mean, np.float32).reshape(3, 1, 1)
        self.std = np.asarray(std, np.float32).reshape(3, 1, 1)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.items[i]
        img = Image.open(path).convert("RGB")
        if self.raw:
            # A.LongestMaxSize(size): scale so that max(h, w) == size (host side; decoding is here anyway)
            k = self.size / max(img.size)
            w, h = max(1, round(img.size[0] * k)), max(1, round(img.size[1] * k))
            canvas = np.zeros((self.size, self.size, 3), np.uint8)
            canvas[:h, :w] = np.asarray(img.resize((w, h)), np.uint8)
            return torch.from_numpy(canvas), torch.tensor([h, w], dtype=torch.int32), label
        arr = np.asarray(img.resize((self.size, self.size)), np.float32).transpose(2, 0, 1) / 255.0
        return torch.from_numpy((arr - self.mean) / self.std), label

    def get_labels(self):
        return [label for _, label in self.items]


class ImbalancedDatasetSampler(Sampler):
    """dataset.py:27-86: every index is drawn with probability proportional to 1 / (number of samples of its label),
    `num_samples` draws with replacement per epoch.  With world > 1 each rank draws its own `num_samples / world`
    from a generator seeded `seed + rank`."""

    def __init__(self, dataset, labels=None, indices=None, num_samples=None, seed=None, rank=0, world=1):
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        labels = dataset.get_labels() if labels is None else labels
        labels = [int(labels[i]) for i in self.indices]
        counts = {}
        for v in labels:
            counts[v] = counts.get(v, 0) + 1
        self.weights = torch.DoubleTensor([1.0 / counts[v] for v in labels])
        total = len(self.indices) if num_samples is None else num_samples
        self.num_samples = total if world == 1 else (total + world - 1) // world
        self.generator = None
        if seed is not None:
            self.generator = torch.Generator().manual_seed(int(seed) + rank)

    def __iter__(self):
        draws = torch.multinomial(self.weights, self.num_samples, replacement=True, generator=self.generator)
        return (self.indices[i] for i in draws)

    def __len__(self):
        return self.num_samples


class ShardedSampler(Sampler):
    """Data-parallel split: the same seeded permutation on every rank (re-drawn per epoch through set_epoch), rank r taking
    positions r::world.  pad=True (training: every rank must run the same number of steps, the gradient exchange is a
    collective) repeats the first indices up to a multiple of the world size — `real_len` tells how many of this rank's samples
    are NOT such repeats, so that gathered epoch results can drop them; pad=False (validation: no collective runs inside
    val_epoch) leaves the shards uneven and every sample is seen exactly once."""

    def __init__(self, n, rank, world, shuffle=True, seed=0, pad=True):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n, rank, world, shuffle, seed, 0
        self.pad = pad
        self.real_len = max(0, (n - rank + world - 1) // world)
        self.num_samples = (n + world - 1) // world if pad else self.real_len

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        if self.pad:
            order += order[: self.num_samples * self.world - self.n]
        return iter(order[self.rank::self.world])

    def __len__(self):
        return self.num_samples


class DeviceLoader:
    """Wraps a host loader that yields (uint8 [B,H,W,3], sizes int32 [B,2] or None, target) and yields
    (float32 [B,3,size,size] on `device`, target) — the batch format engine.train_epoch consumes (engine.py:39-41).

    Batch k+1 is pinned and copied on a dedicated copy stream while batch k trains; the prep kernel runs on the compute
    stream after an event wait, so no host synchronisation is added.  Flip draws come from a host generator (one
    Bernoulli per image and axis, A.HorizontalFlip / A.VerticalFlip(p))."""

    def __init__(self, loader, device, size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), hflip_p=0.0,
                 vflip_p=0.0, fill=0.0, seed=0):
        self.loader, self.device = loader, torch.device(device)
        self.size, self.mean, self.std, self.fill = size, tuple(mean), tuple(std), float(fill)
        self.hflip_p, self.vflip_p = float(hflip_p), float(vflip_p)
        self.gen = torch.Generator().manual_seed(seed)
        self.dataset = getattr(loader, "dataset", None)
        self._copy = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, item):
        """Pinned host batch -> device uint8 on the copy stream; returns what the compute stream needs."""
        if len(item) == 3:
            raw, sizes, target = item
        else:
            (raw, target), sizes = item, None
        if raw.dtype != torch.uint8 or raw.dim() != 4 or raw.shape[-1] != 3:
            raise RuntimeError(f"DeviceLoader expects uint8 [B,H,W,3] batches, got {tuple(raw.shape)} {raw.dtype}")
        B = raw.shape[0]
        flags = None
        if self.hflip_p > 0 or self.vflip_p > 0:
            u = torch.rand(B, 2, generator=self.gen)
            flags = ((u[:, 0] < self.hflip_p).to(torch.uint8) | ((u[:, 1] < self.vflip_p).to(torch.uint8) << 1))
        if self._copy is None:
            self._copy = torch.cuda.Stream(device=self.device)
        pin = lambda t: t if t is None or t.is_pinned() else t.contiguous().pin_memory()       # noqa: E731
        raw, sizes, flags = pin(raw), pin(sizes), pin(flags)
        with torch.cuda.stream(self._copy):
            d_raw = raw.to(self.device, non_blocking=True)
            d_sizes = sizes.to(self.device, non_blocking=True) if sizes is not None else None
            d_flags = flags.to(self.device, non_blocking=True) if flags is not None else None
            if isinstance(target, dict):
                d_target = {k: torch.as_tensor(v).to(self.device, non_blocking=True) for k, v in target.items()}
            else:
                d_target = torch.as_tensor(target).to(self.device, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self._copy)
        return dict(raw=d_raw, sizes=d_sizes, flags=d_flags, target=d_target, ready=ready, host=(raw, sizes, flags))

    def _finish(self, st):
        from . import hip
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(st["ready"])
        raw = st["raw"]
        B, Hs, Ws, _ = raw.shape
        out = torch.empty(B, 3, self.size, self.size, device=self.device, dtype=torch.float32)
        hip.image_prep(raw, st["sizes"], st["flags"], out, B, Hs, Ws, self.size, self.size, self.mean, self.std, self.fill)
        tgt = st["target"]
        for t in (raw, st["sizes"], st["flags"], *(tgt.values() if isinstance(tgt, dict) else (tgt,))):
            if t is not None:
                # allocated on the copy stream, consumed on the compute stream: without this the caching allocator hands the
                # block to the NEXT staged batch as soon as Python drops the tensor — while the loss kernel that reads it is still
                # queued (labels overwritten by another batch's flip flags: out-of-range classes, a NaN loss; bench.py --input
                # host-uint8 ran into it, a loop that synchronises every step does not)
                t.record_stream(cur)
        return out, tgt

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = self._stage(next(it))      # copy of batch k+1 overlaps the step on batch k
            except StopIteration:
                nxt = None
            yield self._finish(cur)


def get_dataset(data, pipeline=None):
    """dataset.py:541-629.  Extra keys: `device_pipeline=True` (+ `device`, `hflip_p`, `vflip_p`, `mean`, `std`) wraps the
    folder loader in a DeviceLoader; `rank` / `world` (+ `shard_seed`, the seed of the permutation all ranks share) shard the data for one-process-per-GPU
    training."""
    kind = data.get("type", "folder")
    size = data.get("size", 224)
    on_device = bool(data.get("device_pipeline", False))
    if kind == "synthetic":
        ds = SyntheticDataset(data["n_images"], data["classes"], size=size, seed=data.get("seed", 1234))
        on_device = False
    elif "root" in data:
        ds = FolderDataset(data["root"], size=size, classes=data.get("classes"), raw=on_device)
    else:
        raise NotImplementedError(
            f"dataset type {kind!r}: the reference's annotation / augmentation datasets (dataset.py:183-538) are "
            "input-pipeline code outside the accelerated path and are not part of this package")
    rank, world = int(data.get("rank", 0)), int(data.get("world", 1))
    sampler, shuffle = None, data.get("shuffle", False)
    if data.get("weighted_sampling", False):              # dataset.py:607-617
        sampler, shuffle = ImbalancedDatasetSampler(ds, seed=data.get("shard_seed", 0) if world > 1 else None, rank=rank,
                                                    world=world), False
    elif world > 1:
        sampler, shuffle = ShardedSampler(len(ds), rank, world, shuffle=shuffle, seed=data.get("shard_seed", 0),
                                          pad=bool(data.get("shard_pad", True))), False
    loader = DataLoader(ds, batch_size=data["batch_size"], shuffle=shuffle, sampler=sampler,
                        num_workers=data.get("num_workers", 0), drop_last=data.get("drop_last", False), pin_memory=True)
    if on_device:
        return DeviceLoader(loader, data.get("device", "cuda:0"), size, mean=data.get("mean", (0.485, 0.456, 0.406)),
                            std=data.get("std", (0.229, 0.224, 0.225)), hflip_p=data.get("hflip_p", 0.0),
                            vflip_p=data.get("vflip_p", 0.0), seed=data.get("seed", 0))
    return loader
