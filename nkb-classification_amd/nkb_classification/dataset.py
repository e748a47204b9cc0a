"""Minimal input side for train.py: `get_dataset(data, pipeline) -> DataLoader` (signature of
/root/reference/nkb_classification/dataset.py:541-629).

The reference's five dataset types and its albumentations / cv2 augmentation stack are CPU-side input code
outside the accelerated path and are not rebuilt here.  Two sources are provided so
the config-driven train.py runs end to end:
  * type "synthetic": seeded randn images / randint labels held in memory (the benchmark's input);
  * default (no/unknown "type" with a "root"): an ImageFolder walk (root/<class>/<image>) decoded with PIL,
    resized, scaled to [0,1] and normalised — `loader.dataset.classes` is the sorted class-folder list, as
    train.py:94 expects.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

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
    def __init__(self, root, size=224, classes=None, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
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
        img = Image.open(path).convert("RGB").resize((self.size, self.size))
        arr = np.asarray(img, np.float32).transpose(2, 0, 1) / 255.0
        return torch.from_numpy((arr - self.mean) / self.std), label


def get_dataset(data, pipeline=None):
    kind = data.get("type", "folder")
    size = data.get("size", 224)
    if kind == "synthetic":
        ds = SyntheticDataset(data["n_images"], data["classes"], size=size, seed=data.get("seed", 1234))
    elif "root" in data:
        ds = FolderDataset(data["root"], size=size, classes=data.get("classes"))
    else:
        raise NotImplementedError(
            f"dataset type {kind!r}: the reference's annotation / augmentation datasets (dataset.py:183-538) are "
            "input-pipeline code outside the accelerated path and are not part of this package")
    return DataLoader(ds, batch_size=data["batch_size"], shuffle=data.get("shuffle", False),
                      num_workers=data.get("num_workers", 0), drop_last=data.get("drop_last", False), pin_memory=True)
