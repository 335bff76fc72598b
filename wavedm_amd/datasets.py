"""RainDrop data loading -- the reference's `datasets/raindrop.py:14-150` contract without torchvision.

Input side of `DiffusiveRestoration.restore`.  Same directory layout (`<data_dir>/raindrop/train`,
`<data_dir>/raindrop/raindrop_test`, each with `input/` and `gt/`, ground truth = input name with "rain" -> "clean"),
same PIL calls (LANCZOS resize to 720x480, then to multiples of 16), same return tuple
`(cat[input, gt] (6,H,W) in [0,1], img_id, total_image)`, same `DistributedSampler(num_replicas=args.world_size,
rank=args.rank)` -- so `eval_diffusion.py:88-90` runs unchanged.  This is host-side I/O: PIL decodes and resizes, the
tensors are pinned by the DataLoader, everything after the H2D copy happens in restoration.py on the GPU."""
from __future__ import annotations

import os
import random
import re

import numpy as np
import torch
import torch.utils.data
from torch.utils.data.distributed import DistributedSampler


def to_tensor(pic) -> torch.Tensor:
    """torchvision.transforms.ToTensor for PIL images: HWC uint8 -> CHW float32 in [0,1]."""
    a = np.asarray(pic)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


def eval_size(w: int, h: int):
    """raindrop.py:130-138 after the fixed 720x480 resize: cap the long side at 1024, round both up to multiples of 16."""
    if h > w and h > 1024:
        w, h = int(np.ceil(w * 1024 / h)), 1024
    elif h <= w and w > 1024:
        h, w = int(np.ceil(h * 1024 / w)), 1024
    return int(16 * np.ceil(w / 16.0)), int(16 * np.ceil(h / 16.0))


def worker_kwargs(config, num_workers):
    """DataLoader keywords for the worker processes.  NOT the reference's default (`fork`): every fork() of a process that holds a HIP context stalls its GPU queue once,
    for ~0.2 s + ~21 ms per MB of PINNED host memory the process holds (scripts/fork_probe.py, profiles/r06_fork_probe.log: 1.8 s with 64 MB pinned, 10.9 s with 512 MB; device
    memory and ordinary heap do not count) -- hipHostMalloc pages go copy-on-write at fork and the queue waits until the driver has them back.  restore() holds staging
    buffers, PNG buffers and the loader's pinned batches: ~2 s per iter(loader) (profiles/r06_restore_fork_interference.log, section E: GPU-side timestamps), a quarter of a
    58-image evaluation, every pass of a benchmark.  Workers from a fork server never fork the GPU process.  (The pinned buffers this package allocates are also kept out of
    fork() -- _lib.pinned_dontfork -- which removes their share of the stall for loaders that do fork.)  So the workers come from a fork server (a clean process that never touched the GPU; torch
    pre-imported there once) and stay alive between epochs / passes, which also takes their start-up (~1 s) out of every pass but the first.
    `config.data.worker_context: fork` restores the reference's behaviour (restore() then reads a short validation set to its end before it launches much)."""
    ctx = getattr(getattr(config, "data", None), "worker_context", "forkserver")
    if num_workers <= 0 or ctx in (None, "", "fork"):
        return {}
    import multiprocessing as mp
    try:
        mp.get_context(ctx).set_forkserver_preload(["torch", "numpy", "PIL.Image", "wavedm_amd.datasets"]) if ctx == "forkserver" else None
    except Exception:                                   # (a server that is already running keeps its preload list)
        pass
    return dict(multiprocessing_context=ctx, persistent_workers=True)


class RainDropDataset(torch.utils.data.Dataset):
    def __init__(self, dir, patch_size, n, transforms=None, filelist=None, parse_patches=True):
        super().__init__()
        if filelist is None:
            inputs = os.path.join(dir, "input")
            names = [f for f in os.listdir(inputs) if os.path.isfile(os.path.join(inputs, f))]
            input_names = [os.path.join(inputs, f) for f in names]
            gt_names = [os.path.join(dir, "gt", f.replace("rain", "clean")) for f in names]
            print(len(input_names))
            order = list(enumerate(input_names))
            random.shuffle(order)                                  # raindrop.py:69-72: the global `random` stream
            idx, input_names = zip(*order) if order else ((), ())
            gt_names = [gt_names[k] for k in idx]
            self.dir = None
        else:
            self.dir = dir
            with open(os.path.join(dir, filelist)) as f:
                input_names = [line.strip() for line in f.readlines()]
            gt_names = [name.replace("input", "gt") for name in input_names]
        self.input_names, self.gt_names = list(input_names), list(gt_names)
        self.patch_size, self.n, self.parse_patches = patch_size, n, parse_patches
        self.transforms = transforms or to_tensor

    @staticmethod
    def get_params(img, output_size, n):
        w, h = img.size
        th, tw = output_size
        if w == tw and h == th:
            return 0, 0, h, w
        rows = [random.randint(0, h - th) for _ in range(n)]
        cols = [random.randint(0, w - tw) for _ in range(n)]
        return rows, cols, th, tw

    @staticmethod
    def n_random_crops(img, x, y, h, w):
        return tuple(img.crop((y[k], x[k], y[k] + w, x[k] + h)) for k in range(len(x)))

    def _open(self, name):
        from PIL import Image
        return Image.open(os.path.join(self.dir, name) if self.dir else name)

    def get_images(self, index):
        from PIL import Image
        input_name, gt_name = self.input_names[index], self.gt_names[index]
        img_id = re.split("/", input_name)[-1][:-4]
        inp = self._open(input_name)
        try:
            gt = self._open(gt_name)
        except Exception:
            gt = self._open(gt_name).convert("RGB")
        T = self.transforms
        if self.parse_patches:
            rows, cols, h, w = self.get_params(inp, (self.patch_size, self.patch_size), self.n)
            total = T(inp.resize((720, 480), Image.LANCZOS)).repeat(self.n, 1, 1, 1)
            ic, gc = self.n_random_crops(inp, rows, cols, h, w), self.n_random_crops(gt, rows, cols, h, w)
            return torch.stack([torch.cat([T(a), T(b)], dim=0) for a, b in zip(ic, gc)], dim=0), img_id, total
        inp = inp.resize((720, 480), Image.LANCZOS)
        wd, ht = eval_size(*inp.size)
        inp = inp.resize((wd, ht), Image.LANCZOS)
        gt = gt.resize((wd, ht), Image.LANCZOS)
        ti = T(inp)
        return torch.cat([ti, T(gt)], dim=0), img_id, ti

    def __getitem__(self, index):
        return self.get_images(index)

    def __len__(self):
        return len(self.input_names)


class RainDrop:
    def __init__(self, args, config):
        self.args, self.config = args, config
        self.transforms = to_tensor

    def get_loaders(self, parse_patches=True, validation="raindrop"):
        print("=> evaluating raindrop test set...")
        cfg = self.config
        root = os.path.join(cfg.data.data_dir, "raindrop")
        mk = lambda sub: RainDropDataset(dir=os.path.join(root, sub), n=cfg.training.patch_n, patch_size=cfg.data.patch_size,
                                         transforms=self.transforms, filelist=None, parse_patches=parse_patches)
        train_dataset, val_dataset = mk("train"), mk("raindrop_test")
        if not parse_patches:
            cfg.sampling.batch_size = 1
        ws, rank = getattr(self.args, "world_size", 1), getattr(self.args, "rank", 0)
        wk = worker_kwargs(cfg, cfg.data.num_workers)
        train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=cfg.training.batch_size,
                                                   sampler=DistributedSampler(train_dataset, num_replicas=ws, rank=rank),
                                                   num_workers=cfg.data.num_workers, pin_memory=True, **wk)
        val_loader = torch.utils.data.DataLoader(val_dataset, batch_size=cfg.sampling.batch_size, shuffle=False,
                                                 sampler=DistributedSampler(val_dataset, num_replicas=ws, rank=rank),
                                                 num_workers=cfg.data.num_workers, pin_memory=True, **wk)
        return train_loader, val_loader
