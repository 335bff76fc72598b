#!/usr/bin/env python3
"""Command-line front end over the package, taking the flags of the reference's two entry points:

    python scripts/wavedm_run.py eval  --config raindrop_wavelet.yml --resume ckpt.pth.tar --test_set raindrop --sampling_timesteps 25
    python scripts/wavedm_run.py train --config raindrop_wavelet.yml [--resume ckpt.pth.tar]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/wavedm_run.py eval ...      # one rank per GPU

`eval` = eval_diffusion.py (DiffusiveRestoration.restore over the validation loader), `train` = train_diffusion.py (diffusion.train).
--config is a file name under ./configs or a path.  Under torchrun every rank restores its share of the validation images (the
loaders use a DistributedSampler) and rank 0 prints the PSNR over all of them; training all-reduces gradients over RCCL.
Extras: --dtype {f16,bf16,f32x3,f32} (default: f16 when the checkpoint fits fp16, else bf16 with a warning), --images_per_call N (eval: images per sampler call, default automatic), --full_length (eval: no early stop), --hfrm_ckpt PATH, --max_steps N (train)."""
import argparse
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wavedm_amd                                       # noqa: E402
from wavedm_amd import datasets                         # noqa: E402
from wavedm_amd.config import load_config               # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("mode", choices=["eval", "train"])
    ap.add_argument("--config", required=True, help="YAML file (name under ./configs, or a path)")
    ap.add_argument("--resume", default="", help="diffusion checkpoint (*.pth.tar) to evaluate / to resume from")
    ap.add_argument("--grid_r", type=int, default=16, help="stride of the overlapping patch grid (wavelet-domain pixels)")
    ap.add_argument("--sampling_timesteps", type=int, default=25, help="DDIM steps")
    ap.add_argument("--test_set", default="raindrop")
    ap.add_argument("--image_folder", default="results/images", help="where restored images / validation sheets are written")
    ap.add_argument("--seed", type=int, default=61)
    ap.add_argument("--ema", action="store_true", help="eval: load the EMA weights of the checkpoint")
    ap.add_argument("--dtype", default=None, choices=["f16", "bf16", "f32x3", "f32"])
    ap.add_argument("--images_per_call", type=int, default=0, help="eval: images per sampler call; 0 = automatic (as many same-sized images as fill the UNet calls), 1 = the reference's loop")
    ap.add_argument("--full_length", action="store_true", help="eval: also run the four DDIM steps behind x0_preds[-5], which restore() never reads (the reference's step count)")
    ap.add_argument("--hfrm_ckpt", default=None)
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--no_save", action="store_true", help="eval: metrics only, no PNGs")
    a = ap.parse_args(argv)
    a.early_stop = not a.full_length
    a.images_per_call = a.images_per_call or None          # None: DiffusiveRestoration's automatic grouping
    a.rank = int(os.environ.get("RANK", 0))
    a.world_size = int(os.environ.get("WORLD_SIZE", 1))
    a.local_rank = int(os.environ.get("LOCAL_RANK", 0))
    path = a.config if os.path.isfile(a.config) else os.path.join("configs", a.config)
    return a, load_config(path)


def main(argv=None):
    args, config = parse(argv)
    if not torch.cuda.is_available():
        raise SystemExit("wavedm_run: no GPU visible (the package has no CPU path)")
    torch.cuda.set_device(args.local_rank)
    config.device = torch.device("cuda", args.local_rank)
    def reseed(seed):
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
    reseed(args.seed)            # every rank builds the SAME initial model (train_diffusion.py:74-77 seeds all ranks identically)
    if args.world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")          # RCCL
    if not getattr(config.data, "wavelet", False):
        raise SystemExit("wavedm_run: only the wavelet-domain model (data.wavelet: True) is built")
    print(f"=> dataset {config.data.dataset}, rank {args.rank} of {args.world_size} on {config.device}")
    DATASET = datasets.__dict__[config.data.dataset](args, config)
    diffusion = wavedm_amd.DenoisingDiffusion_Wavelet(args, config, dtype=args.dtype)
    if args.mode == "train":
        # only now do the ranks diverge: noise, timesteps and crop positions differ per rank (the trainer also broadcasts rank 0's
        # parameters when it is built, as DistributedDataParallel does at construction, ddm_wavelet.py:168)
        reseed(args.seed + args.rank)
        diffusion.train(DATASET, max_steps=args.max_steps)
        if args.world_size > 1:
            dist.destroy_process_group()
        return 0
    if args.ema and args.resume:
        diffusion.load_ddm_ckpt(args.resume, ema=True)
    _, val_loader = DATASET.get_loaders(parse_patches=False, validation=args.test_set)
    restorer = wavedm_amd.DiffusiveRestoration(diffusion, args, config, save_images=not args.no_save)
    _, psnrs = restorer.restore(val_loader, validation=args.test_set, r=args.grid_r)
    if args.world_size > 1:                               # PSNR over every rank's images
        mine = torch.tensor([float(np.sum(psnrs)), float(len(psnrs))], dtype=torch.float64, device=config.device)
        dist.all_reduce(mine)
        if args.rank == 0:
            print(f"psnr all ranks: {float(mine[0] / max(mine[1], 1.0)):.4f} over {int(mine[1])} images")
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
