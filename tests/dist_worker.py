"""Worker for the 2-rank GPU tests (tests/test_gpu_dist.py): launched by torch.distributed.run with the gloo backend so that
both ranks can share the single GPU of the test box.  Exercises the N > 1 code paths of wavedm_amd.parallel / sampling on
the real HIP kernels: image-sharded restore + all-gather, weight broadcast + adopt, and the patch-sharded sampler."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main(out_path):
    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import parallel, procedural as P
    torch.set_grad_enabled(False)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = P.reduced_config()
    cfg.device = dev
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    if rank == 0:                                   # only rank 0 "has the checkpoint"
        d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    parallel.broadcast_weights(d.model, src=0)

    # (1) image-sharded batch of independent crops, all-gather of the outputs
    rainy, x_T = P.synthetic_batch(5, patch_px=64, seed=7)          # 5 images over 2 ranks: 3 + 2
    rainy, x_T = rainy.to(dev), x_T.to(dev)
    out_img = parallel.restore_sharded(lambda r, n: d.restore_batch(r, n)[0], (rainy, x_T), 5)

    # (2) patch-sharded single stitched image: one all-reduce per DDIM step
    g = torch.Generator().manual_seed(31)
    img = torch.rand(1, 3, 96, 112, generator=g).to(dev)
    noise = torch.randn(1, 3, 24, 28, generator=g).to(dev) + (0.0 if rank == 0 else 1.0)      # rank 1's noise must be replaced by rank 0's
    x_cond = d.wavelet_dec((2 * img - 1).contiguous())
    d.patch_group = True
    corners = [(i, j) for i in (0, 4, 8) for j in (0, 4, 8, 12)]
    xs, x0 = d.sample_image(x_cond, noise, x_other=x_cond[:, 3:].contiguous(), last=False, patch_locs=corners, patch_size=16, use_other=True)
    d.patch_group = None
    # (3) data-parallel training step: each rank back-propagates its half of the batch, one all-reduce averages the gradients
    from wavedm_amd.training import Trainer
    from gpu_util import seeded
    tr = Trainer(cfg, dtype="f32")
    tr.load_state_dict(P.procedural_state_dict(cfg, seed=61))
    bx, be, bt = seeded((4, 96, 16, 16), 401), seeded((4, 3, 16, 16), 402), torch.tensor([990, 9, 500, 499])
    lo, hi = parallel.shard_range(4, rank, world)
    tr.loss_and_grads(bx[lo:hi].to(dev), bt[lo:hi], be[lo:hi].to(dev))
    tr.allreduce_grads()
    tr.optimizer_step()
    if rank == 0:
        torch.save({"out_img": out_img.cpu(), "xs_last": xs[-1].cpu(), "x0_m5": x0[-5].cpu(), "world": world,
                    "grads": tr.grads.cpu(), "params1": tr.params.cpu()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
