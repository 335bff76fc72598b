"""Worker for the 2-rank GPU tests, launched by torch.distributed.run.  WDM_TEST_BACKEND=gloo (tests/test_gpu_dist.py): both ranks share
cuda:0, so the N > 1 code paths run on the real HIP kernels on a one-GPU box.  WDM_TEST_BACKEND=nccl (tests/test_gpu_rccl.py, needs two
devices): one rank per GPU over RCCL -- the deployment shape (eval_diffusion.py:83, ddm_wavelet.py:168 in the reference).
Exercised: weight broadcast + adopt, image-sharded restore + all-gather, the patch-sharded sampler, the gradient all-reduce."""
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
    backend = os.environ.get("WDM_TEST_BACKEND", "gloo")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm
    else:
        dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if os.environ.get("WDM_TEST_MODE") == "patch8":
        # one 480x720 image, its 45 patches sharded over the ranks inside DiffusiveRestoration.restore() (tests/test_gpu_dist.py)
        from test_gpu_dist import patch8_setup
        d, args, img = patch8_setup(dev)
        d.patch_group = True
        rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
        torch.manual_seed(77 + 1000 * rank)                         # ranks draw DIFFERENT start noise: the sampler must spread rank 0's
        outs, psnr = rest.restore([(img, ("one",), torch.zeros(1))], validation="raindrop", r=16)
        # ... and two images with the AUTOMATIC grouping (args.images_per_call unset): every rank must form the same groups whatever its loader's timing
        args2 = SimpleNamespace(**{k: v for k, v in vars(args).items() if k != "images_per_call"})
        rest2 = wavedm_amd.DiffusiveRestoration(d, args2, d.config, save_images=False)
        torch.manual_seed(78 + 1000 * rank)
        outs2, _ = rest2.restore([(img, ("a",), torch.zeros(1)), (img.flip(-1), ("b",), torch.zeros(1))], validation="raindrop", r=16)
        shards = [parallel.shard_range(45, r_, world)[1] - parallel.shard_range(45, r_, world)[0] for r_ in range(world)]
        if rank == 0:
            torch.save({"out": outs[0].cpu(), "psnr": psnr[0], "world": world, "shards": shards, "out2": [o.cpu() for o in outs2]}, out_path)
        dist.barrier()
        dist.destroy_process_group()
        return
    cfg = P.reduced_config()
    cfg.device = dev
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=dev.index, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
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
    # ... the same step with the all-reduce in buckets behind events recorded DURING the backward: the same averaged gradients, bit for bit on two ranks
    trb = Trainer(cfg, dtype="f32")
    trb.load_state_dict(P.procedural_state_dict(cfg, seed=61))
    trb.enable_grad_buckets(5)
    trb.loss_and_grads(bx[lo:hi].to(dev), bt[lo:hi], be[lo:hi].to(dev))
    trb.allreduce_grads_overlapped()
    torch.cuda.synchronize()
    bk = trb.grad_buckets()
    buckets_ok = len(bk) >= 2 and torch.equal(trb.grads, tr.grads) and all(a[0] == b[1] for a, b in zip(bk[:-1], bk[1:])) and all(lo_ < hi_ for lo_, hi_ in bk)
    tr.optimizer_step()
    # (4) from-scratch training, no --resume: the ranks build DIFFERENT random models (the ADVICE r1 scenario); make_trainer() must leave
    # every rank with rank 0's parameters, as DistributedDataParallel's construction-time broadcast does (ddm_wavelet.py:168)
    torch.manual_seed(1000 + rank)
    d4 = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    t4 = d4.make_trainer(dtype="f32")
    own = torch.cat([p.detach().flatten() for p in d4.model.parameters()]).to(dev)       # this rank's own random initialisation
    owns = [torch.empty_like(own) for _ in range(world)]
    dist.all_gather(owns, own)
    mine = t4.params.clone()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = (not torch.equal(owns[0], owns[1])) and all(torch.equal(b, both[0]) for b in both) and torch.equal(t4.ema, t4.params)
    # ... and one real iteration of the loop body (DenoisingDiffusion_Wavelet.train_step: DWT, q-sample, loss, backward, gradient all-reduce,
    # Adam, EMA) on DIFFERENT crops and noise per rank must leave the ranks with identical parameters again
    torch.manual_seed(2000 + rank)
    crops = torch.rand(2, 6, 64, 64, generator=torch.Generator().manual_seed(50 + rank))
    d4.train_step(crops)
    after = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(after, t4.params.clone())
    ok = ok and all(torch.equal(b, after[0]) for b in after) and not torch.equal(after[0], both[0]) and t4.step == 1
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        torch.save({"out_img": out_img.cpu(), "fresh_init_synced": bool(flag.item() == 1.0), "xs_last": xs[-1].cpu(), "x0_m5": x0[-5].cpu(), "world": world,
                    "grads": tr.grads.cpu(), "params1": tr.params.cpu(), "buckets_ok": bool(buckets_ok), "buckets": bk}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
