"""GPU, world size 2 on ONE device (gloo backend, both ranks on cuda:0): the multi-GPU code paths on the real kernels.
RCCL itself needs several GPUs; the driver's scaling run covers that.  Here: image-sharded restore == single-process restore
bit for bit, weights arrive by broadcast, and the patch-sharded sampler (one all-reduce per step) == the unsharded one to 1e-5."""
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_linf
from wavedm_amd import procedural as P

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
HERE = os.path.dirname(os.path.abspath(__file__))


def run_two_ranks(tmp_path, backend):
    """Launch tests/dist_worker.py on 2 ranks with `backend` and compare what rank 0 saved with the single-process results."""
    import wavedm_amd
    out = tmp_path / "dist.pt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", WDM_TEST_BACKEND=backend)
    import socket
    with socket.socket() as sock:                      # a free rendezvous port on this box
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    assert got["world"] == 2
    assert got["buckets_ok"], got["buckets"]                      # bucketed, backward-overlapped all-reduce == the flat one
    assert got["fresh_init_synced"]                                 # from-scratch training starts from rank 0's weights on every rank
    # single-process references
    dev = torch.device("cuda", 0)
    cfg = P.reduced_config()
    cfg.device = dev
    args = SimpleNamespace(resume="", sampling_timesteps=5, local_rank=0, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=4)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    rainy, x_T = P.synthetic_batch(5, patch_px=64, seed=7)
    want = d.restore_batch(rainy.to(dev), x_T.to(dev))[0].cpu()
    assert torch.equal(got["out_img"], want)                      # image sharding changes nothing, bit for bit
    g = torch.Generator().manual_seed(31)
    img = torch.rand(1, 3, 96, 112, generator=g).to(dev)
    noise = torch.randn(1, 3, 24, 28, generator=g).to(dev)
    x_cond = d.wavelet_dec((2 * img - 1).contiguous())
    corners = [(i, j) for i in (0, 4, 8) for j in (0, 4, 8, 12)]
    xs, x0 = d.sample_image(x_cond, noise, x_other=x_cond[:, 3:].contiguous(), last=False, patch_locs=corners, patch_size=16, use_other=True)
    assert rel_linf(got["xs_last"], xs[-1].cpu()) <= 1e-5         # only the association of the overlap sums differs
    assert rel_linf(got["x0_m5"], x0[-5].cpu()) <= 1e-5
    # data-parallel training step == the single-process step on the whole batch
    from gpu_util import seeded
    from wavedm_amd.training import Trainer
    tr = Trainer(cfg, dtype="f32")
    tr.load_state_dict(P.procedural_state_dict(cfg, seed=61))
    tr.loss_and_grads(seeded((4, 96, 16, 16), 401).to(dev), torch.tensor([990, 9, 500, 499]), seeded((4, 3, 16, 16), 402).to(dev))
    scale = float(tr.grads.abs().max())
    assert float((got["grads"] - tr.grads.cpu()).abs().max()) <= 1e-4 * scale
    tr.optimizer_step()
    big = tr.grads.cpu().abs() > 1e-3 * scale          # Adam's first step is sign-like: compare where the gradient is not rounding noise
    assert float((got["params1"] - tr.params.cpu())[big].abs().max()) <= 1e-6


def test_two_ranks_match_single_process(tmp_path):
    run_two_ranks(tmp_path, "gloo")


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form the driver's scaling leg may use) re-executes itself under
    torch.distributed.run and prints one JSON line with the rccl sub-record; gloo backend so that both ranks share this box's GPU."""
    import json
    repo = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["WAVEDM_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 128 and rec["config"]["outputs_finite"]
    assert rec["rccl"]["rccl_ranks"] == 2 and rec["rccl"]["weight_broadcast_s"] is not None
    assert rec["value"] > 0 and rec["scaling"] == "weak"


def _bench_ranks(n, extra, timeout=1500):
    import json
    repo = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["WAVEDM_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-extras"] + extra,
                       env=env, capture_output=True, text=True, timeout=timeout, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_eight_rank_dress_rehearsal_configs3():
    """The driver's 8-GPU scaling run, rehearsed on ONE device: `python bench.py --gpus 8` self-launches eight ranks (the reference's launch shape,
    train_weather_script.py:3: eight processes, one per GPU), rank 0 packs the weights, broadcast + adopt on the other seven, 64 crops per rank = BASELINE configs[3]'s
    global batch of 512, barriers and the max over ranks, the roofline leg on rank 0 while seven ranks wait, ragged-free all-gather after the clock.  gloo so that the
    eight ranks can share this box's GPU; the collectives' call sites are the ones RCCL will run."""
    rec = _bench_ranks(8, ["--ddim-steps", "5"])            # (the shortest run that has an x0_preds[-5])
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 512 and rec["config"]["outputs_finite"] and rec["scaling"] == "weak"
    assert rec["rccl"]["rccl_ranks"] == 8 and rec["rccl"]["weight_broadcast_s"] is not None and rec["rccl"]["rank_elapsed_s_max"] >= rec["rccl"]["rank_elapsed_s_min"] > 0
    assert rec["value"] > 0 and rec["roofline"]["launches"] > 0 and rec["cpu_baseline"] is None


def test_bench_eight_rank_dress_rehearsal_configs4():
    """configs[4] on eight ranks, both forms of SURVEY.md §8e: (i) replicas by image -- 8 whole 480x720 images per rank through DiffusiveRestoration.restore();
    (ii) ONE image patch-sharded -- its 45 patches split 6/6/6/6/6/5/5/5, one all-reduce(sum) of the partial sums and counts per DDIM step."""
    rec = _bench_ranks(8, ["--workload", "c4", "--ddim-steps", "5", "--batch", "2", "--no-roofline"])
    assert rec["n_gpus"] == 8 and rec["config"]["global_batch"] == 16 and rec["config"]["outputs_finite"] and rec["rccl"]["rccl_ranks"] == 8
    rec = _bench_ranks(8, ["--workload", "c4", "--ddim-steps", "5", "--patch-sharded"])
    assert rec["rccl"]["patch_shards"] == [6, 6, 6, 6, 6, 5, 5, 5] and rec["rccl"]["allreduce_bytes_per_step"] == 2 * 3 * 120 * 180 * 4
    assert rec["config"]["global_batch"] == 1 and rec["scaling"] == "strong" and rec["config"]["outputs_finite"] and rec["value"] > 0
    assert "patch-sharded x8" in rec["config"]["parallelism"]


def test_patch_sharded_restore_on_eight_ranks_matches_one_process(tmp_path):
    """The 45-patch split on the REAL geometry (one 480x720 image -> 120x180 wavelet domain, 64x64 patches every 16) over eight ranks on one device: the restored
    image equals the single-process restore() to the association of eight partial overlap sums (fp32; 1e-5 max-norm-relative), reduced-width UNet at resolution 64."""
    import wavedm_amd
    out = tmp_path / "ps8.pt"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", WDM_TEST_BACKEND="gloo", WDM_TEST_MODE="patch8")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(HERE, "dist_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(out)
    assert got["world"] == 8 and got["shards"] == [6, 6, 6, 6, 6, 5, 5, 5]
    d, args, img = patch8_setup(torch.device("cuda", 0))
    rest = wavedm_amd.DiffusiveRestoration(d, args, d.config, save_images=False)
    torch.manual_seed(77)
    outs, psnr = rest.restore([(img, ("one",), torch.zeros(1))], validation="raindrop", r=16)
    assert rel_linf(got["out"], outs[0].cpu()) <= 1e-5
    assert abs(got["psnr"] - psnr[0]) <= 1e-3
    # two images, automatic grouping on every rank (one group of two: the patch-sharded mode forms no timing-dependent partial groups)
    args2 = SimpleNamespace(**{k: v for k, v in vars(args).items() if k != "images_per_call"})
    rest2 = wavedm_amd.DiffusiveRestoration(d, args2, d.config, save_images=False)
    torch.manual_seed(78)
    outs2, _ = rest2.restore([(img, ("a",), torch.zeros(1)), (img.flip(-1), ("b",), torch.zeros(1))], validation="raindrop", r=16)
    assert len(got["out2"]) == 2 and all(rel_linf(got["out2"][k], outs2[k].cpu()) <= 1e-5 for k in range(2))


def patch8_setup(dev):
    """One 480x720 image and the raindrop_wavelet UNet at a quarter of its width (ch 32: 9.8 M parameters, same levels, attention at 16 x 16 and in the middle block)
    -- shared by the eight ranks and the single process."""
    import wavedm_amd
    cfg = P.raindrop_wavelet_config(image_size=64, ch=32)
    cfg.device = dev
    args = SimpleNamespace(resume="", sampling_timesteps=6, local_rank=dev.index, image_folder="/tmp/wdm_img", test_set="raindrop", grid_r=16, images_per_call=1)
    d = wavedm_amd.DenoisingDiffusion_Wavelet(args, cfg, generator=lambda x: x, dtype="f32")
    d.model.load_state_dict(P.procedural_state_dict(cfg), strict=True)
    img = torch.rand(1, 6, 480, 720, generator=torch.Generator().manual_seed(91))
    return d, args, img


def test_bench_refuses_more_gpus_than_present():
    """Asking for more RCCL ranks than the node has devices answers with one JSON line (value null, error text) and a non-zero exit."""
    import json
    repo = os.path.dirname(HERE)
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WAVEDM_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=300, cwd=repo)
    assert r.returncode != 0
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["value"] is None and "GPU" in rec["error"]
