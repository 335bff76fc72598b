#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: restored images/sec, raindrop 64x64 patches, 100-step DDIM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input on every rank: DWT of the
rainy crops and of the HFRM stand-in (B,3,256,256) -> S=100 DDIM steps of the wavelet-domain UNet
on (B,96,64,64) -> IDWT of the restored sub-bands, inputs resident in HBM when the clock starts.
Workload at N=1 = BASELINE.json configs[1] (raindrop_wavelet 64x64, batch 64, 100 DDIM steps, bf16);
N>1 shards independent images: every rank runs its own batch of 64 (weak scaling, configs[3]),
weights are packed on rank 0 and broadcast over RCCL, outputs are all-gathered after the clock stops.

Prints ONE JSON line on rank 0 (see README / the driver contract) including
  roofline      live HIP-event timing of the dominant kernel (the fused 3x3 implicit-GEMM conv) inside this run
  cpu_baseline  the CPU oracle timed on this box's host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
# What the board SUSTAINS on random 16-bit operands with nothing but MFMAs in flight (register-resident v_mfma_f32_16x16x32 loop, no LDS / memory traffic:
# tools/mfma_power_ubench.hip, profiles/r05_mfma_power_ubench.log -- 1 845 TFLOP/s against 2 476 on all-zero operands: the power limit, not the pipe).  `roofline.peak`
# stays the nominal figure the contract names; `roofline.sustained` prices the same achieved rate against this one.
MFMA_16BIT_SUSTAINED_RANDOM_TFLOPS = 1845.0
# ... and when every MFMA's operands are fresh ds_read_b128 fragments of random data in LDS (0.375-0.5 reads per MFMA, 8-16 waves per CU, no barriers, no DMA, no
# global traffic: the same tool's LDS-fed mode, profiles/r05_mfma_power_ldsfed.log, r05_mfma_power_probe.log -- 1 420 ... 1 610 TFLOP/s by box; 1 950 ... 2 165 on zeros): the ceiling of ANY LDS-fed
# 16-bit MFMA loop on this board, which is what a convolution kernel is.  (A constant of the board CLASS: measure_board_roofs() below measures the board at hand.)
MFMA_16BIT_LDS_FED_RANDOM_TFLOPS = 1450.0


def measure_board_roofs():
    """What this board sustains right now: tools/abl_mfma_power (built by __graft_entry__.build() from tools/mfma_power_ubench.hip) held for 1.5 s per variant in a child
    process.  {} when the binary is not there or fails (the constants above then stand alone)."""
    import re
    import subprocess
    exe = os.path.join(REPO, "tools", "abl_mfma_power")
    out = {}
    if not os.path.isfile(exe):
        return out
    for mode in ("reg", "lds"):
        try:
            r = subprocess.run([exe], env=dict(os.environ, HOLD="1.5", MODE=mode), capture_output=True, text=True, timeout=60)
            m = re.search(r"([0-9.]+) TFLOP/s", r.stdout)
            if r.returncode == 0 and m:
                out[mode] = float(m.group(1))
        except Exception:                                   # noqa: BLE001  (calibration only: never takes the headline down)
            pass
    return out


def log(*a):
    print(*a, file=sys.stderr, flush=True)


JSON_OUT = None          # the descriptor the one JSON line goes to (main() moves everything else that targets fd 1 to stderr)


def fail(msg, n_gpus, rank=0):
    """A run that cannot start still answers with ONE JSON line on rank 0 (value null), then a non-zero exit."""
    if rank == 0:
        print(json.dumps({"metric": "restored images/sec, raindrop 64x64 patches, 100-step DDIM", "value": None, "unit": "img/s",
                          "n_gpus": n_gpus, "error": msg}), file=JSON_OUT or sys.__stdout__, flush=True)
    log(f"[bench] {msg}")
    sys.exit(2)


def self_launch(n, backend):
    import socket
    import subprocess
    if not torch.cuda.is_available() or (backend == "nccl" and torch.cuda.device_count() < n):
        fail(f"--gpus {n} but this node shows {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s)", n)
    with socket.socket() as s:                                   # a free rendezvous port on the loopback
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] self-launch: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (BASELINE configs[1]: 64)")
    ap.add_argument("--ddim-steps", type=int, default=100)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "f32x3"])
    ap.add_argument("--workload", default="c1", choices=["c1", "c2", "c4"],
                    help="c1: BASELINE configs[1] (default, the headline metric); c2: 128x128 patches, batch 256; "
                         "c4: whole 480x720 images, 45 stitched patches each, 50 DDIM steps (informational extra runs)")
    ap.add_argument("--images-per-call", type=int, default=0, help="c4 only: loader items restored per sampler call (SURVEY.md §8f-2); 0 = restore()'s own default "
                                                                   "(auto: as many same-sized images as fill the UNet calls)")
    ap.add_argument("--patch-sharded", action="store_true", help="c4 with N > 1 only: the latency form of SURVEY.md §8e-ii -- every rank works on the SAME image, its 45 patches "
                                                                 "split 6/6/6/6/6/5/5/5 over 8 ranks, one all-reduce(sum) of 2 x 3 x 120 x 180 floats per DDIM step")
    ap.add_argument("--full-length", action="store_true", help="c4 only: run the 4 steps restore() does not read (args.early_stop = False, the reference's step count)")
    ap.add_argument("--max-batch", type=int, default=0, help="UNet call batch cap (default max(batch, 64))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs of the default N=1 run (configs[2], configs[4]) AND the parity modes")
    ap.add_argument("--parity-only", action="store_true", help="with --no-extras: keep the parity-mode leg (f16 / f32x3 / f32 timed and checked), skip configs[2] / configs[4]")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    # stdout carries the ONE JSON line and nothing else: whatever the libraries underneath print (the restore() front end mirrors the reference's
    # console messages) goes to stderr
    # -- at the file-descriptor level too: native libraries (gloo's "connected to peer ranks", RCCL with NCCL_DEBUG) write to fd 1 directly
    global JSON_OUT
    if JSON_OUT is None and ("RANK" in os.environ or args.gpus == 1):
        sys.__stdout__.flush()
        JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    sys.stdout = sys.stderr

    backend = os.environ.get("WAVEDM_BENCH_BACKEND", "nccl")     # "nccl" is RCCL on ROCm; "gloo" only to smoke-test the N > 1 code path on one GPU
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: become the launcher (one rank per GPU, the reference's own launch shape --
        # train_weather_script.py:3 / eval_diffusion.py:83 run under torch.distributed.launch) and hand the ranks' exit code back
        sys.exit(self_launch(args.gpus, backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.ddim_steps < 5:
        fail(f"--ddim-steps {args.ddim_steps}: the restored image is built from x0_preds[-5] (restoration.py:108), so a run needs at least 5 steps", args.gpus, rank)
    if world != args.gpus:
        fail(f"--gpus {args.gpus} but WORLD_SIZE={world}: start as `python bench.py --gpus {args.gpus}` (self-launching) or under "
             f"torch.distributed.run --nproc-per-node {args.gpus}", args.gpus, rank)
    if not torch.cuda.is_available():
        fail("bench.py needs an MI355X (torch.cuda.is_available() is False)", args.gpus, rank)
    if backend == "nccl" and torch.cuda.device_count() < world:
        fail(f"--gpus {world} but this node shows {torch.cuda.device_count()} GPU(s)", args.gpus, rank)
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    torch.set_grad_enabled(False)

    from types import SimpleNamespace
    import wavedm_amd
    from wavedm_amd import _lib, parallel
    from wavedm_amd import procedural as P

    if args.workload == "c2":
        args.batch = 256 if args.batch == 64 else args.batch
    if args.workload == "c4":
        args.ddim_steps = 50 if args.ddim_steps == 100 else args.ddim_steps
        # per GPU: DiffusiveRestoration.restore() with its own defaults -- early stop at x0_preds[-5], 7 images per stitched sampler call (315 patches = one UNet
        # call under the cap of 384), groups pipelined -- over 14 images per pass
        multi = args.gpus > 1
        if args.patch_sharded:
            if not multi:
                fail("--patch-sharded needs --gpus > 1", args.gpus, rank)
            args.batch, args.images_per_call = (1 if args.batch == 64 else args.batch), 1
        elif args.batch == 64:
            args.batch = 14                                       # two sampler calls of seven images (restore()'s automatic grouping: 7 x 45 patches fill a UNet call)
        if not args.max_batch:
            args.max_batch = 384
    cfg = P.raindrop_wavelet_config(image_size=128 if args.workload == "c2" else 64)
    cfg.device = dev
    a = SimpleNamespace(resume="", sampling_timesteps=args.ddim_steps, local_rank=local_dev, image_folder="/tmp/wdm",
                        test_set="raindrop", grid_r=16, max_batch=args.max_batch or max(args.batch, 64),
                        images_per_call=args.images_per_call or None, early_stop=not args.full_length)
    t0 = time.time()
    d = wavedm_amd.DenoisingDiffusion_Wavelet(a, cfg, generator=lambda x: x, dtype=args.dtype)   # HFRM: identity stand-in
    sd = None
    if rank == 0:
        sd = P.procedural_state_dict(cfg, seed=61)          # random-init weights of the named architecture
        d.model.load_state_dict(sd, strict=True)
        d.model.pack_weights()
    bcast_s = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        tb = time.perf_counter()
        parallel.broadcast_weights(d.model, src=0)           # 324 MB packed buffer over RCCL/xGMI, once
        torch.cuda.synchronize()
        bcast_s = time.perf_counter() - tb
    torch.cuda.synchronize()
    if rank == 0:
        log(f"[bench] model ready in {time.time() - t0:.1f}s ({sum(p.numel() for p in d.model.parameters()) / 1e6:.2f} M params, "
            f"packed {d.model.packed_bytes() / 1e6:.0f} MB, dtype {args.dtype})")

    B = args.batch
    if args.workload == "c4":
        # whole images: DWT -> 45 overlapping 64x64 patches (r = 16) per image through the stitched sampler -> IDWT
        g = torch.Generator().manual_seed(61 + (0 if args.patch_sharded else rank))      # patch-sharded: every rank holds the same image
        imgs = [torch.rand(1, 6, 480, 720, generator=g) for _ in range(B)]
        restorer = wavedm_amd.DiffusiveRestoration(d, a, cfg, save_images=False)
        if args.patch_sharded:
            d.patch_group = True                                  # sampling.ddim_sample: this rank's slice of the patch list, one all-reduce per step
            args.no_roofline = True                               # (the roofline leg is one more pass on rank 0 ALONE: it would wait for the others' all-reduce)
        loader = [(im, f"img{k}", torch.zeros(1)) for k, im in enumerate(imgs)]

        def one_pass():
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                outs, _ = restorer.restore(loader, validation="raindrop", r=16)
            return torch.cat(outs), None, None
    else:
        rainy, x_T = P.synthetic_batch(B, patch_px=4 * cfg.data.image_size, seed=61 + rank)
        rainy, x_T = rainy.to(dev), x_T.to(dev)

        def one_pass():
            return d.restore_batch(rainy, x_T)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out, xs_last, x0 = one_pass()
    fence()
    elapsed = time.perf_counter() - t1
    rank_elapsed = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        tmin = tmax.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        rank_elapsed = [float(tmin.item()), float(tmax.item())]
        elapsed = rank_elapsed[1]                                        # the job is as slow as its slowest rank
        torch.cuda.synchronize()
        tg = time.perf_counter()
        gathered = parallel.all_gather_shards(out, B * world)          # outside the timed region
        torch.cuda.synchronize()
        gather_s, gather_bytes = time.perf_counter() - tg, gathered.numel() * gathered.element_size()
        assert gathered.shape[0] == B * world
    finite = bool(torch.isfinite(out).all())

    # ---- roofline leg: one more pass of the SAME workload with HIP events around every conv launch
    roofline = None
    if rank == 0 and not args.no_roofline:
        _lib.prof_enable(True)
        one_pass()
        torch.cuda.synchronize()
        shapes = [e for e in _lib.prof_report() if e["flops"] > 0]      # one row per (kernel configuration | layer shape); the GroupNorm launches
        _lib.prof_enable(False)                                         # (timed too, no flops) are not part of the matrix-kernel table
        agg = {}
        for e in shapes:
            k = e["kernel"].split("|")[0]
            r = agg.setdefault(k, dict(kernel=k, launches=0, ms=0.0, flops=0.0, bytes=0.0))
            for f in ("launches", "ms", "flops", "bytes"):
                r[f] += e[f]
        rep = sorted(agg.values(), key=lambda e: -e["ms"])
        tot_ms = sum(e["ms"] for e in rep)
        for e in rep:
            log(f"[bench] {e['kernel']:<36s} launches {e['launches']:6d}  avg {e['ms'] / e['launches'] * 1e3:8.1f} us  "
                f"{e['flops'] / e['ms'] / 1e9:7.1f} TFLOP/s  {e['bytes'] / e['ms'] / 1e6:7.1f} GB/s(alg)  {100 * e['ms'] / tot_ms:5.1f}% of conv time")
        for e in sorted(shapes, key=lambda e: -e["ms"]):
            log(f"[shape] {e['kernel']:<64s} n {e['launches']:5d}  avg {e['ms'] / e['launches'] * 1e3:8.1f} us  {e['flops'] / e['ms'] / 1e9:7.1f} TFLOP/s  "
                f"{100 * e['ms'] / tot_ms:5.1f}%")
        dom = rep[0]
        # f32x3: one fp32 product = four MFMA-units of bf16 work (two K = 32 bf16 MFMAs per 16 channels): 2500 / 4
        peak = MFMA_BF16_PEAK_TFLOPS if args.dtype in ("bf16", "f16") else MFMA_BF16_PEAK_TFLOPS / 4 if args.dtype == "f32x3" else MFMA_F32_PEAK_TFLOPS
        ach = dom["flops"] / dom["ms"] / 1e9
        # HBM traffic per launch of that kernel: PMC counters need their own rocprofv3 passes (never combined with the timed
        # run), so the committed summary of scripts/prof_r01.sh is quoted here when it covers the same kernel
        traffic, traffic_src = None, None
        for tag in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                tj = json.load(open(os.path.join(REPO, "profiles", f"{tag}_traffic.json")))
                traffic = round(tj["kernels"][dom["kernel"]]["hbm_bytes_per_launch"])
                traffic_src = f"profiles/{tag}_traffic.json"
                break
            except Exception:
                continue
        roofline = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": "quoted" if traffic is not None else None,
                    "traffic_unit": f"bytes/launch; QUOTED from the committed PMC summary {traffic_src} (rocprofv3 --pmc needs passes of its own: not measured in this run)",
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "kernel": dom["kernel"], "launches": dom["launches"],
                    "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "all_conv_tflops": round(sum(e["flops"] for e in rep) / tot_ms / 1e9, 2),
                    "conv_ms_per_pass": round(tot_ms, 2)}
        if args.dtype in ("bf16", "f16"):
            live = measure_board_roofs()          # this very board, right behind the timed passes (the boxes of the pool differ by ~10 % in what their power limit gives)
            if live.get("reg") and live.get("lds"):
                # every fraction on the line belongs to the timed box: the board-class constants are named, not priced against
                roofline["sustained"] = {"this_board": {
                    "register_resident": live["reg"], "lds_fed": live["lds"], "unit": "TFLOP/s",
                    "frac_of_register_resident": round(ach / live["reg"], 4), "frac_of_lds_fed": round(ach / live["lds"], 4),
                    "what": "tools/mfma_power_ubench.hip held for 1.5 s each on THIS board right after the timed passes (random bf16 operands): the MFMA-only register-resident "
                            "16x16x32 loop and the LDS-fed loop (0.375 ds_read_b128 per MFMA, nothing else); frac_* = achieved / that figure"},
                    "board_class": {"register_resident": MFMA_16BIT_SUSTAINED_RANDOM_TFLOPS, "lds_fed": MFMA_16BIT_LDS_FED_RANDOM_TFLOPS,
                                    "what": "the same two loops as measured on other boards of the pool (profiles/r05_mfma_power_ubench.log, r05_mfma_power_ldsfed.log): context, no fraction taken"}}
            else:
                roofline["sustained"] = {"peak": MFMA_16BIT_SUSTAINED_RANDOM_TFLOPS, "frac": round(ach / MFMA_16BIT_SUSTAINED_RANDOM_TFLOPS, 4),
                                         "what": "calibration binary tools/abl_mfma_power missing or failed: board-CLASS constants (register-resident 16x16x32 MFMA loop on random 16-bit "
                                                 "operands, profiles/r05_mfma_power_ubench.log), not this board's",
                                         "lds_fed": {"peak": MFMA_16BIT_LDS_FED_RANDOM_TFLOPS, "frac": round(ach / MFMA_16BIT_LDS_FED_RANDOM_TFLOPS, 4)}}
        # Since round 3 the 3x3 stride-1 convs of the 16-pixel-multiple maps -- ONE kernel name until round 2 -- run as three tilings / schedules of the
        # same LDS-DMA design (256 x 128 persistent on the 64 x 64 maps, 256 x 128 on 16 x 16, 256 x 256 on 32 x 32), so "the dominant kernel" above is
        # the largest of the three; the family figure is the like-for-like successor of round 2's single-kernel number.
        fam = [e for e in rep if e["kernel"].startswith(("convdma_3x3s1_", "convdmap_3x3s1_"))]
        if fam:
            fms, ffl = sum(e["ms"] for e in fam), sum(e["flops"] for e in fam)
            roofline["family"] = {"name": "LDS-DMA 3x3 stride-1 conv, 16-pixel-multiple maps (conv_dma_kernel.h / conv_dmap_kernel.h / conv_dma256_kernel.h)",
                                  "achieved": round(ffl / fms / 1e9, 2), "frac": round(ffl / fms / 1e9 / peak, 4), "share_of_conv_time": round(fms / tot_ms, 4),
                                  "kernels": [{"kernel": e["kernel"], "launches": e["launches"], "avg_launch_us": round(e["ms"] / e["launches"] * 1e3, 2),
                                               "achieved": round(e["flops"] / e["ms"] / 1e9, 2), "frac": round(e["flops"] / e["ms"] / 1e9 / peak, 4)} for e in fam]}

    # ---- CPU baseline leg: the oracle on this box's host cores, bounded sample (BASELINE configs[0])
    cpu, cpu_sample = None, None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "c1":
        from oracle import wavedm_oracle as O
        nb, ns = 4, 10
        r4, xt4 = P.synthetic_batch(nb, patch_px=256, seed=61)
        xc = O.dwt_fwd(2 * r4 - 1)
        # pick the thread count that serves this small batch best (oneDNN on a 256-thread host is slower with all
        # threads than with a NUMA-domain-sized team), then time the bounded sample with it
        best_n, best_t = torch.get_num_threads(), None
        for nthr in sorted({8, 16, 32, 64, torch.get_num_threads()}):
            if nthr > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(nthr)
            O.ddim_batch(sd, cfg, xt4, xc, xc[:, 3:].contiguous(), 1, chunk=nb)      # 1 step (warm-up + probe)
            tp = time.perf_counter()
            O.ddim_batch(sd, cfg, xt4, xc, xc[:, 3:].contiguous(), 1, chunk=nb)
            tp = time.perf_counter() - tp
            log(f"[bench] cpu oracle probe: {nthr} threads -> {tp:.2f} s per {nb}-image step")
            if best_t is None or tp < best_t:
                best_n, best_t = nthr, tp
            all_n, all_t = nthr, tp                          # the last (largest) team = every hardware thread
        torch.set_num_threads(best_n)
        runs = []
        for _ in range(3):                                   # median of >= 3 runs
            tc = time.perf_counter()
            xs_cpu, x0_cpu = O.ddim_batch(sd, cfg, xt4, xc, xc[:, 3:].contiguous(), ns, chunk=nb)
            runs.append(time.perf_counter() - tc)
        tcpu = sorted(runs)[1]
        cpu_ips = nb / (tcpu * args.ddim_steps / ns)
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        cpu = {"value": round(cpu_ips, 5), "unit": "img/s", "cores": best_n, "kind": "port",
               "sample": f"{nb} images x {ns} DDIM steps of the torch-CPU oracle (fp32, oneDNN), median of 3 runs = {tcpu:.2f} s "
                         f"(runs {', '.join(f'{r:.2f}' for r in runs)}), scaled x{args.ddim_steps // ns} to {args.ddim_steps} steps",
               "host_cpus": os.cpu_count(), "cpu_model": cpu_model,
               "all_cores": {"cores": all_n, "value": round(nb / (all_t * args.ddim_steps), 5), "unit": "img/s",
                             "sample": "1-step probe with every hardware thread (BASELINE.md §3 'all host cores'), scaled to the full length"}}
        cpu_sample = {"rainy": r4, "x_T": xt4, "xs_last": xs_cpu[-1], "x0_m5": x0_cpu[-5], "steps": ns}

    # ---- informational legs of the default N = 1 run (driver-observed numbers instead of prose)
    extras, parity_mode = None, None
    if rank == 0 and world == 1 and args.workload == "c1" and (not args.no_extras or args.parity_only):
        import contextlib, io
        extras = []

        last_runs = []

        def timed(fn, n_warm=1, passes=3):
            """median of `passes` timed passes after the warm-up; the individual times stay in last_runs (no single-pass figures)"""
            for _ in range(n_warm):
                fn()
            runs = []
            for _ in range(passes):
                torch.cuda.synchronize()
                tq = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                runs.append(time.perf_counter() - tq)
            last_runs[:] = runs
            return sorted(runs)[len(runs) // 2]

        def spread():
            return {"passes": len(last_runs), "ms_min": round(min(last_runs) * 1e3, 1), "ms_median": round(sorted(last_runs)[len(last_runs) // 2] * 1e3, 1),
                    "ms_max": round(max(last_runs) * 1e3, 1)}

        def conv_flops_per_pass(fn):
            _lib.prof_enable(True)
            fn()
            torch.cuda.synchronize()
            fl = sum(e["flops"] for e in _lib.prof_report())
            _lib.prof_enable(False)
            return fl

        if not args.no_extras:
            # configs[4] per GPU through the reference's own call surface, DiffusiveRestoration.restore() with ITS defaults (early stop at x0_preds[-5], automatic
            # images per sampler call, groups pipelined two deep), 32 whole 480x720 images = 4 groups of 8, at 50 DDIM steps (configs[4]) and at the reference's
            # default 25 (eval_diffusion.py:26).  Two legs per step count:
            #   sampler-only   : the 32 images already tensors in host memory, identity HFRM stand-in, no PNGs
            #   whole pipeline : 32 PNG pairs on disk -> wavedm_amd.datasets.RainDrop loader (PIL decode + LANCZOS resizes, pinned DataLoader built OUTSIDE the clock,
            #                    its workers start inside) -> device HFRM with procedural weights (models/arch.py, once per image) -> DWT -> stitched sampler -> IDWT ->
            #                    three PSNRs on the device -> 8-bit conversion on the device + seven PNGs per image through the asynchronous writer, flushed inside the clock
            try:
                import tempfile, shutil, copy
                import numpy as np
                from PIL import Image
                from wavedm_amd.datasets import RainDrop
                N4 = 32
                root = tempfile.mkdtemp(prefix="wdm_c4_")
                rng = np.random.default_rng(44)
                for sub_ in ("raindrop_test", "train"):
                    for leaf in ("input", "gt"):
                        os.makedirs(os.path.join(root, "raindrop", sub_, leaf))
                for k in range(N4):
                    clean = rng.integers(0, 256, (480, 720, 3), dtype=np.uint8)
                    drop = np.clip(clean.astype(np.int16) + rng.integers(-40, 41, (480, 720, 3)), 0, 255).astype(np.uint8)
                    Image.fromarray(drop).save(os.path.join(root, "raindrop", "raindrop_test", "input", f"{k}_rain.png"), compress_level=1)
                    Image.fromarray(clean).save(os.path.join(root, "raindrop", "raindrop_test", "gt", f"{k}_clean.png"), compress_level=1)
                cfg4 = copy.deepcopy(cfg)
                cfg4.device = dev
                cfg4.data.data_dir = root
                cfg4.data.num_workers = 8
                g4 = torch.Generator().manual_seed(4)
                loader4 = [(torch.rand(1, 6, 480, 720, generator=g4), f"img{k}", torch.zeros(1)) for k in range(N4)]
                ident = d.generator
                hfrm = d._make_generator("procedural", args.dtype)
                for S4 in (50, 25):
                    a4 = SimpleNamespace(**vars(a))
                    a4.sampling_timesteps, a4.images_per_call, a4.max_batch, a4.early_stop, a4.world_size, a4.rank = S4, None, None, True, 1, 0
                    a4.image_folder = os.path.join(root, f"out{S4}")
                    d.args = a4
                    d.generator = ident
                    rest4 = wavedm_amd.DiffusiveRestoration(d, a4, cfg, save_images=False)
                    per_call = rest4.images_per_call_for(120, 180, 16)

                    def pass_c4():
                        with contextlib.redirect_stdout(io.StringIO()):
                            rest4.restore(loader4, validation="raindrop", r=16)
                    t4 = timed(pass_c4)
                    sp4 = spread()
                    base = {"images": N4, "ddim_steps": S4, "steps_run": S4 - 4, "images_per_sampler_call": per_call, "unet_call_cap": rest4._max_batch(), "unit": "img/s", "steps": 3, "warmup": 1}
                    extras.append(dict(base, workload=f"BASELINE.json configs[4] per GPU, SAMPLER-ONLY leg of DiffusiveRestoration.restore(): {N4} whole 480x720 images in host memory, 45 "
                                                      f"stitched 64x64 patches each (r = 16), {S4} DDIM steps (early stop at x0_preds[-5]: {S4 - 4} run), identity HFRM stand-in, no PNGs",
                                       value=round(N4 / t4, 3), ms_per_step=round(t4 * 1e3, 1), spread=sp4))
                    log(f"[bench] extra configs[4] S={S4} sampler-only: {N4 / t4:.2f} img/s ({t4 * 1e3:.0f} ms per {N4} images, {per_call} per call)")
                    d.generator = hfrm
                    rest5 = wavedm_amd.DiffusiveRestoration(d, a4, cfg4, save_images=True)
                    _, val_loader = RainDrop(a4, cfg4).get_loaders(parse_patches=False, validation="raindrop")      # built once, outside the clock

                    def pass_c4_real():
                        with contextlib.redirect_stdout(io.StringIO()):
                            rest5.restore(val_loader, validation="raindrop", r=16)          # ends with writer.flush(): every PNG is on disk
                    t5 = timed(pass_c4_real)
                    n_png = len(os.listdir(os.path.join(a4.image_folder, cfg4.data.dataset, "raindrop")))
                    extras.append(dict(base, workload=f"BASELINE.json configs[4] per GPU, WHOLE restore() pipeline: {N4} PNG pairs on disk -> RainDrop loader (PIL, 8 workers, built outside "
                                                      f"the clock) -> device HFRM (procedural weights, 15.9 M parameters) -> DWT -> 45 stitched 64x64 patches per image, {S4} DDIM steps (early "
                                                      f"stop: {S4 - 4} run) -> IDWT -> PSNR x3 on the device -> u8 + 7 PNGs per image (async writer, flushed inside the clock)",
                                       value=round(N4 / t5, 3), ms_per_step=round(t5 * 1e3, 1), spread=spread(), pngs_written=n_png, pipeline_over_sampler_only=round(t4 / t5, 3)))
                    log(f"[bench] extra configs[4] S={S4} whole pipeline (real HFRM, loader, PNGs): {N4 / t5:.2f} img/s ({t5 * 1e3:.0f} ms per {N4} images, {n_png} PNGs) = "
                        f"{t4 / t5:.3f} of sampler-only")
                    rest5.writer.close()
                    del rest4, rest5, val_loader
                d.generator = ident
                d.args = a
                del loader4, hfrm
                shutil.rmtree(root, ignore_errors=True)
            except Exception as e:                                      # the informational leg must not take the headline down with it
                import traceback
                log(f"[bench] extra configs[4] FAILED: {type(e).__name__}: {e}\n{traceback.format_exc()}")
                extras.append({"workload": "BASELINE.json configs[4] restore() legs", "value": None, "error": f"{type(e).__name__}: {e}"})
                d.args = a
            # configs[2]: 128x128 wavelet-domain patches, batch 256, 100 steps (its own 163 M-parameter UNet: attention sits one level deeper)
            cfg2 = P.raindrop_wavelet_config(image_size=128)
            cfg2.device = dev
            a2 = SimpleNamespace(**vars(a))
            a2.sampling_timesteps, a2.max_batch = 100, 64
            d2 = wavedm_amd.DenoisingDiffusion_Wavelet(a2, cfg2, generator=lambda x: x, dtype=args.dtype)
            d2.model.load_state_dict(P.procedural_state_dict(cfg2, seed=61), strict=True)
            r2, x2 = P.synthetic_batch(256, patch_px=512, seed=62)
            r2, x2 = r2.to(dev), x2.to(dev)
            a2.sampling_timesteps = 10
            d2.restore_batch(r2, x2)                                            # warm-up: 10 steps of the same shapes (first-use costs, clocks)
            a2.sampling_timesteps = 100
            t2 = timed(lambda: d2.restore_batch(r2, x2), n_warm=0)
            sp2 = spread()
            a2.sampling_timesteps = 5
            fl2 = conv_flops_per_pass(lambda: d2.restore_batch(r2, x2)) * 20
            extras.append({"workload": "BASELINE.json configs[2]: 256 patches of 128x128 (512x512 px crops), 100 DDIM steps", "value": round(256 / t2, 3),
                           "unit": "img/s", "ms_per_step": round(t2 * 1e3, 1), "steps": 3, "warmup": "10 DDIM steps of the same batch", "spread": sp2, "conv_tflops": round(fl2 / t2 / 1e12, 1)})
            log(f"[bench] extra configs[2]: {256 / t2:.2f} img/s ({t2:.2f} s per 256 patches)")
            del d2, r2, x2
            torch.cuda.empty_cache()
        # parity modes: fp32 tensors end to end -- the modes that meet north_star's 1e-3.  "f32x3" (fast): every product of the contractions
        # as three bf16 MFMAs on hi/lo-split operands; "f32" (exact): v_mfma_f32_16x16x4_f32 chains, bit-for-bit fp32 FMA order.
        def rel(u, v):
            return float((u.double() - v.double()).abs().max() / v.double().abs().max())
        a.sampling_timesteps = 10
        modes = {}
        headline_rel = None
        ref32 = None                 # exact-f32 HIP results of crops 0-3 of the timed batch, full length: the yardstick of every faster mode below
        for name in ("f32", "f16", "f32x3"):
            df = wavedm_amd.DenoisingDiffusion_Wavelet(a, cfg, generator=lambda x: x, dtype=name)
            df.model.load_state_dict(sd, strict=True)
            if name != "f32":
                # the tolerance-conformant modes as first-class numbers: the SAME 64 crops as the headline, ALL ddim steps,
                # 1 warm-up pass + 3 timed passes, and their own roofline leg (events around every conv launch of one more pass)
                a.sampling_timesteps = args.ddim_steps
                _, xl_m, x0_m = df.restore_batch(rainy, x_T)
                torch.cuda.synchronize()
                tq = time.perf_counter()
                for _ in range(3):
                    df.restore_batch(rainy, x_T)
                torch.cuda.synchronize()
                tp_ = (time.perf_counter() - tq) / 3
                ips = B / tp_
                m = {"dtype": name, "value": round(ips, 3), "unit": "img/s", "ms_per_step": round(tp_ * 1e3, 2), "steps": args.ddim_steps, "passes": 3, "warmup": 1,
                     "sample": f"the headline's {B} crops, all {args.ddim_steps} DDIM steps, mean of 3 passes after 1 warm-up pass", "tolerance": 1e-3}
                if ref32 is not None:
                    m["rel_linf_vs_f32_full_length"] = float(f"{max(rel(xl_m[:4].cpu(), ref32[0]), rel(x0_m[:4].cpu(), ref32[1])):.3e}")
                _lib.prof_enable(True)
                df.restore_batch(rainy, x_T)
                torch.cuda.synchronize()
                sh3 = [e for e in _lib.prof_report() if e["flops"] > 0]
                _lib.prof_enable(False)
                agg3 = {}
                for e in sh3:
                    k = e["kernel"].split("|")[0]
                    r_ = agg3.setdefault(k, dict(kernel=k, launches=0, ms=0.0, flops=0.0))
                    for f in ("launches", "ms", "flops"):
                        r_[f] += e[f]
                rep3 = sorted(agg3.values(), key=lambda e: -e["ms"])
                tot3 = sum(e["ms"] for e in rep3)
                for e in rep3[:8]:
                    log(f"[bench] {name} {e['kernel']:<40s} launches {e['launches']:6d}  avg {e['ms'] / e['launches'] * 1e3:8.1f} us  {e['flops'] / e['ms'] / 1e9:7.1f} TFLOP/s  {100 * e['ms'] / tot3:5.1f}%")
                d3 = rep3[0]
                # f32x3: one fp32 product = two full-rate bf16 MFMAs with the split operands, four times the MFMA time of a bf16 product: 2500 / 4;
                # f16: v_mfma_f32_16x16x32_f16 runs at the bf16 MFMA's rate
                pk3 = MFMA_BF16_PEAK_TFLOPS / 4 if name == "f32x3" else MFMA_BF16_PEAK_TFLOPS
                m["roofline"] = {"bound": "mfma", "kernel": d3["kernel"], "launches": d3["launches"], "avg_launch_us": round(d3["ms"] / d3["launches"] * 1e3, 2),
                                 "achieved": round(d3["flops"] / d3["ms"] / 1e9, 2), "peak": pk3,
                                 "unit": "TFLOP/s (fp32-equivalent: 2 M N K per product)" if name == "f32x3" else "TFLOP/s",
                                 "frac": round(d3["flops"] / d3["ms"] / 1e9 / pk3, 4), "all_conv_tflops": round(sum(e["flops"] for e in rep3) / tot3 / 1e9, 2),
                                 "conv_ms_per_pass": round(tot3, 2), "profile": f"profiles/r06_kernel_stats_{name}.md"}
                a.sampling_timesteps = 10
            else:
                rp, xp = P.synthetic_batch(B, patch_px=256, seed=63)
                rp, xp = rp.to(dev), xp.to(dev)
                tp_ = timed(lambda: df.restore_batch(rp, xp))
                ips = B / (tp_ * args.ddim_steps / 10)
                m = {"dtype": name, "value": round(ips, 3), "unit": "img/s",
                     "sample": f"{B} crops x 10 DDIM steps = {tp_ * 1e3:.0f} ms, scaled x{args.ddim_steps // 10} to {args.ddim_steps} steps", "tolerance": 1e-3}
            if cpu_sample is not None:
                _, xl, x0g = df.restore_batch(cpu_sample["rainy"].to(dev), cpu_sample["x_T"].to(dev))
                m["rel_linf_vs_oracle"] = float(f"{max(rel(xl.cpu(), cpu_sample['xs_last']), rel(x0g.cpu(), cpu_sample['x0_m5'])):.3e}")
                if cpu:
                    m["speedup_vs_cpu"] = round(ips / cpu["value"], 1)
            if name == "f32" and xs_last is not None:
                # the HEADLINE run's own outputs, full length: the first four of its 64 crops again in exact fp32 (an image's result does not depend
                # on the batch it sits in), compared with what the timed bf16 passes produced for them
                a.sampling_timesteps = args.ddim_steps
                _, xl32, x032 = df.restore_batch(rainy[:4].contiguous(), x_T[:4].contiguous())
                a.sampling_timesteps = 10
                ref32 = (xl32.cpu(), x032.cpu())
                headline_rel = max(rel(xs_last[:4].cpu(), ref32[0]), rel(x0[:4].cpu(), ref32[1]))
            modes[name] = m
            log(f"[bench] parity mode {name}: {ips:.2f} img/s {m}")
            del df
            torch.cuda.empty_cache()
        # the parity mode = the FASTEST mode whose full-length deviation measured inside north_star's 1e-3 in this very run (crops 0-3 of the timed batch, all DDIM
        # steps, against the exact-f32 HIP path); the other conformant mode rides along
        conformant = [m for m in (modes["f16"], modes["f32x3"]) if m.get("rel_linf_vs_f32_full_length", 1.0) <= 1e-3]
        parity_mode = dict(max(conformant, key=lambda m: m["value"]) if conformant else modes["f32x3"])
        parity_mode["selected_by"] = ("fastest mode with rel_linf_vs_f32_full_length <= 1e-3 in this run -- measured on SAMPLER OUTPUTS (xs[-1], x0_preds[-5] over all DDIM steps); "
                                      "f16: a single UNet forward alone sits at 1.15e-3 ... 1.34e-3 (DESIGN 3.9)" if conformant else
                                      "no fast mode measured <= 1e-3 in this run: f32x3 reported, see its rel_linf_vs_f32_full_length")
        parity_mode["f32x3"] = modes["f32x3"]
        parity_mode["f16"] = modes["f16"]
        parity_mode["exact_f32"] = modes["f32"]
        if headline_rel is not None:
            parity_mode["rel_linf_bf16_vs_f32_headline"] = float(f"{headline_rel:.3e}")
            parity_mode["headline_checked_on"] = (f"crops 0-3 of the timed batch, all {args.ddim_steps} DDIM steps: xs[-1] and x0_preds[-5] of the timed bf16 passes "
                                                  "against the exact-f32 HIP path (itself <= 2e-6 from the CPU oracle over 100 steps: tests/test_gpu_unet.py)")
        if cpu_sample is not None:
            _, xl, x0g = d.restore_batch(cpu_sample["rainy"].to(dev), cpu_sample["x_T"].to(dev))
            parity_mode["rel_linf_bf16_vs_oracle"] = float(f"{max(rel(xl.cpu(), cpu_sample['xs_last']), rel(x0g.cpu(), cpu_sample['x0_m5'])):.3e}")
            parity_mode["checked_on"] = "the cpu_baseline sample (4 crops x 10 DDIM steps): xs[-1] and x0_preds[-5] against the CPU oracle"
        a.sampling_timesteps = args.ddim_steps

    if rank == 0:
        sharded_one = args.workload == "c4" and args.patch_sharded                   # every rank worked on the SAME B images
        total_imgs = B * (1 if sharded_one else world) * args.steps
        res = {
            "metric": f"restored images/sec, raindrop 64x64 patches, {args.ddim_steps}-step DDIM" if args.workload == "c1" else
                      f"restored images/sec, workload {args.workload} (informational, not the headline metric)",
            "value": round(total_imgs / elapsed, 3),
            "unit": "img/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded U[0,1) 256x256 crops, procedural random-init weights of the raindrop_wavelet UNet)",
            "config": {"workload": (f"raindrop_wavelet 64x64 (256x256 px crops), batch {B}/GPU, {args.ddim_steps} DDIM steps, "
                                    f"DWT + UNet x{args.ddim_steps} + IDWT (BASELINE.json configs[1]{' x N, configs[3]' if world > 1 else ''})") if args.workload == "c1"
                       else (f"raindrop_wavelet 128x128 patches, batch {B}/GPU, {args.ddim_steps} DDIM steps (BASELINE.json configs[2])" if args.workload == "c2"
                             else f"{B} full 480x720 image(s)/GPU, 45 stitched 64x64 patches each, {args.ddim_steps} DDIM steps (BASELINE.json configs[4])"),
                       "global_batch": B * (1 if sharded_one else world), "ddim_steps": args.ddim_steps,
                       "parallelism": f"patch-sharded x{world} (one all-reduce per DDIM step)" if sharded_one else f"image-sharded x{world}",
                       "outputs_finite": finite},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if parity_mode:
            res["parity_mode"] = parity_mode
        if extras:
            res["extras"] = extras
        if world > 1:
            if sharded_one:
                res["scaling"] = "strong"                                            # the same image(s) whatever N is
            res["rccl"] = {"rccl_ranks": world, "backend": backend, "weight_broadcast_s": round(bcast_s, 4) if bcast_s is not None else None,
                           "weight_broadcast_bytes": int(d.model.packed_bytes()), "output_all_gather_s": round(gather_s, 4), "output_all_gather_bytes": int(gather_bytes),
                           **({"patch_shards": [parallel.shard_range(45 * B, r_, world)[1] - parallel.shard_range(45 * B, r_, world)[0] for r_ in range(world)],
                               "allreduce_bytes_per_step": 2 * B * 3 * 120 * 180 * 4} if sharded_one else {}),
                           "rank_elapsed_s_min": round(min(rank_elapsed), 4), "rank_elapsed_s_max": round(max(rank_elapsed), 4)}
        if cpu:
            res["speedup_vs_cpu"] = round(res["value"] / cpu["value"], 1)
        print(json.dumps(res), file=JSON_OUT or sys.__stdout__, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
