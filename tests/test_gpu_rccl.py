"""GPU, world size 2 on TWO devices over RCCL (backend "nccl"): the deployment shape of the multi-GPU path (the reference
launches one process per GPU, train_weather_script.py:3 / eval_diffusion.py:83, and broadcasts weights through DDP, ddm_wavelet.py:168).
Skipped on a one-GPU box (tests/test_gpu_dist.py covers the same code paths there with gloo).  Checked against single-process
results: `broadcast_weights` + adopt, the image-sharded `restore_batch` + all-gather (bit-identical), the patch-sharded sampler with
one all-reduce per DDIM step (<= 1e-5), `Trainer.allreduce_grads` + the optimiser step."""
import pytest
import torch

from test_gpu_dist import run_two_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
def test_two_ranks_rccl(tmp_path):
    run_two_ranks(tmp_path, "nccl")
