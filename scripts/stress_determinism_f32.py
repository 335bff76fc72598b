#!/usr/bin/env python3
"""scripts/stress_determinism.py for the fp32-tensor modes (register-staged kernels, the buffer-store epilogue): repeated runs and a sub-batch
must reproduce the first run bit for bit."""
import os,sys,torch
sys.path.insert(0,"/root/repo")
from types import SimpleNamespace
import wavedm_amd
from wavedm_amd import procedural as P
torch.set_grad_enabled(False)
dev=torch.device("cuda",0)
cfg=P.raindrop_wavelet_config(); cfg.device=dev
args=SimpleNamespace(resume="",sampling_timesteps=5,local_rank=0,image_folder="/tmp/wdm",test_set="raindrop",grid_r=16,max_batch=64)
for dt in ("f32","f32x3"):
    d=wavedm_amd.DenoisingDiffusion_Wavelet(args,cfg,generator=lambda x:x,dtype=dt)
    d.model.load_state_dict(P.procedural_state_dict(cfg,seed=61),strict=True)
    r,x=P.synthetic_batch(16,patch_px=256,seed=61); r,x=r.to(dev),x.to(dev)
    ref=d.restore_batch(r,x)[0].clone(); bad=0
    for i in range(12):
        if not torch.equal(d.restore_batch(r,x)[0],ref): bad+=1
    sub=d.restore_batch(r[3:9].contiguous(),x[3:9].contiguous())[0]
    print(dt,"12 repeats, mismatches:",bad, "finite", bool(torch.isfinite(ref).all()), "sub-batch identical", torch.equal(sub, ref[3:9]))
