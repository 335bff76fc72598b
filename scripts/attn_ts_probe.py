import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, gpu_util as gu
C = 512
shapes = {}
for p in ("norm", "q", "k", "v", "proj_out"):
    shapes[p + ".weight"] = (C,) if p == "norm" else (C, C, 1, 1)
    shapes[p + ".bias"] = (C,)
sd = gu.blk_sd("at", shapes)
x = gu.seeded((64, C, 16, 16), 3)
for _ in range(3):
    y = gu.attn(sd, "at", x, "bf16")
print("ok", float(y.abs().max()))
