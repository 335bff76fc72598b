"""wavedm_amd -- MI355X-native (gfx950) implementation of WaveDM's sampling hot path.

Public names mirror the reference's `models` package (`models/__init__.py:1-3`) so that
`from wavedm_amd import DenoisingDiffusion_Wavelet, DiffusiveRestoration` replaces
`from models import ...`.  Importing this package does not load the HIP
library; the first call that needs it does, and raises if it was not built."""
from .wavelet import WaveletTransform
from .unet import DiffusionUNet
from .unet_global import DiffusionUNet_Global
from .arch import HFRM
from .ddm_wavelet import DenoisingDiffusion_Wavelet, data_transform, inverse_data_transform
from .restoration import DiffusiveRestoration, torchPSNR
from .sampling import get_beta_schedule, compute_alpha, overlapping_grid_indices, ddim_sample
from .datasets import RainDrop, RainDropDataset
from .imageio import AsyncImageWriter
from .training import Trainer

__all__ = ["WaveletTransform", "DiffusionUNet", "DiffusionUNet_Global", "DenoisingDiffusion_Wavelet", "DiffusiveRestoration",
           "data_transform", "inverse_data_transform", "torchPSNR", "get_beta_schedule", "compute_alpha",
           "overlapping_grid_indices", "ddim_sample", "HFRM", "RainDrop", "RainDropDataset", "AsyncImageWriter", "Trainer"]
