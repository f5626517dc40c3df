"""Drop-in import path of the reference's `src.denoising_toy_utils` (BASELINE configs[0], main_toy.py: `from
src.denoising_toy_utils import *`): the plain-PyTorch restatement in the package, plus the names the star import is
expected to bring along (torch, nn, F, np, os)."""
import os  # noqa: F401

import numpy as np  # noqa: F401
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401
import torch.nn.functional as F  # noqa: F401

from physicsinformeddiffusionmodels_amd.denoising_toy_utils import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd.denoising_toy_utils import device  # noqa: F401
