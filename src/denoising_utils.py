"""Re-export of physicsinformeddiffusionmodels_amd.denoising_utils under the reference's module path (src/denoising_utils.py)."""
from physicsinformeddiffusionmodels_amd.denoising_utils import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd import denoising_utils as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
