"""Re-export of physicsinformeddiffusionmodels_amd.unet_model under the reference's module path (src/unet_model.py)."""
from physicsinformeddiffusionmodels_amd.unet_model import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd import unet_model as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
