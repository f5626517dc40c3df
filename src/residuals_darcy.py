"""Re-export of physicsinformeddiffusionmodels_amd.residuals_darcy under the reference's module path (src/residuals_darcy.py)."""
from physicsinformeddiffusionmodels_amd.residuals_darcy import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd import residuals_darcy as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
