"""Re-export of physicsinformeddiffusionmodels_amd.residuals_mechanics_K under the reference's module path."""
from physicsinformeddiffusionmodels_amd.residuals_mechanics_K import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd import residuals_mechanics_K as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
