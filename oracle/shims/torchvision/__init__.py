"""Stand-in for torchvision (not installed): only transforms.Resize on tensors is used
(src/residuals_mechanics_K.py:20, src/denoising_utils.py:67)."""
from . import transforms  # noqa: F401
