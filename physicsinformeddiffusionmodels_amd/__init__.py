"""physicsinformeddiffusionmodels_amd - MI355X (gfx950) native engine for the UNet + PDE-residual hot path of
jhbastek/PhysicsInformedDiffusionModels, behind the reference's own Python API (see DESIGN.md / INTEGRATION.md)."""
from .unet_model import Unet3D  # noqa: F401
from .residuals_darcy import ResidualsDarcy  # noqa: F401
from .denoising_utils import DenoisingDiffusion, EMA, save_model, load_model, extract  # noqa: F401

__all__ = ["Unet3D", "ResidualsDarcy", "DenoisingDiffusion", "EMA", "save_model", "load_model", "extract"]
