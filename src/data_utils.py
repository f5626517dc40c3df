"""Re-export of physicsinformeddiffusionmodels_amd.data_utils under the reference's module path (src/data_utils.py)."""
from physicsinformeddiffusionmodels_amd.data_utils import *  # noqa: F401,F403
from physicsinformeddiffusionmodels_amd import data_utils as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith('__')})
