"""Drop-in module names for the reference drivers: `from src.unet_model import Unet3D`, ... (main.py:6-11)."""
