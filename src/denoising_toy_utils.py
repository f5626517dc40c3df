"""`main_toy.py` (BASELINE configs[0]) is the reference's CPU-only plumbing example: a 2-D point diffusion with its own
small MLP (`src/denoising_toy_utils.py` of the reference).  It has no UNet, no PDE residual and no GPU work, so it is outside
the hot path this engine replaces.  Run it with the reference's own `src/` package on PYTHONPATH."""
raise ImportError(__doc__)
