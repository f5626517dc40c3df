"""Reference module `src/helper_plot.py` is not part of the accelerated path: the finite-difference stencil engine lives in
csrc/k_darcy.hip, plotting and data generation are host-side utilities of the reference.  Import it from the
reference checkout if you need it."""
raise ImportError(__doc__)
