"""Stand-in for OpenCV (not installed). Only used by the off-path topopt evaluation."""
THRESH_BINARY = 0


def threshold(*a, **k):
    raise NotImplementedError


def connectedComponents(*a, **k):
    raise NotImplementedError
