"""Stand-in for imageio (not installed). Plotting only."""


def get_writer(*a, **k):
    raise NotImplementedError


def imread(*a, **k):
    raise NotImplementedError
