import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """(lib, device): the host-emulated build of csrc on CPU tensors, or the real gfx950 library on cuda:0."""
    import torch
    if request.param == "emu":
        from tests.emu_util import emu_lib
        return emu_lib(), torch.device("cpu")
    from physicsinformeddiffusionmodels_amd._lib import get_lib
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    L = get_lib()
    assert L.backend == "hip"
    return L, torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _fresh_knobs():
    """The native library reads each PIDM_* knob once per process; tests flip knobs, so every test starts (and leaves) with a fresh
    snapshot."""
    from physicsinformeddiffusionmodels_amd._lib import reload_knobs
    reload_knobs()
    yield
    reload_knobs()


@pytest.fixture
def monkeypatch(monkeypatch):
    """pytest's monkeypatch whose setenv / delenv also drop the native library's knob snapshot (see _fresh_knobs)."""
    from physicsinformeddiffusionmodels_amd._lib import reload_knobs
    set_, del_ = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, prepend=None):
        set_(name, value, prepend)
        reload_knobs()

    def delenv(name, raising=True):
        del_(name, raising)
        reload_knobs()
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    return monkeypatch
