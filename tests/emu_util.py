"""Helpers for tests that run the kernel sources through the host emulator (tests/hipemu).
TEST INFRASTRUCTURE: builds tests/hipemu/_build/libpidm_emu.so from csrc/*.hip with the host compiler."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(REPO, "tests", "hipemu", "_build", "libpidm_emu.so")
CSRC = os.path.join(REPO, "physicsinformeddiffusionmodels_amd", "csrc")
_lib = None


def emu_lib():
    global _lib
    if _lib is None:
        subprocess.run(["make", "-C", CSRC, "-j8", "emu"], check=True, stdout=subprocess.DEVNULL)
        from physicsinformeddiffusionmodels_amd._lib import PidmLib
        _lib = PidmLib(EMU_SO)
        assert _lib.backend == "hipemu"
    return _lib
