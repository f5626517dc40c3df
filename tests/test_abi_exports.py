"""The C-ABI library loads and exports every symbol include/pidm.h declares (no compute calls, no GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(REPO, "include", "pidm.h")
SO = os.path.join(REPO, "physicsinformeddiffusionmodels_amd", "csrc", "libpidm_hip.so")


def declared_symbols():
    txt = open(HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pidm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_documented_surface():
    syms = declared_symbols()
    for must in ("pidm_unet_forward", "pidm_unet_backward", "pidm_darcy_residual_fwd", "pidm_darcy_loss_fwd_bwd",
                 "pidm_qsample_nhwc", "pidm_psample_update", "pidm_conv_forward", "pidm_conv_wgrad", "pidm_last_error"):
        assert must in syms


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(SO)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    lib.pidm_backend.restype = ctypes.c_char_p
    assert lib.pidm_backend() == b"hip"
    assert lib.pidm_version() == 1


def test_product_path_fails_loudly_without_gpu():
    import torch
    from physicsinformeddiffusionmodels_amd._lib import PidmError
    from physicsinformeddiffusionmodels_amd.unet_model import Unet3D
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = Unet3D(dim=8, channels=2)
    with pytest.raises(PidmError):
        m(torch.randn(1, 256, 2), torch.tensor([1]))   # CPU tensors: no fallback


def test_missing_library_raises(tmp_path):
    from physicsinformeddiffusionmodels_amd._lib import PidmError, PidmLib
    with pytest.raises(PidmError):
        PidmLib(str(tmp_path / "nope.so"))


def test_communicator_entries_fail_cleanly_without_rccl():
    """pidm_comm_* in the host-emulated build (no device, no RCCL): an error code and a message, never a crash; a null communicator
    is rejected by pidm_allreduce_f32 / accepted by pidm_comm_destroy."""
    import ctypes as C
    from tests.emu_util import emu_lib
    L = emu_lib()
    buf = (C.c_ubyte * 128)()
    assert L.pidm_comm_unique_id(buf) != 0 and b"RCCL" in L.pidm_last_error()
    comm = C.c_void_p()
    assert L.pidm_comm_init(0, 1, bytes(128), C.byref(comm)) != 0 and not comm.value
    assert L.pidm_allreduce_f32(None, None, 0, 1, None) != 0
    assert L.pidm_comm_destroy(None) == 0
