"""ctypes binding of the C ABI declared in include/pidm.h.

The product path loads `csrc/libpidm_hip.so` (hand-written gfx950 kernels) and nothing else: there is no
CPU or eager-PyTorch fallback.  `PidmLib(path)` with an explicit path exists so the unit tests can bind
the host-emulated build of the same sources (tests/hipemu); package code never does that.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libpidm_hip.so")

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p


class UnetCfg(C.Structure):
    _fields_ = [("dim", C.c_int), ("channels", C.c_int), ("out_dim", C.c_int), ("n_levels", C.c_int),
                ("dim_mults", C.c_int * 8), ("heads", C.c_int), ("dim_head", C.c_int), ("groups", C.c_int),
                ("init_kernel", C.c_int), ("image_size", C.c_int), ("sigmoid_last_channel", C.c_int),
                ("self_condition", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("C0", C.c_int), ("C1", C.c_int),
                ("ld0", C.c_int), ("ld1", C.c_int), ("Cout", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int), ("transposed", C.c_int), ("out_nchw", C.c_int),
                ("ldo", C.c_int)]


class PidmError(RuntimeError):
    pass


_loaded: list = []      # every library this process has bound (the product build, and the emulated one under pytest)


class PidmLib:
    def __init__(self, path: str | None = None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise PidmError(
                f"{path} not found: the gfx950 engine is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C physicsinformeddiffusionmodels_amd/csrc`). There is no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.pidm_version.restype = C.c_int
        L.pidm_last_error.restype = C.c_char_p
        L.pidm_backend.restype = C.c_char_p
        i, f, sz = C.c_int, C.c_float, C.c_size_t
        self._sig("pidm_debug_reduce_table_uploads", [], C.c_longlong)
        self._sig("pidm_debug_launch_counts", [C.POINTER(C.c_longlong)])
        self._sig("pidm_prof_enable", [i])
        self._sig("pidm_prof_collect", [C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_double)])
        self._sig("pidm_prof_kernels_begin", [vp])
        self._sig("pidm_prof_kernels_collect", [C.c_char_p, sz], C.c_longlong)
        self._sig("pidm_darcy_residual_fwd", [vp, vp, f, f, vp, i, i, vp])
        self._sig("pidm_darcy_residual_bwd", [vp, vp, f, f, vp, i, i, vp])
        self._sig("pidm_darcy_loss_ws", [i, i], sz)
        self._sig("pidm_darcy_loss_fwd_bwd", [vp, vp, vp, vp, vp, f, f, f, f, vp, vp, vp, vp, i, i, vp])
        self._sig("pidm_darcy_loss_fwd_bwd_t", [vp, vp, vp, vp, vp, vp, f, f, f, f, vp, vp, vp, vp, i, i, vp])
        self._sig("pidm_darcy_jacobian_max", [vp, f, f, vp, i, i, vp])
        self._sig("pidm_qsample_nhwc", [vp, vp, vp, vp, vp, i, i, i, vp])
        self._sig("pidm_qsample_nhwc_t", [vp, vp, vp, vp, vp, vp, i, i, i, vp])
        self._sig("pidm_psample_update", [vp, vp, vp, f, f, f, vp, sz, vp])
        self._sig("pidm_mech_apply", [vp, vp, vp, vp, i, vp, vp, i, vp, vp, i, vp])
        self._sig("pidm_mech_solve_ws_bytes", [i, i], sz)
        self._sig("pidm_mech_solve", [vp, vp, vp, i, vp, vp, i, f, f, f, i, C.c_double, vp, vp, vp, vp, vp, vp, i, vp])
        self._sig("pidm_floating_material", [vp, f, i, vp, i, vp])
        self._sig("pidm_unet_num_cond_params", [vp])
        self._sig("pidm_unet_enable_cond", [vp, i])
        self._sig("pidm_unet_set_condition", [vp, vp])
        self._sig("pidm_clip_adam_ws_bytes", [], sz)
        self._sig("pidm_clip_adam_step", [vp, vp, vp, vp, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.c_longlong,
                                          C.c_double, vp, vp, vp])
        self._sig("pidm_clip_adam_ema_step", [vp, vp, vp, vp, vp, sz, C.c_double, C.c_double, C.c_double, C.c_double, C.c_longlong,
                                              C.c_double, C.c_double, vp, vp, vp])
        self._sig("pidm_ema_update", [vp, vp, sz, C.c_double, vp])
        self._sig("pidm_mech_loss_ws", [i], sz)
        self._sig("pidm_mech_loss_fwd_bwd", [vp, vp, vp, vp, vp, vp, vp, f, f, f, f, vp, i, vp, vp, i, vp, vp, vp, i, vp])
        self._sig("pidm_bilinear_resize", [vp, vp, i, i, i, vp])
        self._sig("pidm_mech_residual_fwd", [vp, vp, vp, vp, i, vp, vp, i, vp, vp, vp, i, vp])
        self._sig("pidm_mech_residual_bwd", [vp, vp, vp, i, vp, vp, i, vp, vp, vp, vp, i, vp])
        self._sig("pidm_unet_create", [C.POINTER(UnetCfg), C.POINTER(vp)])
        self._sig("pidm_unet_destroy", [vp], None)
        self._sig("pidm_unet_num_params", [vp])
        self._sig("pidm_unet_param_name", [vp, i], C.c_char_p)
        self._sig("pidm_unet_param_numel", [vp, i], sz)
        self._sig("pidm_unet_workspace_bytes", [vp, i, i], sz)
        self._sig("pidm_unet_bind", [vp, C.POINTER(vp), C.POINTER(vp)])
        self._sig("pidm_unet_forward", [vp, vp, vp, vp, i, i, i, vp, sz, vp])
        self._sig("pidm_unet_backward", [vp, vp, vp, i, vp, sz, vp])
        self._sig("pidm_unet_set_grad_events", [vp, i, C.POINTER(vp)])
        self._sig("pidm_unet_grad_phase_range", [vp, i, i, C.POINTER(C.c_int), C.POINTER(C.c_int)])
        self._sig("pidm_conv_packed_weight_floats", [C.POINTER(ConvDesc)], sz)
        self._sig("pidm_conv_pack_weights", [C.POINTER(ConvDesc), vp, vp, i, vp])
        self._sig("pidm_conv_forward", [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp])
        self._sig("pidm_conv_dgrad_packed_weight_floats", [C.POINTER(ConvDesc)], sz)
        self._sig("pidm_conv_dgrad", [C.POINTER(ConvDesc), vp, i, vp, vp, vp, i, vp])
        self._sig("pidm_conv_forward_gn_partials", [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, i, vp, vp])
        self._sig("pidm_conv_wgrad_ws", [C.POINTER(ConvDesc)], sz)
        self._sig("pidm_conv_wgrad", [C.POINTER(ConvDesc), vp, vp, vp, i, vp, vp, vp, vp])
        self._sig("pidm_linear_attention_ws", [i, i, i], sz)
        self._sig("pidm_linear_attention_forward", [vp, vp, vp, vp, vp, i, i, i, vp, vp])
        self._sig("pidm_linear_attention_backward", [vp, vp, vp, vp, vp, vp, i, i, i, vp, vp])
        self._sig("pidm_linear_attention_out_forward", [vp, vp, vp, vp, vp, i, vp, vp, vp, i, i, i, vp, vp])
        self._sig("pidm_linear_attention_out_backward_ws", [i, i, i, i], sz)
        self._sig("pidm_linear_attention_out_backward", [vp, vp, vp, vp, vp, i, vp, i, vp, vp, i, i, i, vp, vp])
        self._sig("pidm_lap_ws", [i, i, i, i], sz)
        self._sig("pidm_lap_saved_floats", [i, i, i], sz)
        self._sig("pidm_lap_forward", [vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp])
        self._sig("pidm_lap_backward", [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp])
        self._sig("pidm_debug_stream_trace", [vp])
        self._sig("pidm_debug_conv_rs_trace", [C.POINTER(C.c_ulonglong)])
        self._sig("pidm_debug_lap_trace", [vp])
        self._sig("pidm_reload_knobs", [])
        self._sig("pidm_comm_available", [])
        self._sig("pidm_comm_unique_id", [vp])
        self._sig("pidm_comm_init", [i, i, vp, C.POINTER(vp)])
        self._sig("pidm_allreduce_f32", [vp, vp, sz, i, vp])
        self._sig("pidm_comm_destroy", [vp])
        _loaded.append(self)
        if L.pidm_version() != 1:
            raise PidmError(f"{path}: ABI version {L.pidm_version()} != 1")

    def _sig(self, name, argtypes, restype=C.c_int):
        fn = getattr(self.lib, name, None)
        if fn is None:
            # a stale / partially built library: calling through default int conversions would truncate pointers
            raise PidmError(f"{self.path} does not export {name} (include/pidm.h): rebuild with "
                            f"`make -C physicsinformeddiffusionmodels_amd/csrc` - there is no fallback")
        fn.argtypes = argtypes
        fn.restype = restype

    @property
    def backend(self) -> str:
        return self.lib.pidm_backend().decode()

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            raise PidmError(f"{what}: {self.lib.pidm_last_error().decode()} (rc={rc})")

    def __getattr__(self, name):
        return getattr(self.lib, name)


_default: PidmLib | None = None


def reload_knobs():
    """The native library snapshots each PIDM_* tuning variable the first time a launcher reads it; code that changes one inside a
    live process (unit tests, A/B legs of bench.py) calls this so that the next launch re-reads the environment."""
    for lib in _loaded:
        lib.lib.pidm_reload_knobs()


def get_lib() -> PidmLib:
    """The product library (gfx950).  Raises if it has not been built - no fallback."""
    global _default
    if _default is None:
        # PIDM_LIBRARY: another BUILD of the same gfx950 library (A/B measurements of compiler flags on one box, tools/r04_*.sh);
        # it must still be a HIP build - the backend check below keeps the host emulator and any CPU stand-in out
        _default = PidmLib(os.environ.get("PIDM_LIBRARY") or DEFAULT_LIB)
        if os.environ.get("PIDM_LIBRARY") and _default.backend != "hip":
            raise PidmError(f"PIDM_LIBRARY={os.environ['PIDM_LIBRARY']} is a '{_default.backend}' build, not the gfx950 library")
    return _default


def ptr(t):
    """Raw pointer of a torch tensor (or None) as c_void_p."""
    if t is None:
        return vp(0)
    return vp(t.data_ptr())


def stream_ptr(device=None):
    import torch
    if device is not None and torch.device(device).type == "cuda":
        return vp(torch.cuda.current_stream(device).cuda_stream)
    return vp(0)
