"""ctypes binding of librgs_hip.so (C ABI declared in include/rgs_hip.h).

There is no CPU fallback: if the HIP library cannot be loaded, importing the
product raises, and every failed call raises RgsError with the library's text.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
# RGS_LIB_SUFFIX selects a diagnostic / A-B build made with `make SUFFIX=...` (e.g. _stamps); unset = the product library
LIB_PATH = os.path.join(CSRC, "librgs_hip%s.so" % os.environ.get("RGS_LIB_SUFFIX", ""))

RGS_OK = 0
RGS_ERR_ARG = -1
RGS_ERR_HIP = -2
RGS_ERR_UNSUPPORTED = -3
RGS_ERR_NO_DEVICE = -4

VARIANT_AUTO, VARIANT_WAVE, VARIANT_HYBRID, VARIANT_HYBRID2, VARIANT_POOL, VARIANT_APOOL = 0, 1, 2, 3, 4, 5

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/rgs_hip.h one-to-one
SIGNATURES = {
    "rgs_last_error": (C.c_char_p, []),
    "rgs_state_size": (C.c_int, [C.c_int]),
    "rgs_create": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rgs_destroy": (C.c_int, [_vp]),
    "rgs_reset": (C.c_int, [_vp, C.c_double, C.c_int, _dp, C.c_int, C.c_double, C.c_int, C.c_double,
                            C.c_double, C.c_double, C.c_double]),
    "rgs_step": (C.c_int, [_vp, _dp]),
    "rgs_step_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "rgs_step_device_traj": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "rgs_record_size": (C.c_int, [C.c_int]),
    "rgs_step_device_record": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, _vp]),
    "rgs_log_open": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_char_p)]),
    "rgs_step_device_log": (C.c_int, [_vp, _vp, C.c_int64, C.c_int]),
    "rgs_log_close": (C.c_int, [_vp]),
    "rgs_sync": (C.c_int, [_vp]),
    "rgs_get_state": (C.c_int, [_vp, _dp]),
    "rgs_undecided": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "rgs_get_damage": (C.c_int, [_vp, _dp]),
    "rgs_get_time": (C.c_int, [_vp, _dp]),
    "rgs_set_state": (C.c_int, [_vp, _dp, _dp]),
    "rgs_set_damage": (C.c_int, [_vp, _dp]),
    "rgs_state_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "rgs_copy_state_to_device": (C.c_int, [_vp, _vp]),
    "rgs_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "rgs_device_alloc": (C.c_int, [_vp, C.c_int64, C.POINTER(_vp)]),
    "rgs_device_free": (C.c_int, [_vp, _vp]),
    "rgs_device_upload": (C.c_int, [_vp, _vp, _dp, C.c_int64]),
    "rgs_device_download": (C.c_int, [_vp, _dp, _vp, C.c_int64]),
    "rgs_step_roundtrip": (C.c_int, [_vp, _dp, _dp]),
    "rgs_host_alloc": (C.c_int, [C.c_int64, C.POINTER(_vp)]),
    "rgs_host_free": (C.c_int, [_vp]),
    "rgs_timer_start": (C.c_int, [_vp]),
    "rgs_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "rgs_get_counters": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "rgs_get_counters_ex": (C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int]),
    "rgs_get_stamps": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "rgs_debug_check_guards": (C.c_int, [_vp]),
    "rgs_debug_wave_log": (C.c_int, [_vp, C.POINTER(C.c_int64), C.c_int64]),
    "rgs_debug_inject_pool_timeout": (C.c_int, [_vp]),
    "rgs_set_variant": (C.c_int, [_vp, C.c_int]),
    "rgs_set_noise_mode": (C.c_int, [_vp, C.c_int]),
    "rgs_noise_is_exact": (C.c_int, []),
    "rgs_selftest_rng": (C.c_int, [C.c_int, C.c_longlong, C.c_int, C.c_int, _dp, _dp, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "rgs_set_rebalance": (C.c_int, [_vp, C.c_int]),
    "rgs_get_heat": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "rgs_get_rebalance_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rgs_set_fast_division": (C.c_int, [_vp, C.c_int]),
    "rgs_set_sparse_watch": (C.c_int, [_vp, C.c_int]),
    "rgs_division_info": (C.c_int, [C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.c_int,
                                    C.POINTER(C.c_int)]),
    "rgs_selftest_divtau": (C.c_int, [C.c_int, _dp, C.c_int64, C.c_double, _dp, _dp, _dp]),
    "rgs_selftest_math": (C.c_int, [C.c_int, _dp, _dp, C.c_int64, _dp, _dp]),
}


class RgsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("librgs_hip: %s (status %d)" % (text, code))
        self.code = code


_lib = None


def build(force=False):
    """Compile librgs_hip.so in-tree with hipcc for gfx950 (works without a GPU)."""
    args = ["make", "-j", str(min(8, os.cpu_count() or 1)), "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args + ["librgs_hip.so"], stdout=subprocess.DEVNULL)   # always the product library
    return LIB_PATH


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname as /opt/rocm's).  If this
    library were loaded first it would bind to the system copy and a later `import torch` would bring
    a second HIP runtime into the process: device pointers could then not be passed between the two
    (bench.py and montecarlo.run_closed_loop_on_device hand torch tensors to the stepper).  Loading
    torch's copy first -- without importing torch -- makes every later load resolve to that one.
    Set RGS_USE_SYSTEM_HIP=1 to skip this."""
    if os.environ.get("RGS_USE_SYSTEM_HIP"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    """Load the HIP library, failing loudly when it is missing."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "robo_game_py: %s is missing. Build it with `make -C %s` (needs hipcc); "
                "there is no CPU fallback." % (LIB_PATH, CSRC))
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            if os.environ.get("RGS_LIB_SUFFIX") and not hasattr(L, name):
                continue                   # an older A/B build (tools/ab_libs.sh) may lack the newest entry points
            fn = getattr(L, name)          # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != RGS_OK:
        raise RgsError(code, (lib().rgs_last_error() or b"").decode(errors="replace"))


def dptr(a):
    return a.ctypes.data_as(_dp)
