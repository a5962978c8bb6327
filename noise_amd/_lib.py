"""ctypes binding of libnoise_hip.so (the C-ABI kernel library).

The library is required: there is no fallback path.  Importing this module without the built
library raises, and every wrapper raises RuntimeError(noise_last_error()) on a non-zero return.
"""
import ctypes
import os

import torch  # imported first so the HIP runtime torch ships is the one the library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
# NOISE_HIP_LIB: another build of the same library (diagnostic builds with phase stamps, see scripts/dbg)
LIB_PATH = os.environ.get("NOISE_HIP_LIB") or os.path.join(_HERE, "libnoise_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`) "
        "before importing noise_amd; there is no CPU fallback."
    )

lib = ctypes.CDLL(LIB_PATH)
lib.noise_last_error.restype = ctypes.c_char_p
lib.noise_version.restype = ctypes.c_int

_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_long
_D = ctypes.c_double
_F = ctypes.c_float

_SIGNATURES = {
    "noise_stft_rpad": [_I],
    "noise_stft_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_istft_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_istft_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_stft_fold_cols": [_I],
    "noise_stft_fwd_sym": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_istft_fwd_sym": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_istft_bwd_sym": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_stft_fwd_fft": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_stft_fwd_fft_ref": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "noise_istft_fwd_fft": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_istft_bwd_fft": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "noise_fft_supported": [_I],
    "noise_snr_mix": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _D, _D, _P],
    "noise_snr_mix_peaks": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _D, _D, _P],
    "noise_activity": [_P, _P, _I, _L, _I, _D, _D, _P],
}


def _bind(sigs):
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int


_bind(_SIGNATURES)


def register(sigs):
    """Bind further entry points (used by the modules that own them)."""
    _bind(sigs)
    _SIGNATURES.update(sigs)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# When PROFILE is a list, every entry-point call is bracketed by HIP events on the current stream and
# (name, algorithmic_bytes, start_event, end_event) is appended: bench.py turns that into per-kernel GB/s.
# PROFILE_ONLY (a set of entry-point names, or None for all) restricts the bracketing: two event records per call cost
# about 3.5 us of host time, which on a ~110-launch step is several per cent when every call is instrumented.
PROFILE = None
PROFILE_ONLY = None


def call(name, *args, nbytes=0):
    prof = PROFILE
    if prof is not None and PROFILE_ONLY is not None and name not in PROFILE_ONLY:
        prof = None
    if prof is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {lib.noise_last_error().decode()}")
    if prof is not None:
        ev1.record()
        prof.append((name, nbytes, ev0, ev1))


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("noise_amd kernels run on the GPU only: got a CPU tensor (no CPU fallback exists)")
