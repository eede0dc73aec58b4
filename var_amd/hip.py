"""ctypes binding of libvar_hip.so (include/var_hip.h).  The product path has no fallback: a missing or unloadable
library raises here, and every non-zero return code of a kernel launcher raises VarHipError."""
import ctypes
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VARHIP_LIB') or os.path.join(_HERE, 'libvar_hip.so')      # (VARHIP_LIB: A/B experiments against another build of the library)


class VarHipError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise VarHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              f'(or `make -C var_amd/csrc`). There is no CPU/PyTorch fallback for the sampling path.')
        self.so = ctypes.CDLL(LIB_PATH)
        self.fn = abi.bind(self.so, 'varhip_', with_stream=True)
        so = self.so
        so.varhip_timing_enable.argtypes = [ctypes.c_int]; so.varhip_timing_enable.restype = ctypes.c_int
        so.varhip_timing_reset.argtypes = []; so.varhip_timing_reset.restype = ctypes.c_int
        so.varhip_timing_read.argtypes = [ctypes.c_void_p] * 4; so.varhip_timing_read.restype = ctypes.c_int
        so.varhip_timing_name.argtypes = [ctypes.c_int]; so.varhip_timing_name.restype = ctypes.c_char_p
        so.varhip_gemm16_force_tile.argtypes = [ctypes.c_int]; so.varhip_gemm16_force_tile.restype = ctypes.c_int
        so.varhip_conv16_force_tile.argtypes = [ctypes.c_int]; so.varhip_conv16_force_tile.restype = ctypes.c_int
        so.varhip_gemm16_persistent.argtypes = [ctypes.c_int]; so.varhip_gemm16_persistent.restype = ctypes.c_int
        so.varhip_sampler_force_walk.argtypes = [ctypes.c_int]; so.varhip_sampler_force_walk.restype = ctypes.c_int

    def version(self) -> str:
        return self.fn['version']().decode()


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def _arg(a):
    if a is None:
        return None
    if hasattr(a, 'data_ptr'):
        return ctypes.c_void_p(a.data_ptr())
    return a


def current_stream() -> ctypes.c_void_p:
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name: str, *args, stream=None):
    """Launch `varhip_<name>` on torch's current stream (tensors are passed by data_ptr); raises on a non-zero code."""
    L = lib()
    st = current_stream() if stream is None else stream
    rc = L.fn[name](*[_arg(a) for a in args], st)
    if rc != 0:
        raise VarHipError(f'varhip_{name} returned {rc}' + (' (VARHIP_EINVAL: unsupported shape/argument)' if rc == abi.EINVAL else ' (HIP launch error %d)' % (-rc - 1000)))


def gn_scratch_elems(B, HW, C, G) -> int:
    return int(lib().fn['gn_scratch_elems'](B, HW, C, G))


def conv_gn_blocks(H, W, Cout, phase=False) -> int:
    """blocks per sample of the GroupNorm partials a conv can leave behind (0: unsupported shape)"""
    return int(lib().fn['conv_gn_blocks'](H, W, Cout, 1 if phase else 0))


def conv16_gn_fusable(B, H, W, Cin, Cout) -> bool:
    """can the 16-bit conv apply the GroupNorm + SiLU in front of it to its own input patch (varhip_gnconv3x3_nhwc_*)?"""
    return bool(lib().fn['conv16_gn_fusable'](B, H, W, Cin, Cout))


# ---- timing table --------------------------------------------------------------------------------------------------
NFAM = 15            # VARHIP_NFAM of include/var_hip.h


def timing_enable(on: bool, families=None):
    """families: names of the kernel families to time (None = all); each timed launch records two events on the stream"""
    so = lib().so
    mask = -1
    if families is not None:
        names = [so.varhip_timing_name(i).decode() for i in range(NFAM)]
        mask = sum(1 << names.index(f) for f in families)
    so.varhip_timing_select(mask)
    so.varhip_timing_enable(1 if on else 0)


def timing_reset():
    lib().so.varhip_timing_reset()


def timing_read():
    """-> {family: dict(ms, flops, bytes, launches)} (synchronises the recorded events)"""
    n = abi_nfam = NFAM
    ms = (ctypes.c_double * n)(); fl = (ctypes.c_double * n)(); by = (ctypes.c_double * n)(); ln = (ctypes.c_int64 * n)()
    lib().so.varhip_timing_read(ms, fl, by, ln)
    return {lib().so.varhip_timing_name(i).decode(): dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=int(ln[i])) for i in range(abi_nfam)}
