"""Host emulation of the step kernel behind BatchedSim's API (TEST INFRASTRUCTURE).

tests/hostemu/rs_emu.cpp compiles the kernel source (resco_amd/csrc/resco_step.h) for the CPU and runs the threads of a
workgroup sequentially; it exports the same C ABI, so the product's ctypes wrapper drives it unchanged.  Used by the
`-m "not gpu"` tests to compare the kernel's logic with the oracle; never imported by resco_amd/."""
import ctypes as C
import os
import subprocess

from resco_amd import sim as _sim

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'librs_emu.so')
_lib = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'], stdout=subprocess.DEVNULL)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        _lib = _sim.bind(C.CDLL(os.environ.get('RS_EMU_LIB') or build()))      # (RS_EMU_LIB: a debug build)
    return _lib


_lib_short = None


def lib_shortlists():
    """the build whose work lists hold 8 slots: they overflow, the phases fall back to the flags"""
    global _lib_short
    if _lib_short is None:
        build()
        _lib_short = _sim.bind(C.CDLL(os.path.join(_HERE, '_build', 'librs_emu_shortlists.so')))
    return _lib_short


class EmuSim(_sim.BatchedSim):
    """order: 0 = threads in ascending order, 1 = descending, 2 = shuffled per phase (the `device` argument of the ABI
    carries it: the emulation has no device)."""

    def __init__(self, scenario, n_envs, order=0, short_lists=False, **kw):
        kw.pop('device', None)
        self._short = short_lists
        super().__init__(scenario, n_envs, device=order, **kw)

    def _load(self):
        return lib_shortlists() if self._short else lib()

    def tensor(self, name):
        raise RuntimeError('the host emulation has no device tensors')
