"""resco_amd: MI355X-native vectorised traffic-signal RL environment (RESCO-compatible API).

The simulator is a HIP library (resco_amd/csrc) reached through the C ABI in include/resco_sim.h; the
Python layer mirrors the reference's MultiSignal / Signal / states / rewards / config surface.
"""
__all__ = ['MultiSignal', 'VecMultiSignal', 'BatchedSim', 'Scenario', 'states', 'rewards']


def __getattr__(name):      # lazy: importing the package must not require the built library
    if name in ('MultiSignal', 'VecMultiSignal'):
        from . import multi_signal
        return getattr(multi_signal, name)
    if name == 'BatchedSim':
        from .sim import BatchedSim
        return BatchedSim
    if name == 'Scenario':
        from .scenario import Scenario
        return Scenario
    if name in ('states', 'rewards'):
        import importlib
        return importlib.import_module('.' + name, __name__)
    raise AttributeError(name)
