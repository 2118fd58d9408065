"""pyvb_amd: MI355X-native variational Bayes for pyvb's linear-dynamical-system path.

Importing this package loads libpyvb_hip.so (built by `make -C pyvb_amd/csrc`); there is no
CPU fallback.  `pyvb_amd.synth` (input generation) is importable without the library.
"""
from . import synth  # noqa: F401


def __getattr__(name):
    if name in ("LDSBatch",):
        from .lds import LDSBatch
        return LDSBatch
    if name in ("nodes", "Network"):
        import importlib
        mod = importlib.import_module(".nodes" if name == "nodes" else ".network", __name__)
        return mod if name == "nodes" else mod.Network
    raise AttributeError(name)
