"""a2c_amd: MI355X-native drop-in for the rollout+update hot path of grantsrb/PyTorch-A2C.

Same surface as the reference's ``a2c`` package for that path:
``a2c_amd.utils`` (discount, sample_action, next_state, cuda_if), ``a2c_amd.models`` (A3CModel,
ConvModel, FCModel, GRU, GRUFCModel, GRUModel), ``a2c_amd.runner`` (Runner, StatsRunner,
SequentialEnvironment), ``a2c_amd.updater`` (Updater), ``a2c_amd.preprocessing``.
All arithmetic runs in liba2c_mi355x.so (hand-written HIP for gfx950, C ABI in
include/a2c_mi355x.h); there is no CPU fallback.

Sub-modules and names are imported lazily: the env worker processes of the host pool
(``python -m a2c_amd.hostpool_worker``) import this package without pulling in torch or the GPU runtime.
"""
import importlib

_SUBMODULES = ("_lib", "ops", "utils", "models", "optim", "parallel", "preprocessing", "runner", "updater", "training",
               "engine", "hostpool", "hostpool_worker", "synthetic")
_NAMES = {
    "A3CModel": "models", "ConvModel": "models", "FCModel": "models", "GRU": "models", "GRUFCModel": "models",
    "GRUModel": "models",
    "Runner": "runner", "StatsRunner": "runner", "SequentialEnvironment": "runner", "HostEnvPool": "runner",
    "ProcessEnvPool": "hostpool",
    "Updater": "updater",
    "discount": "utils", "sample_action": "utils", "next_state": "utils", "cuda_if": "utils", "try_key": "utils",
    "deque_maxmin": "utils",
}
__all__ = list(_SUBMODULES) + list(_NAMES)


def __getattr__(name):
    if name in _SUBMODULES:
        return importlib.import_module("." + name, __name__)
    if name in _NAMES:
        return getattr(importlib.import_module("." + _NAMES[name], __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(__all__)
