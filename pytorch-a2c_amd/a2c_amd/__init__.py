"""a2c_amd: MI355X-native drop-in for the rollout+update hot path of grantsrb/PyTorch-A2C.

Same surface as the reference's ``a2c`` package for that path:
``a2c_amd.utils`` (discount, sample_action, next_state, cuda_if), ``a2c_amd.models`` (A3CModel,
ConvModel, FCModel, GRU, GRUFCModel, GRUModel), ``a2c_amd.runner`` (Runner, StatsRunner,
SequentialEnvironment), ``a2c_amd.updater`` (Updater), ``a2c_amd.preprocessing``.
All arithmetic runs in liba2c_mi355x.so (hand-written HIP for gfx950, C ABI in
include/a2c_mi355x.h); there is no CPU fallback.
"""
from . import _lib, ops, utils, models, optim, parallel, preprocessing, runner, updater, training  # noqa: F401
from .models import A3CModel, ConvModel, FCModel, GRU, GRUFCModel, GRUModel  # noqa: F401
from .runner import Runner, StatsRunner, SequentialEnvironment, HostEnvPool  # noqa: F401
from .updater import Updater  # noqa: F401
from .utils import discount, sample_action, next_state, cuda_if, try_key, deque_maxmin  # noqa: F401
