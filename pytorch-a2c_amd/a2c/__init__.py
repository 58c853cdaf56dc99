"""``import a2c`` alias: puts the MI355X package under the reference's own package name so
``from a2c.runner import Runner`` / ``from a2c.updater import Updater`` / ``from a2c.models import *``
/ ``from a2c.utils import discount`` resolve to the drop-in (add ``pytorch-a2c_amd`` to sys.path)."""
import sys

import a2c_amd
from a2c_amd import models, preprocessing, runner, training, updater, utils  # noqa: F401

for _name in ("models", "preprocessing", "runner", "training", "updater", "utils"):
    sys.modules[__name__ + "." + _name] = getattr(a2c_amd, _name)
