"""mmda_amd - MI355X-native (gfx950) hot path of SoyeonHH/MMDA's MISA training step.

Public surface mirrors the reference's modules: ``models.MISA`` (alias ``Model``), ``solver.Solver``,
``config.get_config``, ``utils.{DiffLoss, CMD, ReverseLayerF, getBinaryTensor, to_gpu, to_cpu}``.
"""
from . import _lib  # noqa: F401
from .models import MISA, Model  # noqa: F401
from .config import Config, get_config, make_config  # noqa: F401
