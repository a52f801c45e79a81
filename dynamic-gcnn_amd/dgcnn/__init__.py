"""dgcnn -- MI355X-native EdgeConv hot path behind the operator surface of
DeepLearnPhysics/dynamic-gcnn (dgcnn/__init__.py:1-5 exports io_factory, build, trainval, ops,
DGCNN_FLAGS; `main_funcs` holds the run loops).

Importing this package never needs a GPU; calling an op does (there is no CPU fallback)."""
from . import ops
from . import model
from .model import build
from .trainval import trainval
from .flags import DGCNN_FLAGS
from .iotool import io_factory
from . import main_funcs
from ._engine import variable_scope, ctx, reset

__all__ = ["ops", "model", "build", "trainval", "DGCNN_FLAGS", "io_factory", "main_funcs", "variable_scope", "ctx", "reset"]
