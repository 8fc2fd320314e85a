"""omok-ai_amd — MI355X-native batched self-play engine for the `mcts` + `alpha-zero` hot path
of AcrylicShrimp/omok-ai.  The compute lives in csrc/ (HIP kernels behind the C ABI declared in
include/omok_mi355x.h); this package is the thin Python host mirror of the reference's crate API.
"""
from . import weights  # noqa: F401
from . import binding  # noqa: F401
from . import dist  # noqa: F401
from . import precision  # noqa: F401
from . import model_file  # noqa: F401
from .api import Engine, Environment, SelfPlay  # noqa: F401
