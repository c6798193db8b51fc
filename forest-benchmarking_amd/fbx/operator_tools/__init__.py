"""operator_tools on MI355X -- same star-exports as forest/benchmarking/operator_tools/__init__.py."""
from .apply_superoperator import *  # noqa: F401,F403
from .channel_approximation import *  # noqa: F401,F403
from .compose_superoperators import *  # noqa: F401,F403
from .project_superoperators import *  # noqa: F401,F403
from .superoperator_transformations import *  # noqa: F401,F403
from .random_operators import *  # noqa: F401,F403
from .validate_operator import *  # noqa: F401,F403
from .validate_superoperator import *  # noqa: F401,F403
from . import calculational  # noqa: F401
