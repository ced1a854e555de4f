from .torch_nn import *      # noqa: F401,F403
from .torch_edge import *    # noqa: F401,F403
from .torch_vertex import *  # noqa: F401,F403
