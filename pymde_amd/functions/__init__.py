from pymde_amd.functions import losses  # noqa: F401
from pymde_amd.functions import penalties  # noqa: F401
from pymde_amd.functions.function import Function  # noqa: F401
