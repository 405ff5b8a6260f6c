"""pymde_amd -- minimum-distortion embedding on AMD Instinct MI355X (gfx950 / CDNA4).

A from-scratch, GPU-native implementation of the hot path of cvxgrp/pymde behind PyMDE's own
object API: ``MDE``, ``penalties.*`` / ``losses.*``, ``Centered / Standardized / Anchored``.
The average-distortion forward/backward, the constraint projections and the projected
L-BFGS step are hand-written HIP kernels in ``libmde_hip.so`` (C ABI: ``include/mde_hip.h``).
"""
__version__ = "0.1.0"

from pymde_amd.problem import MDE  # noqa: F401
from pymde_amd.constraints import Centered, Anchored, Standardized  # noqa: F401
from pymde_amd.functions import losses, penalties  # noqa: F401
from pymde_amd.util import align, all_edges, center, rotate, seed  # noqa: F401
from pymde_amd import quadratic  # noqa: F401
from pymde_amd import preprocess  # noqa: F401
from pymde_amd.graph import Graph  # noqa: F401
from pymde_amd.quadratic import pca  # noqa: F401
from pymde_amd.recipes import laplacian_embedding, preserve_distances, preserve_neighbors  # noqa: F401
