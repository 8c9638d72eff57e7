"""Drop-in for the reference's compiled `_gridencoder` extension module
(gridencoder/src/bindings.cpp:5-8; imported at gridencoder/grid.py:9-12), backed by libsdnative."""
from scenedreamer_amd.ops import grid_encode_backward, grid_encode_forward  # noqa: F401
