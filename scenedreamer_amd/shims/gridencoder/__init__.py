"""Drop-in for the reference's `gridencoder` package (gridencoder/__init__.py:1)."""
from scenedreamer_amd.gridencoder import GridEncoder, grid_encode  # noqa: F401
