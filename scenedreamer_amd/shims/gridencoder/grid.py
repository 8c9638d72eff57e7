from scenedreamer_amd.gridencoder import GridEncoder, grid_encode  # noqa: F401
