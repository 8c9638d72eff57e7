"""Drop-in for the reference's compiled `voxlib` extension module
(imaginaire/model_utils/gancraft/voxlib/voxlib.cpp:25-31), backed by libsdnative."""
from scenedreamer_amd.ops import (positional_encoding, positional_encoding_backward,  # noqa: F401
                                  ray_voxel_intersection_perspective, sp_trilinear_worldcoord,
                                  sp_trilinear_worldcoord_backward)
