"""Drop-in for the hot-path helpers of the reference's ``utils.py`` (lines 17-23, 72-78, 112-189): same
names, argument meaning and return shapes, computed by libscream_hip.so on the MI355X.  The open3d /
pickle / image helpers of the reference's utils.py are outside the registration hot path."""
from scream_amd.geometry import (integrate_trans, nn_search_pair, processbar, rigid_transform_3d,  # noqa: F401
                                 square_distance, transformation_error)

__all__ = ["square_distance", "nn_search_pair", "rigid_transform_3d", "integrate_trans", "transformation_error",
           "processbar"]
