"""scream_amd: MI355X-native registration hot path of xujiabo/SCREAM.

Importing the package is cheap and CPU-safe (synthetic data, packing metadata); anything that
computes needs libscream_hip.so and an MI355X and raises ``ScreamHipError`` otherwise.
"""
from ._lib import ScreamHipError  # noqa: F401

__all__ = ["ScreamHipError"]
__version__ = "0.1.0"
