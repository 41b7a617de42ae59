"""goal_force_amd — MI355X-native Goal-Force (Wan2.2-I2V-A14B + force ControlNet) sampling path.

Host side: Python mirrors of the reference's pipeline interfaces (diffsynth WanVideoPipeline /
FlowMatchScheduler / WanModel / ControlNet / force-map dataset).  Compute: hand-written HIP kernels for
gfx950 behind the C ABI in include/goalforce.h (libgoalforce_hip.so, loaded with ctypes).
"""
from ._lib import GoalForceError, LIB_PATH, version  # noqa: F401

__all__ = ["GoalForceError", "LIB_PATH", "version"]
