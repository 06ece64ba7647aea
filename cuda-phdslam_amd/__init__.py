"""cuda-phdslam_amd — MI355X-native (gfx950) Rao-Blackwellised GM-PHD-SLAM hot path.

The directory name carries a hyphen (it is the project's name), so import it with
    importlib.import_module("cuda-phdslam_amd")
The package holds the hand-written HIP kernels + C-ABI (csrc/, built into libphdslam.so), the
ctypes binding (_lib), the host-side mirror of the reference interface (filter), the multi-GPU
host glue (dist) and the synthetic workload generator (synthetic).
"""
from . import _lib  # noqa: F401
from ._lib import GAUSSIAN, MEAS, NOISE, POSE, Control, Options, PhdError, SlamConfig, default_config  # noqa: F401
from .filter import PhdFilter, load_config, load_controls, load_measurements, write_state_log  # noqa: F401
