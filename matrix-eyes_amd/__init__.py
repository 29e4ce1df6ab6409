"""matrix-eyes_amd — MI355X (gfx950) back end for the Depth Pro hot path of zlogic/matrix-eyes.

The product is `libmatrixeyes_hip.so` (hand-written HIP kernels + a C ABI, `include/*.h`).  This
package is the host-side mirror of the reference's interface for that path
(`depth_pro::DepthProModelLoader`, `output::DepthMap`), a thin ctypes layer over the C ABI.
There is no CPU fallback: every compute call raises `MatrixEyesError` when the library or a GPU
is missing.
"""
__version__ = "0.1.0"

from .config import ModelConfig, expected_weights  # noqa: F401
from ._lib import MatrixEyesError, load_library, library_path  # noqa: F401
from .depth_pro import Context, DepthProModelLoader, IMG_SIZE  # noqa: F401
from .output import DepthMap, DeviceDepthMap, ImageOutputFormat, VertexMode  # noqa: F401
from .reconstruction import ReconstructionError, SourceImage, extract_depth  # noqa: F401
