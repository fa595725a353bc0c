"""image_processing_amd — MI355X-native (gfx950) replacement for the OpenCV CPU pixel chain of the
image_processing ROS nodes.  The product is csrc/ (HIP kernels + C ABI, built to lib/libimgproc_hip.so);
this package is the thin host side above that ABI."""
from . import _lib  # noqa: F401
from .pipeline import (D, E, ImgProc, ImgProcError, MORPH_BASELINE, MORPH_POLYGON_PATH, MORPH_POSE_ESTIMATE,  # noqa: F401
                       MorphOp, gaussian_taps, getStructuringElement)
