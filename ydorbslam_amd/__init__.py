"""ydorbslam_amd — MI355X-native (gfx950) ORB front-end, descriptor matcher and local-BA back-end.

The package is a thin ctypes mirror of the C ABI in include/ydorb/c_api.h (the drop-in boundary of the
reference's OrbExtractor / OrbMatcher / Optimizer hot path).  All computation happens in
ydorbslam_amd/libydorb.so (host C++ + hand-written HIP kernels).  There is no CPU fallback: importing
succeeds without the library, but every operator raises YdorbError until it is built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C ydorbslam_amd/csrc`).
"""
from ._lib import YdorbError, lib, library_path, build_library  # noqa: F401
from .extractor import OrbExtractor, KP_DTYPE  # noqa: F401
from .matcher import OrbMatcher, FrameView, FeatureVector, QUERY_DTYPE  # noqa: F401
from .optimizer import Optimizer  # noqa: F401
from .vocabulary import Vocabulary  # noqa: F401

__all__ = ["YdorbError", "lib", "library_path", "build_library", "OrbExtractor", "KP_DTYPE", "OrbMatcher", "FrameView",
           "FeatureVector", "QUERY_DTYPE", "Optimizer", "Vocabulary"]
