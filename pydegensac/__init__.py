"""`import pydegensac` drop-in alias of the MI355X-native implementation (pydegensac_amd).

Mirrors the reference package surface (src/pydegensac/__init__.py:1-4: `from .pydegensac import *`,
`from .utils import findHomography, findFundamentalMatrix, convert_cv2_kpts_to_xyA`), so a script written
against the reference — e.g. examples/simple-example.py:18-37 — runs unchanged on the HIP path.
"""
from pydegensac_amd import (findHomography, findFundamentalMatrix, convert_cv2_kpts_to_xyA,   # noqa: F401
                            findHomography_, findFundamentalMatrix_)
from . import utils, pydegensac                                                               # noqa: F401  (real submodules, as in the reference)

__version__ = "0.1.2+mi355x"            # the reference's version (src/pydegensac/__init__.py:1) + local tag
