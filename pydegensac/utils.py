"""`pydegensac.utils` of the drop-in alias: the reference keeps its Python wrappers in a real module of this name
(src/pydegensac/utils.py:24-146: convert_cv2_kpts_to_xyA, convert_and_check, findHomography, findFundamentalMatrix and
the error-type tables), and user code may write `from pydegensac.utils import convert_and_check`."""
from pydegensac_amd.api import *                      # noqa: F401,F403
from pydegensac_amd.api import (convert_and_check, convert_cv2_kpts_to_xyA, findHomography, findFundamentalMatrix,   # noqa: F401
                                findHomography_, findFundamentalMatrix_)
