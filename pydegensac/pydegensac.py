"""`pydegensac.pydegensac` of the drop-in alias: in the reference this is the pybind11 extension module
(src/pydegensac/bindings.cpp:469-506) that exports the two private entry points; here they are the ctypes shims over
libmi_degensac.so with the same names, argument order and defaults."""
from pydegensac_amd.api import findHomography_, findFundamentalMatrix_   # noqa: F401

__all__ = ["findHomography_", "findFundamentalMatrix_"]
