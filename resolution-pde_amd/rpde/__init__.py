"""Host-side binding of the MI355X spectral hot path (librpde_hip.so)."""
from . import _lib, ops  # noqa: F401
from ._lib import RpdeError, load  # noqa: F401
