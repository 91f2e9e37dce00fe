"""oxli_amd -- MI355X-native engine for oxli's ``KmerCountTable`` count / consume / get path.

``from oxli_amd import KmerCountTable`` is the drop-in for ``from oxli import KmerCountTable`` on
that path.  The hashing and counting run in hand-written gfx950 kernels behind the C ABI of
``include/kct.h`` (``oxli_amd/csrc``); there is no CPU fallback.
"""
from ._lib import LIB_PATH, load  # noqa: F401
from .table import VERSION, KmerCountTable  # noqa: F401

__all__ = ["KmerCountTable", "VERSION", "load", "LIB_PATH"]
