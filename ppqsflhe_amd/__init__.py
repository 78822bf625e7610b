"""ppqsflhe_amd -- MI355X-native multikey-CKKS secure-aggregation hot path.

Python side of the C-ABI in include/mkckks.h (ctypes over the in-tree
libmkckks_hip.so).  There is no CPU fallback: importing works anywhere the
library loads, but every compute call needs a gfx950 device and fails loudly
otherwise.
"""
from .binding import Context, DeviceBuffer, MkckksError, lib_path, load_library  # noqa: F401
