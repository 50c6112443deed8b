"""sicelore-2.1_amd -- MI355X-native barcode/UMI assignment path of SiCeLoRe 2.1.

Host-side Python above the C ABI of ``include/sicelore_mi.h`` (``csrc/libsicelore_mi.so``, hand-written HIP
for gfx950).  The directory name is not a Python identifier; import it through
``__graft_entry__.load_package()`` which registers it as ``sicelore_amd``.

There is no CPU fallback: every compute call goes through the HIP library and raises if it is missing.
"""
from . import codec  # noqa: F401
from .lib import (  # noqa: F401
    BC_RESULT_DTYPE,
    BC_WINDOW_DTYPE,
    CHIMERA_RESULT_DTYPE,
    SCAN_CONFIG_DTYPE,
    SCAN_RESULT_DTYPE,
    Context,
    DeviceSpan,
    SmiError,
    library_path,
    load_library,
)

__all__ = ["codec", "Context", "DeviceSpan", "SmiError", "load_library", "library_path", "BC_WINDOW_DTYPE", "BC_RESULT_DTYPE"]
