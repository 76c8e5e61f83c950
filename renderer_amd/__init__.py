"""renderer_amd -- MI355X (gfx950) raster/shade path behind the reference's Drawer interface.

`api`     ctypes binding of libsxe_hip.so + `HipDrawer` (the Drawer trait)
`scenes`  deterministic synthetic scenes (inputs only)
`shard`   sort-first / frame-parallel work split and the RCCL tile gather
"""
from . import api  # noqa: F401
from .api import HipDrawer, SxeError  # noqa: F401
