"""skder_amd -- MI355X (gfx950) all-pairs ANI engine behind skDER's skani call sites.

Only the hot path of raufs/skDER is here: the three functions of src/skDER/skder.py that spawn
`skani` (runSkaniTriangle, runSkaniDist, lowMemGreedyDerep) with the same names, argument meaning
and error behaviour, routed through the C ABI of libskder_amd.so (include/skder_amd.h)."""
import os as _os

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); in a process where RCCL and torch own several
# streams the library's two chaining queues can end up on one hardware queue and stop overlapping.  Only a default, and
# only effective when set before the HIP runtime initialises.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .skder import lowMemGreedyDerep, runSkaniDist, runSkaniTriangle  # noqa: F401
