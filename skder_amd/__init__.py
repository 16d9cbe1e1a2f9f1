"""skder_amd -- MI355X (gfx950) all-pairs ANI engine behind skDER's skani call sites.

Only the hot path of raufs/skDER is here: the three functions of src/skDER/skder.py that spawn
`skani` (runSkaniTriangle, runSkaniDist, lowMemGreedyDerep) with the same names, argument meaning
and error behaviour, routed through the C ABI of libskder_amd.so (include/skder_amd.h)."""
# Tuning note (not applied here: importing this package never touches the host application's environment).  ROCm maps
# HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); in a process where RCCL and torch own several streams
# the library's two chaining queues can end up on one hardware queue and stop overlapping (a few per cent of the chain
# stage).  A launcher may export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises: bench.py does, INTEGRATION.md says so.
from .skder import lowMemGreedyDerep, runSkaniDist, runSkaniTriangle  # noqa: F401
