"""skder_amd -- MI355X (gfx950) all-pairs ANI engine behind skDER's skani call sites.

Only the hot path of raufs/skDER is here: the three functions of src/skDER/skder.py that spawn
`skani` (runSkaniTriangle, runSkaniDist, lowMemGreedyDerep) with the same names, argument meaning
and error behaviour, routed through the C ABI of libskder_amd.so (include/skder_amd.h)."""
from .skder import lowMemGreedyDerep, runSkaniDist, runSkaniTriangle  # noqa: F401
