"""fusionsense_amd — the MI355X-native Gaussian-splatting hot path behind FusionSense's gsplat / nerfstudio operator
surface.  Importing the package touches no GPU."""
import os as _os

# The HIP runtime forces a host wait every DEBUG_CLR_MAX_BATCH_SIZE commands (default 1000) of an unsynchronised stream;
# now and then that wait sleeps for 5-20 ms — one training step in ~110 of a loop that never synchronises (DESIGN.md §7:
# always ~1000 commands after the last synchronisation).  The limit is read when the runtime initialises, i.e. at the first
# HIP call of the process: set here, where the product is imported, it holds for every user of the package that imports it
# before touching the GPU (bench.py used to be the only one that set it — ADVICE r5).  A caller's own setting wins.
_os.environ.setdefault("DEBUG_CLR_MAX_BATCH_SIZE", "16384")
