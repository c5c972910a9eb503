"""What a barrier-separated two-set ping-pong costs per phase (tools/microbench/pp_probe.hip)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
out = (C.c_float * 8)()
phases = 2000
assert L.atdn_microbench_pp_probe(phases, 5, out) == 0
names = ["ping-pong, partner idle", "ping-pong, partner 16 ds_read_b128", "ping-pong, partner 16 ds_read + 4 global loads",
         "no barriers, all 8 waves MFMA"]
for nm, off in ((24, 0), (48, 4)):
    for i, n in enumerate(names):
        us = out[off + i]
        # per SIMD: ping-pong issues nm MFMAs per phase; the free-running mode 2 waves x nm per loop trip
        mf = phases * nm * (2 if i == 3 else 1)
        print("%2d MFMAs/phase  %-46s %8.1f us  = %6.1f ns per phase, %5.1f ns per MFMA (32 cycles @2.4 GHz = 13.3 ns)"
              % (nm, n, us, us * 1e3 / phases, us * 1e3 / mf))
