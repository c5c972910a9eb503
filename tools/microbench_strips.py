"""Read rate of the attention x V access pattern (diagnostic): every wave streams its own contiguous run, three 3 KB steps ahead,
against variations of layout, cache policy, load width, barrier coupling and ring depth (tools/microbench/microbench.hip)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
L.atdn_microbench_strips.argtypes = [C.c_long, C.c_int, C.c_int, C.POINTER(C.c_float)]
out = (C.c_float * 7)()
names = ["per-wave runs, nt (the kernel's pattern)", "block-interleaved runs (24 KB per block step), nt", "per-wave runs, default policy",
         "per-wave runs, 3 x 1 KB 16-B loads", "per-wave runs + 2 block barriers per step", "block-interleaved + 2 barriers per step",
         "per-wave runs, ring 6 deep"]
for nblocks in (464, 512, 232):
    assert L.atdn_microbench_strips(227 * 3072, nblocks, 10, out) == 0
    print("%d blocks of 8 runs x 681 KB (%.2f GB):" % (nblocks, nblocks * 8 * 226 * 3072 / 1e9) +
          "".join("\n   %-52s %7.0f GB/s" % (n, v) for n, v in zip(names, out)))
