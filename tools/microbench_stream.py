"""HBM stream rates of this chip (diagnostic): plain / non-temporal stores, reads, copy, and stores in scattered 4 KB / 128 B
runs — what the kernels that write the correlation volume and the attention matrix can expect from the write side."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
L.atdn_microbench_stream.argtypes = [C.c_long, C.c_int, C.POINTER(C.c_float)]
out = (C.c_float * 7)()
names = ["plain 16-B stores", "non-temporal 16-B stores", "16-B loads", "copy (read + write bytes)",
         "nt stores, 4 KB runs 30 KB apart", "nt stores, 128 B runs 30 KB apart",
         "nt stores, 64-B half lines (16 per instruction), halves by consecutive instructions"]
for gb in (1.0, 3.5):
    assert L.atdn_microbench_stream(int(gb * (1 << 30)), 10, out) == 0
    print("buffer %.1f GiB:" % gb + "".join("\n   %-36s %7.0f GB/s" % (n, v) for n, v in zip(names, out)))
