"""Executed-FLOP ceilings of the split-f16 MFMA pattern under DVFS (diagnostic): random vs zero operands,
registers vs LDS re-reads, 32x32x16 vs 16x16x32 f16 MFMA."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
out = (C.c_float * 8)()
torch.cuda.synchronize()
assert L.atdn_microbench_mfma(2000, 200, out) == 0
names = ["32x32x16 regs", "32x32x16 LDS re-read", "16x16x32 regs", "16x16x32 LDS re-read"]
for d, data in enumerate(("random", "zero")):
    for i, n in enumerate(names):
        print("%-7s %-22s %8.0f TF/s executed" % (data, n, out[d * 4 + i]))

# round 3: the conv-like loop (pixel operands from a conflict-free LDS image, weights in registers, two waves per SIMD)
out4 = (C.c_float * 4)()
assert L.atdn_microbench_mfma_convlike(2000, 100, out4) == 0
for d, data in enumerate(("random", "zero")):
    for i, n in enumerate(("32x32x16 conv-like (pitch 144)", "16x16x32 conv-like (pitch 160)")):
        print("%-7s %-32s %8.0f TF/s executed" % (data, n, out4[d * 2 + i]))
