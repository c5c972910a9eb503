"""Timings of the halo kernel on encoder-shaped (thin) convolutions: tile heights and the ablation ladder of the 64-channel
12x16 block (diagnostic, not the product path). NIMG=16 runs the benchmark's clip length."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
out = (C.c_float * 12)()
names = ["8x16 px x 64 ch (2x2 waves)", "12x16 px x 64 ch (2x2 waves)", "16x16 px x 64 ch (4x2 waves)", "  12x16 minus epilogue",
         "  ... minus weight loads", "  ... minus LDS reads", "  ... minus patch refresh (bare MFMA)",
         "  12x16, only: no weight loads", "  12x16, only: no patch refresh", "  12x16, only: no LDS reads", "  12x16, full epilogue but every block stores to the same 192 pixels (L2)", ""]
NIMG = int(os.environ.get("NIMG", "8"))
for (nimg, H, W, Cc, N) in ((NIMG, 188, 616, 64, 64), (NIMG, 94, 308, 96, 96), (NIMG, 47, 154, 128, 128), (NIMG, 47, 154, 256, 192)):
    torch.cuda.synchronize()
    rc = L.atdn_microbench_conv_thin(nimg, H, W, Cc, N, 30, out)
    assert rc == 0
    flop = 2.0 * nimg * H * W * N * 9 * Cc
    print("conv 3x3 %dx%d C=%d N=%d B=%d: %.1f GFLOP algorithmic, 3x-f16 MFMA floor at 2.5 PF %.1f us" % (H, W, Cc, N, nimg, flop / 1e9, 3 * flop / 2.5e15 * 1e6))
    for n, v in zip(names, out):
        if v <= 0.0:
            continue
        print("   %-42s %8.1f us   %6.1f TF algorithmic   %6.0f TF executed" % (n, v, flop / v / 1e6, 3 * flop / v / 1e6))
