"""Timings of the halo-patch split-f16 kernels on ConvGRU-shaped convolutions (diagnostic, not the product path):
the halo kernel (v_mfma_f32_16x16x32_f16 loop) at three block widths and the ablation ladder of its 128-wide block.  ATDN_MB_ZERO=1 runs on all-zero
operands (the chip then holds ~2.4 GHz: the difference to the default run is the DVFS share)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (initialises the HIP runtime the same way the product does)
from atdn_vslam_amd import _lib
_lib.lib()
L = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libatdn_microbench.so"))
out = (C.c_float * 16)()
names = ["8x16 px x 256 ch (8 waves)", "8x16 px x 128 ch (4 waves)", "8x16 px x 64 ch (2x2 waves)",
         "  x128 minus epilogue", "  ... minus weight loads", "  ... minus LDS reads", "  ... minus patch refresh (bare MFMA)",
         "", "", "",
         "128-wide, SfBias epilogue (z|r shape only)", "128-wide, SfGruZR gate epilogue (z|r shape only)",
         "12x16 px x 64 ch (2x2 waves, 3x3 only)", "8x16 px x 96 ch (2x3 waves, 3x3, N % 96 == 0)",
         "12x16 px x 96 ch (2x3 waves, 3x3, N % 96 == 0)", ""]
SHAPES = ((8, 47, 154, 384, 256, 1, 5), (16, 47, 154, 384, 256, 1, 5), (8, 47, 154, 256, 192, 3, 3), (8, 47, 154, 128, 256, 3, 3))
if "--b16" in sys.argv:   # the refinement loop's convolutions at the benchmark's clip length
    SHAPES = ((16, 47, 154, 384, 128, 1, 5), (16, 47, 154, 384, 128, 5, 1), (16, 47, 154, 256, 192, 3, 3), (16, 47, 154, 256, 128, 3, 3),
              (16, 47, 154, 128, 64, 3, 3), (16, 47, 154, 128, 256, 3, 3))
for (nimg, H, W, Cc, N, KH, KW) in SHAPES:
    torch.cuda.synchronize()
    rc = L.atdn_microbench_conv(nimg, H, W, Cc, N, KH, KW, 50, out)
    assert rc == 0
    flop = 2.0 * nimg * H * W * N * KH * KW * Cc
    print("conv %dx%d C=%d N=%d B=%d: %.1f GFLOP algorithmic, 3x-f16 MFMA floor at 2.5 PF %.1f us" % (KH, KW, Cc, N, nimg, flop / 1e9, 3 * flop / 2.5e15 * 1e6))
    for n, v in zip(names, out):
        if v <= 0.0:
            continue
        print("   %-42s %8.1f us   %6.1f TF algorithmic   %6.0f TF executed" % (n, v, flop / v / 1e6, 3 * flop / v / 1e6))
