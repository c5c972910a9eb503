"""Ablation timings of the halo-patch split-f16 kernel on a ConvGRU-shaped convolution (diagnostic)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (initialises the HIP runtime the same way the product does)
from atdn_vslam_amd import _lib
L = C.CDLL(_lib.LIB_PATH)
out = (C.c_float * 8)()
names = ["gen3 full", "gen3 -global loads", "gen3 -loads -LDS stores", "gen3 -loads -stores -ds_reads",
         "gen4 LDS-DMA 16x16", "gen4 LDS-DMA 8x16", "gen2 16x16 tiles (512 thr)", "gen2 full"]
for (nimg, H, W, Cc, N, KH, KW) in ((8, 47, 154, 384, 256, 1, 5), (8, 47, 154, 256, 192, 3, 3), (8, 47, 154, 128, 256, 3, 3)):
    torch.cuda.synchronize()
    rc = L.atdn_microbench_conv(nimg, H, W, Cc, N, KH, KW, 20, out)
    assert rc == 0
    flop = 2.0 * nimg * H * W * N * KH * KW * Cc
    print("conv %dx%d C=%d N=%d B=%d: %.1f GFLOP algorithmic, MFMA floor %.1f us" % (KH, KW, Cc, N, nimg, flop / 1e9, 3 * flop / 2.5e15 * 1e6))
    for n, v in zip(names, out):
        print("   %-34s %8.1f us   %6.1f TF-equivalent" % (n, v, flop / v / 1e6))
