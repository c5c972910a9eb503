"""Diagnostic: where the life of a block of the ConvGRU kernels goes. Needs a library built with -DATDN_CONV_STAMP
(python -m atdn_vslam_amd.build --variant stamp -DATDN_CONV_STAMP; ATDN_LIB_PATH=.../libatdn_hip_stamp.so).
For the last launch of each of the four kernels (z|r and q, 1x5 and 5x1) of a B-pair forward: per block, cycles of the prologue,
the main loop (of which: waiting at the chunk barriers), the epilogue; from the 100 MHz real-time stamps the launch's span, how
many blocks are alive over time (occupancy), and blocks per CU."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from atdn_vslam_amd import _lib
from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import RAFTGMA

H, W, B = 376, 1232, int(os.environ.get("B", "16"))
net = RAFTGMA(max_batch=B, saturation_check_every=0)
net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
net = net.to("cuda:0").eval()
fr = torch.from_numpy(syn.make_frames(B + 1, H, W, seed=100)).cuda()
for _ in range(3):
    net.forward_sequence(fr, iters=12)
torch.cuda.synchronize()
SLOTS = 4096
buf = (C.c_ulonglong * (4 * SLOTS * 12))()
L = C.CDLL(_lib.LIB_PATH)
assert L.atdn_conv_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4, SLOTS, 12)
names = ["z|r 1x5", "z|r 5x1", "q 1x5", "q 5x1"]
nblk = [B * 10 * 6 * 2, B * 10 * 6 * 2, B * 10 * 6, B * 10 * 6]
for k in range(4):
    n = min(nblk[k], SLOTS)
    x = a[k, :n]
    r0, r1 = x[:, 0].astype(np.float64), x[:, 1].astype(np.float64)
    t_start = r0.min()
    span_us = (r1.max() - t_start) / 100.0
    life_us = (r1 - r0) / 100.0
    pro, main, bar, epi, tot = (x[:, j].astype(np.float64) for j in (2, 3, 4, 5, 7))
    hw = x[:, 6]
    cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)   # xcc, se, sh, cu
    print("%s: %d of %d blocks stamped, launch span %.1f us; block life mean %.1f us (p10 %.1f, p90 %.1f)"
          % (names[k], n, nblk[k], span_us, life_us.mean(), np.percentile(life_us, 10), np.percentile(life_us, 90)))
    print("   cycles per block (wave 0): prologue %.0f | main loop %.0f (of which chunk-barrier waits %.0f = %.1f %%) | epilogue %.0f | total %.0f"
          % (pro.mean(), main.mean(), bar.mean(), 100 * bar.mean() / main.mean(), epi.mean(), tot.mean()))
    print("   shares of a block's life: prologue %.1f %%, main loop %.1f %%, epilogue %.1f %%; clock %.2f GHz"
          % (100 * pro.mean() / tot.mean(), 100 * main.mean() / tot.mean(), 100 * epi.mean() / tot.mean(),
             tot.mean() / (life_us.mean() * 1e3)))
    es, ew, ea = (x[:, j].astype(np.float64).mean() for j in (8, 9, 10))
    print("   inside the epilogue (wave 0, 4 tiles): transpose through the slab %.0f | waiting for operands / older stores %.0f (stamp "
          "variant 2 only: a drain before the arithmetic) | arithmetic + store issue %.0f cycles" % (es, ew, ea))
    # occupancy over time: blocks alive at 20 sample points
    ts = np.linspace(0, span_us, 21)[:-1] + span_us / 40
    alive = [int(((r0 - t_start) / 100.0 <= t).sum() - ((r1 - t_start) / 100.0 <= t).sum()) for t in ts]
    print("   blocks alive over the launch (20 samples):", alive)
    ncu = len(np.unique(cu))
    starts = np.sort((r0 - t_start) / 100.0)
    print("   distinct CUs seen %d; blocks started in the first 5 us: %d; last block started at %.1f us" % (ncu, int((starts < 5).sum()), starts[-1]))
    # are the co-resident blocks of a CU in phase? For every block: the distance from its start to the nearest OTHER block start on
    # its CU (two blocks per CU: ~0 = they run in lock step, their epilogues never meet the partner's main loop; ~half a block
    # life = staggered)
    st_us = (r0 - t_start) / 100.0
    dists = []
    for cid in np.unique(cu):
        t = np.sort(st_us[cu == cid])
        if len(t) < 2:
            continue
        d = np.minimum(np.diff(t, prepend=t[0] - 1e9), np.diff(t, append=t[-1] + 1e9))
        dists.append(d)
    dists = np.concatenate(dists)
    print("   start-to-nearest-start on the same CU: median %.1f us, p25 %.1f, p75 %.1f (block life %.1f us; blocks per CU %.1f)"
          % (np.median(dists), np.percentile(dists, 25), np.percentile(dists, 75), life_us.mean(), n / float(ncu)))
    # early vs late blocks
    order = np.argsort(r0)
    q = n // 4
    print("   block life by start order: first quarter %.1f us, second %.1f, third %.1f, last %.1f"
          % tuple(life_us[order[i * q:(i + 1) * q]].mean() for i in range(4)))
