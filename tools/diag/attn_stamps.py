"""Diagnostic: where a phase of attention x V spends its cycles. Needs a library built with -DATDN_ATTN_STAMP
(python -m atdn_vslam_amd.build --variant stamp -DATDN_ATTN_STAMP; ATDN_LIB_PATH=.../libatdn_hip_stamp.so): s_memtime stamps
around the segments of every phase, summed over the 227 chunks, for wave 0 (set A) and wave 4 (set B) of every block."""
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
buf = (C.c_uint * (1024 * 2 * 8))()
L = _lib.lib()
assert L.atdn_attn_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint32).reshape(1024, 2, 8).astype(np.float64)
nblk = B * 29
a = a[:nblk]
chunks = 228.0
names = [["multiply + decode", "barrier 1", "stage V^T + loads issue", "fragment reads", "-", "barrier 2"],
         ["loads issue", "fragment reads", "-", "barrier 1", "multiply + decode", "barrier 2"]]
for s in (0, 1):
    tot = a[:, s, :6].sum(axis=1)
    print("set %s: cycles per chunk, mean over %d blocks (total %.0f, min %.0f, max %.0f per chunk)" %
          ("AB"[s], nblk, tot.mean() / chunks, tot.min() / chunks, tot.max() / chunks))
    for k in range(6):
        col = a[:, s, k] / chunks
        print("   %-26s mean %7.0f   p10 %7.0f   p90 %7.0f" % (names[s][k], col.mean(), np.percentile(col, 10), np.percentile(col, 90)))
