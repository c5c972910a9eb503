"""Wall time of the CLVO head's recurrent scan over a KITTI-00-length sequence (4,540 steps, one batch row): the per-step kernel
(ATDN_SCAN_PERSISTENT=0) against the persistent kernel (csrc/lstm_scan.hip), and the latter's experiment modes (ATDN_SCAN_MODE)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from atdn_vslam_amd import synthetic as syn
from atdn_vslam_amd.modules import ATDNVO

DEV = "cuda:0"
T = int(os.environ.get("T", "4540"))


def head(persistent):
    os.environ["ATDN_SCAN_PERSISTENT"] = "1" if persistent else "0"
    h = ATDNVO()
    h.load_state_dict(syn.to_torch(syn.make_clvo_state(seed=1)))
    h = h.to(DEV).eval()
    h.scan(torch.zeros(2, 1, 512, device=DEV))
    return h


def run(h, f, reps=5):
    h.scan(f); h.scan(f)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r, t, s = h.scan(f)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts)), r


r = np.random.RandomState(9)
f = torch.from_numpy((r.normal(0, 0.12, (1, 512)) + np.cumsum(r.normal(0, 0.01, (T, 512)), axis=0)).astype(np.float32)).to(DEV)[:, None, :]
per, one = head(False), head(True)
ms, ref = run(per, f)
print("per-step kernel              %7.2f ms  (%.2f us per step)" % (ms, ms * 1e3 / T))
for mode in [int(m) for m in os.environ.get("MODES", "0,1,2,3,4,5,6,7").split(",")]:
    os.environ["ATDN_SCAN_MODE"] = str(mode)
    ms, out = run(one, f)
    print("persistent, mode %d           %7.2f ms  (%.2f us per step)  max |rot - per-step| %.2e  finite %s"
          % (mode, ms, ms * 1e3 / T, float((out - ref).abs().max()), bool(torch.isfinite(out).all())))
