"""Golden fixture for RAFTGMA.forward(test_mode=False) — the per-iteration `flow_predictions` of the REFERENCE itself
(GMA wheel, core/network.py:106-129), on the seeded synthetic checkpoint and the C1 frames of make_golden.py.

Run only in the build container (needs /root/reference; it never travels):

    python tests/golden/make_golden_preds.py

Stores numbers only: every iteration's upsampled flow on a 4 x 4 pixel grid, its sum / absolute sum over the full image (the
generator checks that the last iteration IS gma_c1.npz:flow_up, the test-mode output of the same pair).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, ROOT, install_stubs  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, os.path.join(REF, "GMA-1.0.0-py3-none-any.whl"))
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from GMA.core.network import RAFTGMA
    from atdn_vslam.utils.gma_parameters import GMA_Parameters
    from atdn_vslam_amd import synthetic as syn

    gma = RAFTGMA(GMA_Parameters()).eval()
    gma.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
    with torch.no_grad():
        fr = torch.from_numpy(syn.make_frames(2, 160, 512, seed=3))
        preds = gma(fr[0:1], fr[1:2], iters=8, test_mode=False)
        assert isinstance(preds, list) and len(preds) == 8
        p = torch.stack(preds, 0)[:, 0]                       # [8, 2, 160, 512]
        assert np.array_equal(p[-1].numpy(), np.load(os.path.join(HERE, "gma_c1.npz"))["flow_up"])
        # with a flow_init (network.py:103-104), B = 2 (the pair and its reverse), 3 iterations
        r = np.random.RandomState(11)
        fi = torch.from_numpy(r.uniform(-2, 2, (2, 2, 20, 64)).astype(np.float32))
        preds2 = gma(torch.cat([fr[0:1], fr[1:2]]), torch.cat([fr[1:2], fr[0:1]]), iters=3, flow_init=fi, test_mode=False)
        p2 = torch.stack(preds2, 0)                           # [3, 2, 2, 160, 512]
    np.savez_compressed(
        os.path.join(HERE, "gma_preds.npz"), seed_weights=1, seed_frames=3, iters=8,
        preds_s4=p[:, :, ::4, ::4].numpy(), preds_sum=p.double().sum(dim=(2, 3)).numpy(),
        preds_abs=p.double().abs().sum(dim=(2, 3)).numpy(),
        flow_init=fi.numpy(), preds2_s4=p2[:, :, :, ::4, ::4].numpy(), preds2_abs=p2.double().abs().sum(dim=(3, 4)).numpy())
    print("written", os.path.join(HERE, "gma_preds.npz"))


if __name__ == "__main__":
    main()
