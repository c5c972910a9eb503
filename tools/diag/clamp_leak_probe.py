"""Diagnostic: which GPU test leaves unread clamps in the per-device saturation counter of the split-f16 format?
    python tools/diag/clamp_leak_probe.py tests/test_gpu_parity.py [more pytest arguments]
Runs pytest in-process with a plugin that reads (and resets) the device counter through its own small handle after every test
and prints the tests behind which the count was non-zero. (Reading resets the counter: later tests no longer see the leak —
a probe, not part of the suite.)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
import torch


class Probe:
    def __init__(self):
        self.net = None
        self.leaks = []

    def _read(self):
        from atdn_vslam_amd import _lib
        from atdn_vslam_amd import synthetic as syn
        from atdn_vslam_amd.modules import RAFTGMA
        if self.net is None:
            self.net = RAFTGMA(saturation_check_every=0)
            self.net.load_state_dict(syn.to_torch(syn.make_gma_state(seed=1)))
            self.net = self.net.to("cuda:0").eval()
            fr = torch.from_numpy(syn.make_frames(2, 128, 128, seed=3)).to("cuda:0")
            self.net._sat_pending = False
            self.net(fr[0:1], fr[1:2], iters=1, test_mode=True)
            self.h = next(iter(self.net._handles.values()))[0]
        out = torch.empty(1, dtype=torch.float32)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.lib().atdn_gma_debug_read(self.h, b"sf_clamped", C.c_void_p(out.data_ptr()), 1, st)
        return int(out[0])

    def pytest_runtest_setup(self, item):
        n = self._read()
        if n:
            self.leaks.append(("<before> " + item.nodeid, n))

    def pytest_runtest_teardown(self, item, nextitem):
        n = self._read()
        if n:
            self.leaks.append((item.nodeid, n))
            print("\nCLAMP LEAK: %d behind %s" % (n, item.nodeid), flush=True)


if __name__ == "__main__":
    p = Probe()
    rc = pytest.main(sys.argv[1:] + ["-q", "-m", "gpu", "-p", "no:cacheprovider"], plugins=[p])
    print("leaks:", p.leaks)
    sys.exit(0)
