"""Occupancy guard (no GPU: reads the code objects of the built library). A change that "only" touches an epilogue can move a
kernel across a register step — round 5: the 128-wide 3x3 kernel went from 244 to 276 registers, two waves per SIMD to one, and
the motion encoder lost 5 % before anyone looked. tests/golden/kernel_occupancy.json records, for every kernel of the hot path, the
waves per SIMD its register count allows and that it spills nothing; a kernel may gain occupancy, never lose it or start spilling
without this table being regenerated on purpose (tools/diag/kernel_regs.py prints the current numbers)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _current():
    from atdn_vslam_amd import _lib
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag", "kernel_regs.py"), _lib.LIB_PATH],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    for line in r.stdout.splitlines()[1:]:
        name, rest = line[:140].strip(), line[140:].split()
        if len(rest) == 5:
            v, _, sp, lds, w = (int(x) for x in rest)
            out[name] = {"vgpr": v, "spilled_dwords": sp, "lds": lds, "waves_per_simd": w}
    return out


@pytest.mark.timeout(900)
def test_hot_kernels_keep_their_occupancy_and_do_not_spill():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("needs the ROCm LLVM tools to read the code objects")
    import shutil
    if not (shutil.which("c++filt") or shutil.which("llvm-cxxfilt") or os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt")):
        pytest.skip("no demangler (c++filt / llvm-cxxfilt): the table is keyed by demangled kernel names")
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_occupancy.json")))
    got = _current()
    missing = [k for k in want if k not in got]
    assert not missing, "kernels of the table no longer in the library (regenerate the table if they were renamed): %s" % missing[:5]
    worse = []
    for k, w in want.items():
        g = got[k]
        if g["waves_per_simd"] < w["waves_per_simd"] or g["spilled_dwords"] > w["spilled_dwords"]:
            worse.append("%s: %d registers / %d spilled / %d waves per SIMD, table has %d / %d / %d"
                         % (k[:110], g["vgpr"], g["spilled_dwords"], g["waves_per_simd"], w["vgpr"], w["spilled_dwords"], w["waves_per_simd"]))
    assert not worse, "occupancy lost:\n" + "\n".join(worse)


if __name__ == "__main__":   # python tests/test_kernel_occupancy.py --regen : rewrite the table from the built library, on purpose
    if "--regen" in sys.argv:
        sys.path.insert(0, ROOT)
        old = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_occupancy.json")))
        cur = _current()
        # a kernel whose argument list changed keeps its place in the table under its new name (same name up to the "(")
        bases = {k.split("(")[0] for k in old}
        new = {k: v for k, v in cur.items() if k in old or k.split("(")[0] in bases}
        for k in sorted(set(new) - set(old)):
            was = [old[o] for o in old if o.split("(")[0] == k.split("(")[0]]
            print("renamed: %s  now %s, was %s" % (k[:100], new[k], was[:1]))
        json.dump(new, open(os.path.join(ROOT, "tests", "golden", "kernel_occupancy.json"), "w"), indent=0, sort_keys=True)
        print("table: %d entries (%d before)" % (len(new), len(old)))
