"""Registers, LDS and occupancy of every kernel of a built library, from the code objects' metadata:
    python tools/diag/kernel_regs.py [path/to/lib.so] [name fragment]
For each kernel: vgpr_count (VGPR + AGPR on gfx950: what sets the waves per SIMD, 512 / count rounded down to the allocation
granule of 8), sgpr_count, spills, LDS bytes, and the waves per SIMD that follow (the smaller of the register and the LDS bound). A change that "only" touches an epilogue can
move a kernel across an occupancy step (round 5: 244 -> 276 registers = two waves per SIMD -> one, +5 % on every launch)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(lib):
    out = []
    with tempfile.TemporaryDirectory() as td:
        sec = os.path.join(td, "fatbin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, sec], check=True)
        blob = open(sec, "rb").read()
        # the fat binary is a concatenation of clang-offload-bundles; every device ELF starts with \x7fELF
        starts = [m.start() for m in re.finditer(b"\x7fELF", blob)]
        for i, st in enumerate(starts):
            co = os.path.join(td, "co%d.o" % i)
            open(co, "wb").write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            r = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            cur = {}
            for line in r.stdout.splitlines():
                m = re.match(r"\s*[-]?\s*\.(\w+):\s*(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "name" and v.startswith("_Z") or (k == "name" and cur.get("name") is None and not v.startswith("_")):
                    pass
                if k == "agpr_count" and cur:
                    out.append(cur)
                    cur = {}
                cur[k] = v
            if cur:
                out.append(cur)
    return [k for k in out if "vgpr_count" in k and "name" in k]


def demangle(names):
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        tool = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    if not tool:
        return list(names)
    r = subprocess.run([tool], input="\n".join(names), stdout=subprocess.PIPE, text=True)
    return r.stdout.splitlines()


def main():
    here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(here, "atdn_vslam_amd", "libatdn_hip.so")
    frag = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
    ks = kernels_of(lib)
    names = demangle([k["name"] for k in ks])
    rows = []
    for k, n in zip(ks, names):
        if frag and frag not in n:
            continue
        v = int(k["vgpr_count"])
        waves = min(8, 512 // max(8, -(-v // 8) * 8))
        # the LDS bound: blocks per CU by the 160 KB of LDS x waves per block / 4 SIMDs (a kernel whose register count allows three
        # waves but whose LDS allows two blocks of four waves runs at two)
        lds, wg = int(k.get("group_segment_fixed_size", 0)), int(k.get("max_flat_workgroup_size", 256))
        if lds > 0:
            waves = min(waves, max(1, (163840 // lds) * max(1, wg // 64) // 4))
        rows.append((n.replace("atdn::", "").replace("(anonymous namespace)::", ""), v, int(k.get("sgpr_count", 0)),
                     int(k.get("vgpr_spill_count", 0)), int(k.get("group_segment_fixed_size", 0)), waves))
    rows.sort()
    print("%-140s %5s %5s %6s %7s %s" % ("kernel", "vgpr", "sgpr", "spill", "lds", "waves/SIMD (registers, LDS)"))
    for r in rows:
        print("%-140s %5d %5d %6d %7d %d" % (r[0][:140], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main()
