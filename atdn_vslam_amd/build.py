"""Builds libatdn_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m atdn_vslam_amd.build [--force] [--microbench] [--variant NAME -DMACRO[=V] -Xhipcc-flag ...]

--microbench also builds the diagnostic micro-benchmark library (tools/microbench/, ablation builds of the kernels): it is
not part of the product and a break in it must not fail the product build, so it is only built on request
(tools/microbench_*.py ask for it themselves).

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container;
the resulting .so travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libatdn_hip.so")
# diagnostic micro-benchmarks (ablation builds of the kernels): their own library, not part of the product one
MB_SRC = os.path.join(os.path.dirname(HERE), "tools", "microbench", "microbench.hip")
MB_LIB = os.path.join(HERE, "libatdn_microbench.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations into v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32. In the
# epilogues that run beside other waves' MFMAs those are no faster than the scalar forms (MI355X_MICROARCH.md, constants table),
# and the register-pair shuffles they need are extra instructions in loops bound by instruction issue. Whole forward, same job:
# 41.67 -> 41.35 ms per 16 pairs (cnet -3 %, motion encoder -2 %, lookup -3 %, flow head -1.5 %; q gate +0.7 %, softmax +1 %).
# -mllvm -amdgpu-mfma-vgpr-form=1 (round 5): MFMA results in ordinary vector registers. Left to itself the register allocator keeps
# accumulators in the accumulation half of the file and pays for it in copies — the chunk loop of the 64-wide 3x3 kernel carried 24
# v_accvgpr_write + 24 v_accvgpr_read + 24 v_accvgpr_mov per 324 MFMAs (tools/diag/isa_loop_census.py) — and in registers: with the
# flag 198 of the library's 406 kernels use fewer (the 128-wide normalise-on-load kernel 276 -> 248: two waves per SIMD instead of
# one; the GEMM-shaped 128 x 128 kernel 176-184 -> 156: three instead of two; no kernel spills). fnet 4.53 -> 4.45 ms, the rest
# within noise (profiles/r05_ab_vgpr_form.txt); results are bit-identical (the flag moves registers, not arithmetic).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "atdn_hip.h"))
    return hs


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src, objdir=OBJ, defines=()):
    obj = os.path.join(objdir, src[:-4] + ".o")
    newest = max([_mtime(os.path.join(CSRC, src))] + [_mtime(h) for h in _headers()])
    if _mtime(obj) >= newest:
        return obj, False
    cmd = ([HIPCC] + FLAGS + [d[5:] if d.startswith("FLAG:") else "-D" + d for d in defines] +
           ["-c", os.path.join(CSRC, src), "-o", obj])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stdout))
    if r.stdout.strip():
        sys.stderr.write(r.stdout)
    return obj, True


def build(force=False, jobs=None, microbench=False, variant=None, defines=()):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path.
    `variant` + `defines`: a second build of the same sources with extra -D macros, into libatdn_hip_<variant>.so (its own
    object directory): A/B timing of two builds inside one GPU job via ATDN_LIB_PATH. Diagnostics only, never loaded by default."""
    objdir = OBJ if not variant else OBJ + "_" + variant
    lib = LIB if not variant else os.path.join(HERE, "libatdn_hip_%s.so" % variant)
    os.makedirs(objdir, exist_ok=True)
    if force:
        for f in os.listdir(objdir):
            os.remove(os.path.join(objdir, f))
    jobs = jobs or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        results = list(ex.map(lambda src: _compile(src, objdir, tuple(defines)), _sources()))
    objs = [o for o, _ in results]
    if variant:
        LIBV = lib
    else:
        LIBV = LIB
    if any(changed for _, changed in results) or not os.path.exists(LIBV) or _mtime(LIBV) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIBV] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    if microbench and not variant:
        build_microbench()
    return LIBV


def build_microbench():
    """The diagnostic library libatdn_microbench.so (links against libatdn_hip.so); returns its path."""
    if not os.path.exists(LIB):
        build()
    newest = max([_mtime(MB_SRC), _mtime(LIB)] + [_mtime(h) for h in _headers()])
    if _mtime(MB_LIB) >= newest:
        return MB_LIB
    cmd = [HIPCC] + FLAGS + ["-shared", MB_SRC, "-o", MB_LIB, "-L" + HERE, "-latdn_hip", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on the micro-benchmark library:\n%s" % r.stdout)
    return MB_LIB


if __name__ == "__main__":
    _variant = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    # -DMACRO[=V] defines; -Xflag passes `flag` to hipcc as it is (variant builds only: e.g. -X-fno-slp-vectorize)
    _defs = [a[2:] for a in sys.argv if a.startswith("-D")] + ["FLAG:" + a[2:] for a in sys.argv if a.startswith("-X")]
    print(build(force="--force" in sys.argv, microbench="--microbench" in sys.argv, variant=_variant, defines=_defs))
