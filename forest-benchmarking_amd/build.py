"""Build libfbx.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfbx.so")
MAP = os.path.join(CSRC, "libfbx.map")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on"] + os.environ.get("FBX_EXTRA_FLAGS", "").split()
# Per-file flags: the instruction scheduler's max-ILP strategy (same-box A/Bs, scripts/ab_time*.py; results are bit-identical).
#   fbx_pgdb.hip       one-wavefront-per-SIMD PGDB kernels: a lone wavefront has nobody to hide its latencies behind: B = 1024
#                      12.85 -> 12.6 ms.  The register-starved two-wavefronts-per-SIMD kernel of fbx_pgdb_lean.hip runs at HALF
#                      its speed with it -- hence the two translation units.
#   fbx_pgdb3.hip      3-qubit kernel (1024-thread workgroups, 128 registers): 360.6 -> 335.0 ms per 256 reconstructions (-7 %).
#                      Round 5, on the rebuilt solver: max-ILP 176.5, default 175.9, ITERATIVE-ILP 174.5 ms (Pauli in-basis 167.8 -> 167.0),
#                      outputs identical -- iterative-ilp for this unit.
#   fbx_pgdb1.hip      lane-per-item single-qubit kernel: no difference (66.4 vs 66.8 ms), default kept.
#   fbx_pgdb_lean.hip  two-wavefronts-per-SIMD kernel (256 registers): -Os instead of -O3 (round 3: 560 instead of 664 B of scratch,
#                      8192 items 73.2 -> 71.2 ms; round 4, after the kernel's re-layout: 66.3 against 71.5 ms -- the -O3 code of the
#                      540-setting instantiation is larger than the 64 KB instruction cache eight wavefronts share) and no machine
#                      LICM: loop-invariant per-lane addresses and masks hoisted out of the outer loop were what still spilled
#                      (70 -> 38 spilled registers, 62.0 -> 61.5 ms per 8192 reconstructions).
#   fbx_pgdb.hip       also -DFBX_JACOBI_TWO_WORKERS: the 16 x 16 Jacobi with two workers per upper block (csrc/fbx_eigh.hpp): B = 1024
#                      12.38 -> 12.10 ms, to convergence 10.25 -> 9.75 ms; the two-waves kernel is 5 % SLOWER with it (spills) and keeps
#                      the full-block form, as does every other unit.
_MAX_ILP = "-mllvm -amdgpu-sched-strategy=max-ilp"
FILE_FLAGS = {"fbx_pgdb.hip": os.environ.get("FBX_PGDB_FLAGS", _MAX_ILP).split(),
              "fbx_pgdb3.hip": os.environ.get("FBX_PGDB3_FLAGS", "-mllvm -amdgpu-sched-strategy=iterative-ilp").split(),
              "fbx_pgdb_lean.hip": os.environ.get("FBX_PGDB_LEAN_FLAGS", "-Os -mllvm -disable-machine-licm").split()}


def file_flags(src):
    return FILE_FLAGS.get(os.path.basename(src), [])


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        [os.path.join(HERE, "..", "include", "fbx.h"), MAP]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=True, profile=False):
    """profile=True builds libfbx_prof.so with per-phase cycle timers (diagnostics only)."""
    global OUT
    out = OUT
    flags = list(FLAGS)
    tag = ""
    if profile:
        flags += os.environ.get("FBX_PROFILE_FLAGS", "").split()
        out = os.path.join(HERE, "libfbx_prof.so")
        flags += ["-DFBX_PHASE_TIMERS", "-DFBX_DIAGNOSTICS"]
        tag = ".prof"
        force = True
    if not force and not stale():
        return out
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "build", os.path.basename(src) + tag + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                [os.path.getmtime(src)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))]
                + [os.path.getmtime(os.path.join(HERE, "..", "include", "fbx.h"))]):
            cmd = [HIPCC] + flags + file_flags(src) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + MAP, "-ldl"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_guard_test(verbose=True):
    """libfbx_cor.so: the product sources with every stored basis that is loaded for Dykstra iteration 1
    damaged on purpose and the rejections counted -- only loaded by tests/test_basis_guard_gpu.py.
    Only fbx_pgdb.hip differs; the other objects are those of libfbx.so."""
    out = os.path.join(HERE, "libfbx_cor.so")
    lib = build(verbose=verbose)
    src = os.path.join(CSRC, "fbx_pgdb.hip")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(lib):
        return out
    obj = os.path.join(HERE, "build", "fbx_pgdb.hip.cor.o")
    cmd = [HIPCC] + FLAGS + file_flags(src) + ["-DFBX_DBG_CORRUPT_BASIS", "-DFBX_DEBUG_REJECT", "-DFBX_DIAGNOSTICS", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    objs = [obj] + [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if s != src]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + MAP, "-ldl"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_barrier_test(verbose=True):
    """libfbx_fullbar.so: the product sources with every workgroup barrier of fbx_pgdb3.hip a full __syncthreads() instead of the
    LDS-only form (-DFBX_FULL_BARRIERS) -- only loaded by tests/test_barriers_gpu.py, which requires bit-identical results."""
    out = os.path.join(HERE, "libfbx_fullbar.so")
    lib = build(verbose=verbose)
    src = os.path.join(CSRC, "fbx_pgdb3.hip")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(lib):
        return out
    obj = os.path.join(HERE, "build", "fbx_pgdb3.hip.fullbar.o")
    cmd = [HIPCC] + FLAGS + file_flags(src) + ["-DFBX_FULL_BARRIERS", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    objs = [obj] + [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if s != src]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + MAP, "-ldl"] + objs + ["-o", out])
    return out


def build_solver_test(verbose=True):
    """libfbx_localrot.so: the product sources with the 3-qubit kernels' 64 x 64 eigensolver in its round-4 form (every thread
    evaluates the rotations it applies, sys_pos layout, one column pair per eigenvector thread: -DFBX_EIGH64_LOCAL_ROTATIONS) -- only
    loaded by tests/test_barriers_gpu.py, which requires the rebuilt solver to reproduce it bit for bit."""
    out = os.path.join(HERE, "libfbx_localrot.so")
    lib = build(verbose=verbose)
    src = os.path.join(CSRC, "fbx_pgdb3.hip")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(lib):
        return out
    obj = os.path.join(HERE, "build", "fbx_pgdb3.hip.localrot.o")
    cmd = [HIPCC] + FLAGS + file_flags(src) + ["-DFBX_EIGH64_LOCAL_ROTATIONS", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    objs = [obj] + [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if s != src]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + MAP, "-ldl"] + objs + ["-o", out])
    return out


def build_variant(name, extra_flags, verbose=True):
    """Experiment builds: libfbx_<name>.so = the product objects with fbx_pgdb.hip and fbx_pgdb_lean.hip recompiled with extra
    -D flags (e.g. `python build.py --variant nosmall -DFBX_NO_SMALL_STEP`).  Not shipped, not loaded by tests."""
    out = os.path.join(HERE, f"libfbx_{name}.so")
    build(verbose=verbose)
    # FBX_VARIANT_SOURCES=fbx_pgdb3.hip,... recompiles other files with the flags (default: the 2-qubit kernel)
    names = os.environ.get("FBX_VARIANT_SOURCES", "fbx_pgdb.hip,fbx_pgdb_lean.hip").split(",")
    srcs = [os.path.join(CSRC, n) for n in names]
    objs = []
    for src in srcs:
        obj = os.path.join(HERE, "build", f"{os.path.basename(src)}.{name}.o")
        cmd = [HIPCC] + FLAGS + file_flags(src) + list(extra_flags) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    objs += [os.path.join(HERE, "build", os.path.basename(s) + ".o") for s in sources() if s not in srcs]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + MAP, "-ldl"] + objs + ["-o", out])
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
        sys.exit(0)
    if "--guard-test" in sys.argv:
        build_guard_test()
        build_barrier_test()
        build_solver_test()
    else:
        build(force="--force" in sys.argv, profile="--profile" in sys.argv)
