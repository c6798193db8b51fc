#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use of a built libfbx*.so (metadata notes of its gfx950 code object).

    python scripts/kernel_resources.py [forest-benchmarking_amd/libfbx.so] [name regex]
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib):
    with tempfile.TemporaryDirectory() as tmp:
        fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob, notes = open(fat, "rb").read(), ""
        magic = b"__CLANG_OFFLOAD_BUNDLE__"                 # one bundle per translation unit
        starts = [m.start() for m in re.finditer(magic, blob)] + [len(blob)]
        for k in range(len(starts) - 1):
            part = os.path.join(tmp, f"part{k}.bin")
            open(part, "wb").write(blob[starts[k]:starts[k + 1]])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def g(k):
            mm = re.search(r"\." + k + r":\s+(\S+)", blk)
            return mm.group(1) if mm else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        out.append({"name": re.sub(r"\(.*", "", name), "vgpr": g("vgpr_count"), "agpr": blk.split()[0], "sgpr": g("sgpr_count"),
                    "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                    "vgpr_spills": g("vgpr_spill_count")})
    return out


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else "forest-benchmarking_amd/libfbx.so"
    flt = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    for r in resources(lib):
        if flt.search(r["name"]):
            print(f"{r['name'][:72]:72s} vgpr {r['vgpr']:>4s} agpr {r['agpr']:>3s} sgpr {r['sgpr']:>4s} "
                  f"scratch {r['scratch']:>5s} B  static-lds {r['lds']:>6s}  spilled vgprs {r['vgpr_spills']:>4s}")
