#!/usr/bin/env python3
"""Developer tool: registers / spills / scratch of every kernel in libgrpath_hip.so
(the gfx950 code object is cut out of the .hip_fatbin section; no GPU needed).
usage: tools/dev/kernel_resources.py [substring ...]   (default: all kernels)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.environ.get("GRP_LIB", os.path.join(ROOT, "goldrush_amd", "lib", "libgrpath_hip.so"))


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    rows = []
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
        rows.append((re.search(r"\.name:\s+(\S+)", k).group(1), g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"),
                     g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-72s %5s %6s %5s %6s %8s %6s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "scratch", "lds"))
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        n = re.sub(r"\(.*", "", n)
        if want and not any(w in n for w in want):
            continue
        print("%-72s %5d %6d %5d %6d %8d %6d" % (n[:72], r[1], r[2], r[3], r[4], r[5], r[6]))
    print("%d kernels" % len(rows))


if __name__ == "__main__":
    main()
