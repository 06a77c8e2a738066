"""Registers, scratch and LDS of every kernel in one object of the build (the numbers that decide occupancy and tell
spills): python tools/kernel_regs.py conv [filter]   -- reads yolo_v1_amd/lib/obj/<name>.o (hipcc -c output), unbundles the
gfx950 code object and prints the AMDGPU metadata per kernel.  No GPU needed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_table(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
        rows, cur = [], {}
        for line in notes.splitlines():
            m = re.match(r"\s+(- )?\.(\w+):\s+(.*)$", line)
            if not m:
                continue
            if m.group(1) and cur.get("name"):
                rows.append(cur)
                cur = {}
            elif m.group(1):
                cur = {}
            cur[m.group(2)] = m.group(3).strip()
        if cur.get("name"):
            rows.append(cur)
        names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), text=True,
                               capture_output=True).stdout.splitlines()
        for r, n in zip(rows, names):
            r["demangled"] = n.replace("(anonymous namespace)::", "").replace("void ", "")
        return [r for r in rows if "vgpr_count" in r]


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "conv"
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = kernel_table(os.path.join(ROOT, "yolo_v1_amd", "lib", "obj", name + ".o"))
    print("%-64s %5s %5s %5s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds"))
    for r in sorted(rows, key=lambda r: r["demangled"]):
        if flt and flt not in r["demangled"]:
            continue
        print("%-64s %5s %5s %5s %8s %8s" % (r["demangled"].split("(")[0][:64], r.get("vgpr_count"), r.get("agpr_count", "0"),
                                             r.get("sgpr_count"), r.get("private_segment_fixed_size", "0"),
                                             r.get("group_segment_fixed_size", "0")))
