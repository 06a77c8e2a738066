"""Builds libyv1.so (the C-ABI HIP library) for gfx950 with hipcc, in-tree.

``python -m yolo_v1_amd.build`` or ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libyv1.so")
ARCH = "gfx950"

# file -> extra flags.  The loss / decoder / NMS must reproduce the reference's fp32 op
# sequence (bit-exact kept-box indices), so FMA contraction is off for them.
SOURCES = {
    "loss.hip": ["-ffp-contract=off"],
    "decode_nms.hip": ["-ffp-contract=off"],
    "encode.hip": ["-ffp-contract=off"],
    "conv.hip": [],
    "conv_fp8.hip": [],
    "wgrad.hip": [],
    "elementwise.hip": [],
    "optim.hip": [],
    "cfglog.hip": [],
    "bn3alg.hip": [],
    "bn_deferred.hip": [],
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp(path, flags):
    h = hashlib.sha1()
    h.update(open(path, "rb").read())
    for inc in sorted(os.listdir(CSRC)):
        if inc.endswith(".h"):
            h.update(open(os.path.join(CSRC, inc), "rb").read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def build(verbose=True, force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    objs, rebuilt = [], False
    procs = []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
                 "-I" + CSRC, "-I" + os.path.join(os.path.dirname(HERE), "include")] + extra
        stamp = _stamp(path, flags)
        sfile = obj + ".stamp"
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.exists(sfile) and open(sfile).read() == stamp):
            continue
        cmd = [hipcc, "-c", path, "-o", obj] + flags
        if verbose:
            print("[yv1 build]", " ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), sfile, stamp, src))
        rebuilt = True
    for p, sfile, stamp, src in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
        open(sfile, "w").write(stamp)
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        if verbose:
            print("[yv1 build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
