"""CPU: the C-ABI library loads and exports every symbol include/yv1.h declares (no compute calls),
and the ctypes table in yolo_v1_amd/_lib.py lists exactly those symbols."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "yv1.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yv1_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_ctypes_table_and_library_exports():
    from yolo_v1_amd import _lib, build
    syms = _header_symbols()
    assert len(syms) >= 30
    assert syms == sorted(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)          # hipcc cross-compiles without a GPU
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), "libyv1.so does not export " + s
    # argument-count sanity against the header prototypes
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "yv1.h")).read(), flags=re.S)
    for s in syms:
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % s, txt, flags=re.S)
        nargs = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert nargs == len(_lib.SIGNATURES[s][1]), s


def _proto_args(arglist):
    return [re.sub(r"\s+", " ", a).strip() for a in arglist.split(",") if a.strip() and a.strip() != "void"]


def test_argument_types_agree_between_header_ctypes_table_and_definitions():
    """Type by type, not only by count: include/yv1.h against yolo_v1_amd/_lib.py's ctypes table and against every
    extern "C" definition in csrc/*.hip (the sources do not include the public header, so the compiler checks neither)."""
    import glob
    from yolo_v1_amd import _lib
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "yv1.h")).read(), flags=re.S)
    protos = {n: _proto_args(a) for _, n, a in re.findall(r"^(int|size_t)\s+(yv1_\w+)\s*\((.*?)\)\s*;", txt, flags=re.S | re.M)}
    assert sorted(protos) == sorted(_lib.SIGNATURES)
    scalar = {"int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong, "size_t": ctypes.c_size_t,
              "unsigned": ctypes.c_uint, "double": ctypes.c_double}

    def ctype(a):
        if "*" in a or "yv1_stream_t" in a or "hipStream_t" in a:
            return ctypes.c_void_p
        return scalar[a.rsplit(" ", 1)[0].replace("const ", "").strip()]

    def kind(a):                               # pointer-ness + scalar type, names ignored
        if "*" in a or "yv1_stream_t" in a or "hipStream_t" in a:
            return "ptr"
        return a.rsplit(" ", 1)[0].replace("const ", "").strip()

    for name, args in protos.items():
        want = [ctype(a) for a in args]
        got = [ctypes.c_void_p if t is ctypes.c_char_p else t for t in _lib.SIGNATURES[name][1]]
        assert want == got, (name, [(i, args[i]) for i, (w, g) in enumerate(zip(want, got)) if w != g])
    seen = set()
    for f in glob.glob(os.path.join(ROOT, "yolo_v1_amd", "csrc", "*.hip")):
        src = re.sub(r"//[^\n]*", "", open(f).read())
        for _, name, a in re.findall(r'extern "C"\s+(int|size_t)\s+(yv1_\w+)\s*\((.*?)\)\s*\{', src, flags=re.S):
            seen.add(name)
            assert [kind(x) for x in _proto_args(a)] == [kind(x) for x in protos[name]], (os.path.basename(f), name)
    assert seen == set(protos), set(protos) ^ seen


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    from yolo_v1_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libyv1.so")
    with pytest.raises(_lib.Yv1Error):
        _lib.lib()


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under yolo_v1_amd/ may import it
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "yolo_v1_amd")):
        for fn in fns:
            if fn.endswith(".py"):
                src = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_cpu_tensors_are_rejected_not_silently_computed():
    import pytest
    import torch
    from yolo_v1_amd import _lib
    from yolo_v1_amd.utils import utils as yu
    from yolo_v1_amd.v1Loss import YOLOLossV1
    with pytest.raises(_lib.Yv1Error):
        yu.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)
    with pytest.raises(_lib.Yv1Error):
        YOLOLossV1(1, 7, 2, 20, _quiet=True)(torch.zeros(1, 7, 7, 30), torch.zeros(1, 7, 7, 30))


def test_header_is_plain_c(tmp_path):
    """include/yv1.h must be consumable by a C compiler (the boundary is a C ABI: no C++ types, no torch types) and a C
    translation unit must be able to take the address of every declared entry point."""
    import re
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc in this image")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "yv1.h")).read()
    names = re.findall(r"^(?:int|size_t)\s+(yv1_\w+)\s*\(", hdr, re.M)
    assert len(names) >= 40
    src = tmp_path / "use_header.c"
    src.write_text('#include "yv1.h"\n#include <stddef.h>\nvoid* table[] = {\n' +
                   "".join("  (void*)%s,\n" % n for n in names) + "};\nint count(void) { return (int)(sizeof table / sizeof table[0]); }\n")
    obj = tmp_path / "use_header.o"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-Wno-pedantic", "-c", str(src), "-o", str(obj),
                           "-I" + os.path.join(root, "include")])
    assert obj.exists()
