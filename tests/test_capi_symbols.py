"""The C-ABI library loads without a GPU, exports every symbol include/segdino3d_hip.h declares, and
the ctypes signature table of segdino3d_amd/_lib.py agrees with the header prototype by prototype."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prototypes(header="segdino3d_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(sd3d_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        arglist = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, arglist)
    return protos


def _ctype_of(decl: str):
    if "*" in decl:
        return C.c_void_p
    base = decl.rsplit(" ", 1)[0].replace("const ", "").strip()
    return {"int": C.c_int, "int64_t": C.c_int64, "size_t": C.c_size_t, "float": C.c_float}[base]


def test_header_and_binding_agree():
    from segdino3d_amd import _lib
    protos = _prototypes()
    assert len(protos) >= 30
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (ret, args) in protos.items():
        res, argtypes = _lib.SIGNATURES[name]
        want = [_ctype_of(a) for a in args]
        assert list(argtypes) == want, f"{name}: binding {argtypes} != header {want}"
        assert res == {"int": C.c_int, "size_t": C.c_size_t, "const char*": C.c_char_p}[ret], name


def test_library_loads_and_exports_every_symbol():
    from segdino3d_amd import _lib
    lib = _lib.load()
    for name in _prototypes():
        assert hasattr(lib, name), name
    assert lib.sd3d_abi_version() == _lib.ABI_VERSION
    assert lib.sd3d_selftest_host() == 0, lib.sd3d_last_error()
    assert lib.sd3d_sort_ws_bytes(150000) > 0 and lib.sd3d_unique_ws_bytes(1000) > 0


def test_experimental_library_is_separate_from_the_product_library():
    """Kernels that are not on the product path live in their own library with their own header: the product library exports none
    of them, nothing under segdino3d_amd/ (but the binding module itself) imports the binding, and header == binding == exports."""
    import subprocess
    from segdino3d_amd import _lib, experimental
    protos = _prototypes("segdino3d_hip_experimental.h")
    assert set(protos) == set(experimental.SIGNATURES) and not (set(protos) & set(_lib.SIGNATURES))
    for name, (ret, args) in protos.items():
        res, argtypes = experimental.SIGNATURES[name]
        assert list(argtypes) == [_ctype_of(a) for a in args], name
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in protos:
        assert f" {name}\n" not in exported, f"{name} must not be part of the product library"
    pkg = os.path.join(ROOT, "segdino3d_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py") and fn != "experimental.py":
            assert "experimental" not in open(os.path.join(pkg, fn)).read(), f"segdino3d_amd/{fn} must not depend on the experimental library"
    if experimental.available():
        lib = experimental.load()
        for name in protos:
            assert hasattr(lib, name), name


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from segdino3d_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 32), torch.zeros(8, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.scene_stats(torch.zeros(10, 6))


def test_entry_scripts_compile():
    """bench.py / __graft_entry__.py / tools are run on the GPU box only: at least make sure they parse here."""
    import ast
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + glob.glob(os.path.join(root, "tools", "*.py")):
        with open(path) as f:
            ast.parse(f.read(), filename=path)
