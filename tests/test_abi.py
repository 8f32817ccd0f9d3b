"""CPU: the C-ABI shared library builds, loads, and exports exactly the symbols include/ace355.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ace355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ace355_[a-z0-9_]+)\s*\(", src)))


def test_build_entry_point():
    import __graft_entry__ as g
    g.build()
    assert os.path.exists(os.path.join(ROOT, "ace-step-1.5-for-windows_amd", "csrc", "libace355.so"))


def test_library_exports_every_declared_symbol():
    from ace355 import native
    lib = native.lib()
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ace355.h but not exported"
    assert set(native.SIGNATURES) == set(syms), set(native.SIGNATURES) ^ set(syms)
    assert lib.ace355_version() >= 100


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct crossing the ABI, taken from the HEADER by the C compiler, vs the ctypes mirrors."""
    import subprocess
    from ace355 import native
    structs = {"ace355_dit_config": native.DitConfigC, "ace355_vae_config": native.VaeConfigC,
               "ace355_sample_params": native.SampleParamsC, "ace355_cond_config": native.CondConfigC,
               "ace355_detok_config": native.DetokConfigC}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ace355.h"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _t in ct._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, ct in structs.items():
        assert int(got[cname]) == ctypes.sizeof(ct), cname
        for fname, _t in ct._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(ct, fname).offset, f"{cname}.{fname}"
    assert ctypes.sizeof(native.SampleParamsC) == 96


def test_missing_library_fails_loudly(monkeypatch):
    from ace355 import native
    monkeypatch.setattr(native, "_LIB", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libace355.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.lib()


def test_errors_do_not_abort_without_gpu():
    """On a box without a GPU a create call must come back with an error code + message, not crash."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_dit_gpu.py::test_errors_are_reported_not_fatal")
    from ace355 import native
    lib = native.lib()
    cfg = native.DitConfigC(256, 768, 2, 2, 1, 128, 128, 2, 192, 64, 1e-6, 1e6, 1)
    h = ctypes.c_void_p()
    rc = lib.ace355_dit_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc != 0 and "HIP error" in native.last_error()
    bad = native.DitConfigC(256, 768, 2, 2, 1, 64, 128, 2, 192, 64, 1e-6, 1e6, 1)  # head_dim 64 unsupported
    rc = lib.ace355_dit_create(ctypes.byref(bad), ctypes.byref(h))
    assert rc == 1 and "head_dim" in native.last_error()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "ace-step-1.5-for-windows_amd")
    for dirpath, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
