"""The C-ABI shared library loads on a CPU-only box and exports every symbol include/minppo_hip.h declares;
the ctypes table covers the header.  No compute calls are made here."""
import ctypes as C
import re
from pathlib import Path

import pytest

from minppo_amd import _native as nat

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "minppo_hip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mppo_[a-z0-9_]+)\s*\(", txt)))


def test_header_matches_ctypes_table():
    assert _declared() == sorted(nat.SIGNATURES)


def test_hip_library_exports_every_declared_symbol():
    if not nat.HIP_LIB_PATH.exists():
        from minppo_amd import build

        build.build(verbose=False)
    dll = C.CDLL(str(nat.HIP_LIB_PATH))
    for name in _declared():
        assert hasattr(dll, name), f"libminppo_hip.so does not export {name}"
    dll.mppo_abi_version.restype = C.c_int32
    assert dll.mppo_abi_version() == 1
    dll.mppo_last_error.restype = C.c_char_p
    assert dll.mppo_last_error() == b""


def test_product_loader_fails_loudly_without_the_library(monkeypatch, tmp_path):
    monkeypatch.setattr(nat, "HIP_LIB_PATH", tmp_path / "libminppo_hip.so")
    monkeypatch.setattr(nat, "_LIB", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        nat.load()


def test_product_package_never_imports_the_oracle():
    for f in (ROOT / "minppo_amd").rglob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
        assert "libminppo_emu" not in src, f"{f} references the emulator build"


def test_header_is_plain_c_and_cxx(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 and as C++17 with nothing but the standard headers."""
    import shutil
    import subprocess

    hdr = str(Path(__file__).resolve().parents[1] / "include" / "minppo_hip.h")
    for cc, args in (("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++17", "-x", "c++"])):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not installed")
        r = subprocess.run([cc, "-fsyntax-only", "-Wall", "-Werror", *args, hdr], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # and a C translation unit that calls through it links against the library's symbol names (no C++ mangling)
    src = tmp_path / "use.c"
    src.write_text('#include "minppo_hip.h"\nint main(void) { mppo_net_t n = {225, 228, 10, 256, 1, 0}; return mppo_param_count(&n) == 250140 ? 0 : 1; /* 250133 parameters + 7 alignment words */ }\n')
    r = subprocess.run(["gcc", "-std=c99", "-c", "-I", str(Path(hdr).parent), str(src), "-o", str(tmp_path / "use.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    nm = subprocess.run(["nm", "-u", str(tmp_path / "use.o")], capture_output=True, text=True).stdout
    assert "mppo_param_count" in nm and "_Z" not in nm
