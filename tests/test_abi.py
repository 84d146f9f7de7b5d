"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ipx_amd import kkt
    kkt.build_library()                      # hipcc cross-compiles without a GPU
    return kkt.load_library()


def header_functions():
    text = open(os.path.join(ROOT, "include", "ipx_kkt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ipxk_[a-z_0-9]+)\s*\(", text)
    return sorted(set(n for n in names if n != "ipxk_interrupt_fn"))


def test_exports_match_header(lib):
    from ipx_amd import kkt
    names = header_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert sorted(kkt.EXPORTS) == names       # the python binding covers the whole ABI


def test_no_gpu_means_loud_failure(lib):
    """Without a GPU the product refuses to run instead of falling back to the CPU."""
    from ipx_amd import kkt, synth
    if lib.ipxk_device_count() > 0:
        pytest.skip("a GPU is present")
    A = synth.synthetic_lp(20, 40, 4, 1)
    with pytest.raises(kkt.KktError) as e:
        kkt.KktContext(A)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_product_does_not_touch_oracle():
    """ipx_amd/ never imports, links or loads anything under oracle/."""
    pkg = os.path.join(ROOT, "ipx_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.path.basename(dirpath) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in text and "ipx_oracle" not in text and "libipx_ref" not in text, f
    so = os.path.join(pkg, "lib", "libipx_kkt_hip.so")
    if os.path.exists(so):
        import subprocess
        needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
        assert "oracle" not in needed and "ipx_ref" not in needed


def test_no_tracked_binaries():
    """history stays source-only: no tracked file is an ELF object / executable / shared library"""
    import subprocess
    r = subprocess.run(["git", "ls-files"], cwd=ROOT, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("not a git checkout")
    elf = []
    for f in r.stdout.split():
        path = os.path.join(ROOT, f)
        if os.path.isfile(path):
            with open(path, "rb") as fh:
                if fh.read(4) == b"\x7fELF":
                    elf.append(f)
    assert not elf, elf


def test_struct_layout():
    from ipx_amd import kkt
    assert ctypes.sizeof(kkt.Times) == 5 * 8
