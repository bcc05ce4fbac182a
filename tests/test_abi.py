"""CPU: the C-ABI libraries load and export every symbol the headers declare
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def _exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_engine_library_exports_every_declared_symbol(native):
    decl = _declared("grpath.h", "grp_") + _declared("grpath_synth.h", "grp_synth_") + _declared("grpath_ingest.h", "grp_fastq_")
    assert len(decl) >= 24
    exp = _exported(native.LIB_PATH)
    missing = [d for d in decl if d not in exp]
    assert not missing, missing
    # and the binding types every one of them
    assert set(decl) == set(native.SIGNATURES), set(decl) ^ set(native.SIGNATURES)
    lib = native.load()
    for name in decl:
        assert getattr(lib, name) is not None
    # ... and nothing else (round 6): the product library exports exactly what the headers declare — the measurement-only
    # prototypes of include/grpath_dev.h exist in developer builds alone (make DEV=1), the frozen commit loop is gone
    extra = {e for e in exp if e.startswith("grp_")} - set(decl)
    if lib.grp_dev_hooks():
        assert extra == set(_declared("grpath_dev.h", "grp_")), extra
    else:
        assert not extra, extra
        assert not any("commit_loop" in e or "pshard" in e for e in exp)


def test_host_library_exports_every_declared_symbol(native):
    from goldrush_amd import host

    decl = [d for d in _declared("grpath_host.h", "gr_") if not d.endswith("_fn")]
    exp = _exported(host.LIB_PATH)
    missing = [d for d in decl if d not in exp]
    assert not missing, missing
    assert set(decl) == set(host.SIGNATURES), set(decl) ^ set(host.SIGNATURES)
    host.load()


def test_struct_layouts_match_headers(native):
    from goldrush_amd import host

    assert ctypes.sizeof(native.grp_params) == 40
    assert native.tile_summary_dtype.itemsize == 24 and native.id_count_dtype.itemsize == 8
    assert ctypes.sizeof(host.gr_read_decision) == 32 == host.decision_dtype.itemsize
    assert ctypes.sizeof(host.gr_commit) == 48
    assert ctypes.sizeof(host.grp_engine_vt) == 48 * ctypes.sizeof(ctypes.c_void_p)


def test_no_gpu_means_loud_failure_not_fallback(native):
    """Without a HIP device grp_create must fail (GRP_ERR_NO_DEVICE); on a GPU box
    it succeeds.  Either way nothing is computed on the CPU."""
    import pytest

    try:
        eng = native.Engine(22, 3, 1000, 1 << 20, ["1" * 22, "1" * 23, "1" * 24])
    except native.GrpError as e:
        assert e.code in (native.GRP_ERR_NO_DEVICE, native.GRP_ERR_HIP)
    else:
        eng.close()


def test_cli_links_the_engine(native):
    from goldrush_amd import host

    out = subprocess.run(["ldd", host.CLI_PATH], capture_output=True, text=True).stdout
    assert "libgrpath_hip.so" in out and "libgrpath_host.so" in out
    # product sources never reference the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "goldrush_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "orc_" not in txt and "import orc" not in txt and "liboracle" not in txt, os.path.join(dirpath, f)
