"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/csp_minsnap.h
declares, validates arguments, sizes workspaces -- and refuses to compute without a device."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in ("csp_minsnap.h", "csp_geo.h", "csp_alt.h", "csp_bezier.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        syms |= set(re.findall(r"\b(csp_(?:minsnap|geo|alt|bezier)_[a-z0-9_]+)\s*\(", hdr))
    return sorted(syms)


def test_every_declared_symbol_is_exported(csp):
    syms = _declared_symbols()
    assert set(syms) == set(csp.EXPORTED_SYMBOLS), syms
    lib = ctypes.CDLL(csp.LIB_PATH)
    for s in syms:
        assert getattr(lib, s) is not None
    out = subprocess.check_output(["nm", "-D", "--defined-only", csp.LIB_PATH]).decode()
    for s in syms:
        assert re.search(r"\bT %s\b" % s, out), s


def test_library_contains_gfx950_code_object(csp):
    blob = open(csp.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"minsnap_fixed_kernel" in blob and b"minsnap_generic_kernel" in blob


def test_desc_struct_layout_matches_header(csp):
    # 4+4+4+4+8+8+4+4+8+8+8+4+4+4+4 = 80 bytes on LP64
    assert ctypes.sizeof(csp.Desc) == 80
    assert csp.Desc.batch.offset == 16 and csp.Desc.seg_offsets.offset == 24
    assert csp.Desc.path_weight.offset == 40 and csp.Desc.mem_space.offset == 64


def test_validation_and_dispatch_names(csp):
    d = csp.make_desc(4, 65536, 16)
    assert csp.kernel_name(d) == "fixed_o4_s16_f64"
    assert csp.workspace_bytes(d) == 0
    d = csp.make_desc(4, 4096, 8)
    assert csp.kernel_name(d) == "fixed_o4_s8_f64"
    d = csp.make_desc(4, 65536, 16, flags=csp.FLAG_FORCE_GENERIC)
    assert csp.kernel_name(d) == "generic_o4_f64"
    assert csp.workspace_bytes(d) == 15 * 18 * 65536 * 8
    assert csp.kernel_name(csp.make_desc(3, 100, 16)) == "fixed_o3_s16_f64"
    assert csp.kernel_name(csp.make_desc(2, 100, 6)) == "fixed_o2_s6_f64"
    assert csp.kernel_name(csp.make_desc(5, 100, 8)) == "fixed_o5_s8_f64"
    assert csp.kernel_name(csp.make_desc(5, 100, 10)) == "chunked_o5_f64_l4"     # order 5 buckets stop at S = 8
    assert csp.kernel_name(csp.make_desc(5, 100, 10, flags=csp.FLAG_FORCE_GENERIC)) == "generic_o5_f64"
    assert csp.kernel_name(csp.make_desc(4, 100, 7)) == "fixed_o4_s7_f64"        # odd S: 4 + 3 segments
    assert csp.kernel_name(csp.make_desc(4, 100, 17)) == "chunked_o4_f64_l8"     # long: 8 lanes x <= 4 segments
    assert csp.kernel_name(csp.make_desc(4, 100, 64)) == "chunked_o4_f64_l16"
    assert csp.kernel_name(csp.make_desc(4, 100, 256)) == "chunked_o4_f64_l64"
    assert csp.workspace_bytes(csp.make_desc(4, 100, 256)) == 0
    assert csp.kernel_name(csp.make_desc(4, 100, 257)) == "span_o4_f64_l32"      # very long: lanes x <= 16 segments
    assert csp.kernel_name(csp.make_desc(4, 100, 1024)) == "span_o4_f64_l64"
    assert csp.workspace_bytes(csp.make_desc(4, 100, 1024)) == 0
    assert csp.kernel_name(csp.make_desc(4, 100, 1025)) == "generic_o4_f64"
    assert csp.kernel_name(csp.make_desc(4, 100, 64, flags=csp.FLAG_SPAN)) == "span_o4_f64_l4"
    assert csp.kernel_name(csp.make_desc(5, 40000, 40)) == "span_o5_f64_l4"      # order 5, big batch: span kernel from 17 segments
    assert csp.kernel_name(csp.make_desc(5, 100, 40)) == "chunked_o5_f64_l16"
    assert csp.kernel_name(csp.make_desc(5, 100, 16)) == "chunked_o5_f64_l4"
    assert csp.kernel_name(csp.make_desc(4, 100, 16, flags=csp.FLAG_SPAN)) == "fixed_o4_s16_f64"
    assert csp.kernel_name(csp.make_desc(4, 100, 1)) == "chunked_o4_f64_l1"
    assert csp.kernel_name(csp.make_desc(1, 100, 5)) == "generic_o1_f64"         # order 1 has no free derivative
    assert csp.kernel_name(csp.make_desc(3, 100, 16, flags=csp.FLAG_SEGMENT_MAJOR)) == "generic_o3_f64"
    d = csp.make_desc(4, 10, 16, path_weight=1e-3)
    assert csp.kernel_name(d) == "fixedpath_o4_s16_f64"       # pre-solve + t* pick + penalised solve in registers
    assert csp.workspace_bytes(d) == 0
    d = csp.make_desc(4, 10, 16, path_weight=1e-3, flags=csp.FLAG_FORCE_GENERIC)
    assert csp.kernel_name(d) == "generic_o4_f64"
    assert csp.workspace_bytes(d) >= 15 * 18 * 10 * 8 + 16 * 10 * 4
    assert csp.kernel_name(csp.make_desc(5, 10, 8, path_weight=1e-3)) == "generic_o5_f64"
    assert csp.kernel_name(csp.make_desc(4, 10, 17, path_weight=1e-3)) == "generic_o4_f64"
    d = csp.make_desc(3, 10, 7, dtype=csp.DTYPE_F32)
    assert csp.kernel_name(d) == "chunked_o3_f32io_f64_l2"
    d = csp.make_desc(3, 10, 7, dtype=csp.DTYPE_F32, flags=csp.FLAG_FORCE_GENERIC)
    assert csp.kernel_name(d) == "generic_o3_f32io_f64"
    d = csp.make_desc(3, 10, 7, dtype=csp.DTYPE_F32, flags=csp.FLAG_F32_ARITH)
    assert csp.kernel_name(d) == "generic_o3_f32"
    assert csp.kernel_name(csp.make_desc(6, 1, 4)) is None          # unsupported order
    assert csp.kernel_name(csp.make_desc(4, 1, 0)) is None          # ragged without offsets
    bad = csp.make_desc(4, 1, 4)
    bad.abi_version = 99
    assert csp.kernel_name(bad) is None


def test_no_cpu_fallback(csp):
    """Without a gfx950 device every compute entry point must fail loudly, never compute."""
    if csp.device_count() > 0:
        pytest.skip("a device is present; this test is for the CPU-only container")
    wp = np.zeros((2, 5, 3))
    tm = np.ones((2, 4))
    with pytest.raises(csp.CspError) as e:
        csp.solve_batch(wp, tm, order=4)
    assert e.value.code == -5
    with pytest.raises(csp.CspError) as e:
        csp.time_alloc_batch(wp, 5.0, 0.1)
    assert e.value.code == -5
    with pytest.raises(csp.CspError) as e:
        csp.solve_batch(wp, tm, order=4, ngpu=2)     # single-process multi-GPU entry
    assert e.value.code == -5


def test_product_code_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under cs-pathplan_amd/ may import or link it."""
    pkg = os.path.join(ROOT, "cs-pathplan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("#", "//", "*", '"""')) and "include" not in s:
                        continue
                    assert not re.search(r"(import\s+oracle|from\s+oracle|libcsp_oracle|dense_oracle\.c\b.*include|numpy_ref)", s) \
                        or "generated" in s.lower() or "GENERATED" in s, (f, s)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libcsp_minsnap.so")]).decode()
    assert "oracle" not in out


def test_host_shim_compiles_against_the_cabi():
    import importlib.util
    spec = importlib.util.spec_from_file_location("csp_build", os.path.join(ROOT, "cs-pathplan_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_host_check()
    rc = subprocess.call([exe, "kat"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert rc in (0, 3)   # 3 = no device (CPU container): the shim refuses, it does not fall back
