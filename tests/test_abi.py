"""The C-ABI library loads and exports every symbol include/gorp_hip.h declares
(no compute calls: this runs without a GPU)."""
import os
import re

from gorp_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "gorp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = declared_functions()
    assert len(names) >= 14
    L = N.lib()
    for n in names:
        assert getattr(L, n) is not None, n
    assert sorted(N.SYMBOLS) == names


def test_library_is_in_tree():
    assert os.path.dirname(N.LIB_PATH) == os.path.join(ROOT, "gorp_amd")


def test_error_string_and_arg_checks():
    import ctypes as C
    L = N.lib()
    h = C.c_void_p()
    assert L.gx_create_from_patterns(None, None, 0, N.GX_CREATE_HOST_ONLY, C.byref(h)) == N.GX_E_ARG
    assert "bad argument" in N.last_error()
    assert L.gx_create_from_blob(b"xxxxxxxxxxxxxxxx", 16, N.GX_CREATE_HOST_ONLY, C.byref(h)) == N.GX_E_ARG
    assert "blob" in N.last_error()
    assert L.gx_num_extractions(None) == 0 and L.gx_max_groups(None) == 0 and L.gx_blob_size(None) == 0
    L.gx_destroy(None)


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "gorp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), os.path.join(dirpath, f)


def test_release_library_reads_no_environment_variables():
    """the product library carries no developer hooks.  No GX_* environment name is in the shipped
    binary (they live behind -DGX_DEV in libgorp_hip_dev.so) and `getenv` is not among its imports."""
    import re
    import subprocess
    from gorp_amd import build as gbuild
    blob = open(gbuild.LIB, "rb").read()
    names = set(re.findall(rb"GX_(?:DEV|REC|DEBUG|BENCH|PROF)[A-Z0-9_]*", blob))
    assert not names, names
    syms = subprocess.run(["nm", "-D", "--undefined-only", gbuild.LIB], capture_output=True, text=True).stdout
    assert "getenv" not in syms


def test_reference_jvm_baseline_driver_is_source_only_and_says_so_here():
    """tools/RefBench.java is the reference-as-baseline driver: it uses nothing but the reference's public API
    (README.md:63-79), ships as source, and bench.py reports "unavailable" where there is no JDK -- as in this image."""
    import shutil
    import sys
    sys.path.insert(0, ROOT)
    import bench
    src = open(os.path.join(ROOT, "tools", "RefBench.java")).read()
    imports = [ln.split()[1].rstrip(";") for ln in src.splitlines() if ln.startswith("import com.salesforce")]
    assert sorted(imports) == ["com.salesforce.gorp.DefinitionReader", "com.salesforce.gorp.ExtractionException",
                               "com.salesforce.gorp.ExtractionResult", "com.salesforce.gorp.Gorp"]
    assert "DefinitionReader.reader(new File(" in src and ".read()" in src and "gorp.extract(" in src
    if not (shutil.which("java") and shutil.which("javac") and os.environ.get("GORP_REFERENCE_CLASSPATH")):
        assert bench.reference_jvm_baseline("extract a {\n template x\n}\n", [b"x"], 1).startswith("unavailable (no ")


def test_unpack_rows_on_the_host():
    """gorp.unpack_rows: u16 rows (int16 id, 0xFFFF = unset) and u8 rows (int8 id, 0xFF = unset) back to int32."""
    import numpy as np
    from gorp_amd.gorp import unpack_rows
    r16 = np.array([[1, 0, 65534, 0xFFFF, 0xFFFF], [0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF], [0xFFFE - 5, 3, 4, 5, 6]], np.uint16)
    m, c = unpack_rows(r16)
    assert m.tolist() == [1, -1, -7] and c.tolist() == [[0, 65534, -1, -1], [-1, -1, -1, -1], [3, 4, 5, 6]]
    m, c = unpack_rows(r16.view(np.int16))
    assert m.tolist() == [1, -1, -7] and c[0].tolist() == [0, 65534, -1, -1]
    r8 = np.array([[127, 0, 254, 0xFF, 0xFF], [0xFF, 0xFF, 0xFF, 0xFF, 0xFF], [0x80, 1, 2, 3, 4]], np.uint8)
    m, c = unpack_rows(r8)
    assert m.tolist() == [127, -1, -128] and c.tolist() == [[0, 254, -1, -1], [-1, -1, -1, -1], [1, 2, 3, 4]]
