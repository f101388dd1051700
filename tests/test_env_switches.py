"""the product carries only the environment switches DESIGN.md section 5 documents -- the settings
rounds 2-5 measured and rejected are gone (or live in the -DGSTVD_DIAG build that only tools/ load).  The documented list
must equal (a) the GSTVD_* strings in the shipped library and (b) the GSTVD_* names the package's os.environ reads use."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REJECTED = {"GSTVD_GROUP_ORDER", "GSTVD_GEMM_ST", "GSTVD_GEMM_PP", "GSTVD_GEMM_VARIANT", "GSTVD_KV_SIDE", "GSTVD_GEMM128_N96",
            "GSTVD_GEMM256_NIU"}


def _documented():
    txt = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = txt.split("<!-- env:begin -->")[1].split("<!-- env:end -->")[0]
    rows = [l for l in block.splitlines() if l.startswith("| `GSTVD_")]
    return {re.match(r"\| `(GSTVD_[A-Z0-9_]+)`", l).group(1): l for l in rows}


def _library_strings():
    from gst_visdial_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(m.decode() for m in re.findall(rb"GSTVD_[A-Z0-9_]+", blob))
    return {n for n in names if not n.startswith(("GSTVD_E_", "GSTVD_EPI_", "GSTVD_LN_"))}     # (enum names never reach the binary; belt and braces)


def _package_reads():
    names = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "gst_visdial_amd", "*.py"))):
        for line in open(f):
            if "environ" in line or "getenv" in line:
                for n in re.findall(r"GSTVD_[A-Z0-9_]+", line):
                    names.setdefault(n, os.path.basename(f))
    return names


def test_documented_switches_are_exactly_the_ones_the_product_reads():
    doc = _documented()
    lib, pkg = _library_strings(), _package_reads()
    assert lib | set(pkg) == set(doc), (sorted((lib | set(pkg)) ^ set(doc)))
    assert not (REJECTED & (lib | set(pkg)))
    for n, f in pkg.items():                       # the table names the file that reads the switch
        assert f in doc[n], (n, f)
    for n in lib:
        assert "csrc/" in doc[n], n


def test_diag_selectors_are_not_in_the_product_library():
    assert not {n for n in _library_strings() if n.startswith("GSTVD_DIAG")}
