"""CPU: the C-ABI library loads and exports every symbol include/semivl_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "semivl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svl_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import __graft_entry__ as g
    g.build()
    import semivl_amd.lib as L
    lib = L.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in semivl_hip.h but not exported"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in semivl_amd/lib.py"
    assert sorted(L.SIGNATURES) == names, set(L.SIGNATURES) ^ set(names)
    assert lib.svl_version() >= 300
    # error convention: bad arguments -> negative status + message, never an exception across the ABI
    rc = lib.svl_fill_f32(None, 0.0, 0, None)
    assert rc == -1 and "svl_fill_f32" in L.last_error()
    rc = lib.svl_gemm_f32(None, None)
    assert rc == -1 and "null desc" in L.last_error()


def test_struct_layouts_match_header_field_order():
    import semivl_amd.lib as L
    src = open(os.path.join(ROOT, "include", "semivl_hip.h")).read()
    for cname, cls in (("svl_gemm_desc", L.GemmDesc), ("svl_conv_geom", L.ConvGeom), ("svl_ce_desc", L.CeDesc), ("svl_ce_up_desc", L.CeUpDesc),
                       ("svl_seqattn_desc", L.SeqAttnDesc), ("svl_operand", L.Operand), ("svl_pgemm_desc", L.PGemmDesc)):
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.sub(r"^(const\s+)?[A-Za-z_0-9]+\s*\*?\s*", "", decl, count=1)
            fields += [re.sub(r"[\s\*]", "", n) for n in names.split(",")]
        assert fields == [f[0] for f in cls._fields_], (cname, fields, [f[0] for f in cls._fields_])
