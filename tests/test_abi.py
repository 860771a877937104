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


def test_resize_fused_loss_geometry_query_is_host_only():
    """svl_ce_up_num_blocks runs the kernels' own tile functions on the host (no GPU): the training geometries are taken,
    reductions / ratios above 4.5 / class counts whose staged tile would not fit are refused (callers then resize)."""
    import semivl_amd.lib as L
    lib = L.load()
    q = lambda B, N, h, w, H, W, a: lib.svl_ce_up_num_blocks(B, N, h, w, H, W, a)
    for align in (0, 1):
        assert q(16, 21, 128, 128, 512, 512, align) == 16 * 16 * 16          # VOC / COCO / ADE crops
        assert q(8, 19, 204, 204, 801, 801, align) == 8 * 26 * 26            # Cityscapes 801^2 (51 patches per side)
        assert q(2, 5, 8, 8, 32, 32, align) == 2                             # the fixtures' 32^2 crops
        assert q(1, 150, 24, 40, 96, 160, align) == 3 * 5
        assert q(2, 21, 64, 64, 512, 512, align) == -1                       # ratio 8
        assert q(2, 21, 128, 128, 64, 64, align) == -1                       # a reduction
        assert q(2, 161, 128, 128, 512, 512, align) == -1                    # N x 11 x 11 floats + the pixel state > 160 KB
        assert q(0, 21, 128, 128, 512, 512, align) == -1
