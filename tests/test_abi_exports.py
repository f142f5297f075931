"""The C-ABI library loads without a GPU and exports every symbol include/dsv2_hip.h declares."""
import ctypes as C
import os
import re

import dsvabi as A


def declared_functions():
    text = open(os.path.join(A.ROOT, "include", "dsv2_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b(dsv_[a-z0-9_]+|dsv2hip_[a-z0-9_]+)\s*\(", text, flags=re.M):
        names.add(m.group(1))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(A.HIP_SO)
    names = declared_functions()
    assert len(names) > 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/dsv2_hip.h but not exported: %s" % missing
    assert C.c_void_p.in_dll(lib, "dsv_lvlname")


def test_struct_layouts_match_reference_abi():
    # sizes the reference's own CLI relies on (it embeds DSV_ENCODER / DSV_DECODER by value)
    assert C.sizeof(A.MV) == 16
    assert C.sizeof(A.META) == 36
    assert C.sizeof(A.PLANE) == 32
    assert C.sizeof(A.FRAME) == 8 + 3 * 32 + 5 * 4 + 4
    lib = C.CDLL(A.HIP_SO)
    lib.dsv2hip_version.restype = C.c_char_p
    assert b"dsv2hip" in lib.dsv2hip_version()


def test_host_helpers_work_without_gpu():
    lib = A.load_hip()
    f = lib.dsv_mk_frame(A.SUBSAMP_420, 352, 288, 1)
    fr = f.contents
    assert fr.planes[0].stride == 416 and fr.planes[1].stride == 240 and fr.border == 1
    lib.dsv_frame_ref_dec(f)
    assert lib.dsv_lb2(1080) == 11
