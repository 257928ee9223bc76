"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/mmf_amd.h declares (no compute calls here: there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

from mmf_amd import _native


def _declared_functions():
    src = open(_native.HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


@pytest.fixture(scope="module")
def built_lib():
    if not os.path.exists(_native.LIB_PATH):
        from mmf_amd.csrc.build import build
        build(verbose=False)
    return ctypes.CDLL(_native.LIB_PATH)


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for required in ("mmf_gemm_bf16", "mmf_attention_fwd", "mmf_attention_bwd", "mmf_layernorm_fwd",
                     "mmf_layernorm_bwd", "mmf_embed_text_fwd", "mmf_bce_logits_fwd", "mmf_adamw_step"):
        assert required in names


def test_library_exports_every_declared_symbol(built_lib):
    missing = [n for n in _declared_functions() if not hasattr(built_lib, n)]
    assert not missing, "declared in include/mmf_amd.h but not exported: %s" % missing


def test_abi_version_and_target(built_lib):
    built_lib.mmf_amd_target.restype = ctypes.c_char_p
    assert built_lib.mmf_amd_abi_version() == 1
    assert built_lib.mmf_amd_target() == b"gfx950"


def test_binding_structs_match_header_field_order():
    src = re.sub(r"/\*.*?\*/", "", open(_native.HEADER_PATH).read(), flags=re.S)
    body = re.search(r"typedef struct mmf_gemm_desc \{(.*?)\} mmf_gemm_desc;", src, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].split()[-1].lstrip("*")
        fields.append(first)
        fields.extend(n.strip().lstrip("*") for n in names[1:])
    assert fields == [f[0] for f in _native.GemmDesc._fields_]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.NativeLibraryError):
        _native.lib()


def test_argument_errors_are_reported_not_swallowed(built_lib):
    built_lib.mmf_amd_last_error.restype = ctypes.c_char_p
    d = _native.GemmDesc()
    rc = built_lib.mmf_gemm_bf16(ctypes.byref(d), None)
    assert rc != 0
    assert b"null operand" in built_lib.mmf_amd_last_error()


def test_library_is_not_older_than_its_sources():
    """The in-tree .so travels to the GPU box as built: a stale build (header / kernel edited, library not rebuilt)
    shows up there as struct-layout garbage.  __graft_entry__.build() rebuilds; this catches forgetting to."""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "mmf_amd", "libmmf_amd.so")
    srcs = glob.glob(os.path.join(root, "mmf_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "mmf_amd", "csrc", "*.h")) + \
        [os.path.join(root, "include", "mmf_amd.h")]
    newest = max(srcs, key=os.path.getmtime)
    assert os.path.getmtime(so) >= os.path.getmtime(newest), "rebuild: %s is newer than libmmf_amd.so" % os.path.relpath(newest, root)


def test_tunables_round_trip_and_the_site_tags_stay_clear_of_the_other_flag_bits(built_lib):
    """MMF_TUN_NT_SITE_KEEP (the per-call-site exception to the non-temporal epilogue stores) and every other knob default to 0 = the measured
    behaviour; seven knobs are left (round 6 removed the measurement switches whose losing branches are gone); the site tag occupies bits 20..23 of mmf_gemm_desc::debug_flags, which no other switch reads."""
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(_native.HEADER_PATH).read()
    count = int(re.search(r"MMF_TUN_COUNT = (\d+)", header).group(1))
    keep = int(re.search(r"MMF_TUN_NT_SITE_KEEP = (\d+)", header).group(1))
    assert 0 <= keep < count
    assert len(re.findall(r"MMF_TUN_[A-Z_0-9]+ = \d+", header)) - 1 <= 8          # (MMF_TUN_COUNT itself is not a knob)
    for t in range(count):
        assert built_lib.mmf_amd_get_tunable(t) == 0, t                  # every knob defaults to the built-in behaviour
    try:
        assert built_lib.mmf_amd_set_tunable(keep, 0x1FE) == 0 and built_lib.mmf_amd_get_tunable(keep) == 0x1FE
    finally:
        assert built_lib.mmf_amd_set_tunable(keep, 0) == 0
    assert built_lib.mmf_amd_set_tunable(count, 1) != 0                  # unknown tunable: refused
    sites = {n: int(v) for n, v in re.findall(r"(MMF_SITE_[A-Z_]+) = (\d+)", header)}
    assert sorted(sites.values()) == list(range(1, 9))
    used = [int(x) for x in re.findall(r"debug_flags & (\d+)", open(os.path.join(ROOT, "mmf_amd", "csrc", "gemm.hip")).read())]
    assert used and all(u < (1 << 20) for u in used)
    ops = open(os.path.join(ROOT, "mmf_amd", "csrc", "torch_ops.cpp")).read()
    assert all(n in ops for n in sites), [n for n in sites if n not in ops]


def test_python_site_tags_and_descriptor_mirror_match_the_header():
    """The Python twin of the layer node tags its GEMM calls with the header's MMF_SITE_* values (the per-site store policy of round 4 reads them),
    and the ctypes mirror of mmf_attn_desc carries every field of the C struct in order (round 4 added mask_query_stride)."""
    header = open(_native.HEADER_PATH).read()
    sites = {n: int(v) for n, v in re.findall(r"MMF_SITE_([A-Z_]+) = (\d+)", header)}
    for name, value in sites.items():
        assert getattr(_native, "SITE_" + name) == value, name
    assert _native.gemm_site(3) == 3 << 20 and _native.gemm_site(0) == 0
    body = re.search(r"typedef struct mmf_attn_desc \{(.*?)\} mmf_attn_desc;", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    c_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if " " in decl else decl
        for part in names.split(","):
            c_fields.append(part.replace("*", " ").split()[-1])
    assert c_fields == [f[0] for f in _native.AttnDesc._fields_], (c_fields, [f[0] for f in _native.AttnDesc._fields_])
    assert int(re.search(r"#define MMF_MT_MAX (\d+)", header).group(1)) == _native.MT_MAX


def test_every_ctypes_mirror_has_the_size_and_field_offsets_the_c_compiler_gives_the_header_struct(tmp_path):
    """The ctypes structures of mmf_amd/_native.py are hand-written mirrors of include/mmf_amd.h: a field added to one side only (round 6 did that to
    mmf_attn_draw_site for one commit) shifts every later field silently.  gcc compiles the header and prints sizeof / offsetof of every field of
    every mirrored struct; the mirrors must agree."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"mmf_gemm_desc": _native.GemmDesc, "mmf_attn_desc": _native.AttnDesc, "mmf_attn_draw_site": _native.AttnDrawSite,
             "mmf_attn_bwd_desc": _native.AttnBwdDesc, "mmf_adamw_multi_desc": _native.AdamWMultiDesc, "mmf_wra_desc": _native.WraDesc,
             "mmf_ln_reduce_list": _native.LnReduceList, "mmf_tensor_list": _native.TensorList, "mmf_transpose_list": _native.TransposeList,
             "mmf_offset_list": _native.OffsetList, "mmf_ln_bwd_desc": _native.LnBwdDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include <stdint.h>', '#include "mmf_amd.h"', 'int main(void) {']
    for cname, mirror in pairs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f in mirror._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f[0]))
        lines.append('printf("\\n");')
    lines += ["return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.dirname(_native.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out:
        parts = line.split()
        mirror = pairs[parts[0]]
        want = [int(x) for x in parts[1:]]
        got = [C.sizeof(mirror)] + [getattr(mirror, f[0]).offset for f in mirror._fields_]
        assert got == want, (parts[0], got, want)
