"""CPU tier: libmagicdance_hip.so builds for gfx950 (hipcc cross-compiles without a GPU), loads, and exports exactly
the entry points include/magicdance_hip.h declares, with the ctypes signatures the product binds (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "magicdance_hip.h")


@pytest.fixture(scope="module")
def lib_path():
    from magicdance_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "magicdance_amd", "csrc", "build.sh")])
    return _lib.LIB_PATH


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(md_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound(lib_path):
    from magicdance_amd import _lib
    declared = _declared_functions()
    assert len(declared) >= 20
    lib = ctypes.CDLL(lib_path)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/magicdance_hip.h but not exported"
    # the Python binding covers the same set (a symbol added to the header must be bound, and vice versa)
    assert sorted(_lib.SIGNATURES) == declared
    loaded = _lib.load()
    assert loaded.md_version() == 10 and loaded.md_arch() == b"gfx950"
    # the driver's build check (__graft_entry__.build) asserts the same ABI version: a bump that forgets it fails build() and
    # smoke() on the GPU box (round 5 found it that way)
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert int(re.search(r"md_version\(\) == (\d+)", entry).group(1)) == loaded.md_version()


def test_param_structs_match_header_layout():
    """ctypes mirrors of the parameter structs: field order / count must follow the header."""
    from magicdance_amd import _lib
    src = open(HEADER).read()
    for cname, cls in (("md_igemm_params", _lib.IgemmParams), ("md_attention_params", _lib.AttentionParams),
                       ("md_groupnorm_params", _lib.GroupNormParams), ("md_ff_block_params", _lib.FfBlockParams)):
        body = re.search(r"typedef struct \{([^{}]*)\} " + cname + ";", src).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1].lstrip("*")
            fields.append(first)
            fields.extend(n.strip().lstrip("*") for n in names[1:])
        assert [f for f, _ in cls._fields_] == fields, cname


def test_launchers_reject_bad_arguments_without_a_gpu(lib_path):
    """Argument validation happens before any HIP call, so it is checkable here: NULL / misaligned -> MD_ERR_BAD_ARG."""
    from magicdance_amd import _lib
    lib = _lib.load()
    p = _lib.IgemmParams()
    assert lib.md_igemm(ctypes.byref(p), None) == -1
    a = _lib.AttentionParams()
    assert lib.md_attention(ctypes.byref(a), None) == -1
    f = _lib.FfBlockParams()
    assert lib.md_ff_block(ctypes.byref(f), None) == -1
    assert lib.md_ff_block_supported(8192, 320) == 1 and lib.md_ff_block_supported(8192, 1280) == 0
    assert lib.md_layernorm(None, None, None, None, 4, 320, 1e-5, None) == -1
    assert lib.md_groupnorm_workspace_bytes(2, 4096, 32) > 0


def test_param_structs_match_header_offsets(tmp_path):
    """byte layout, not just field names: compile the header with gcc and compare sizeof / offsetof of every field of the three
    parameter structs with the ctypes mirrors (a type slip -- int32 vs int64, pointer vs int -- would keep the names in order)."""
    from magicdance_amd import _lib
    structs = (("md_igemm_params", _lib.IgemmParams), ("md_attention_params", _lib.AttentionParams),
               ("md_groupnorm_params", _lib.GroupNormParams), ("md_ff_block_params", _lib.FfBlockParams))
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void) {"]
    for cname, cls in structs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in structs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, f"{cname}.{f}"


def test_gemm_translation_units_are_built_without_packed_fp32():
    """csrc/build.sh: the four GEMM translation units carry -target-feature -packed-fp32-ops (round 5: v_pk_fma_f32 with op_sel operands in
    the folded-LayerNorm epilogue was not repeatable on gfx950, tests/test_gpu_repeatability.py); attention / norm / elementwise do not"""
    sh = open(os.path.join(ROOT, "magicdance_amd", "csrc", "build.sh")).read()
    m = re.search(r'case "\$f" in ([a-z_|]+)\) EXTRA="\$EXTRA -Xclang -target-feature -Xclang -packed-fp32-ops"', sh)
    assert m and set(m.group(1).split("|")) == {"igemm", "igemm_ring", "igemm_stream", "ffblock"}
