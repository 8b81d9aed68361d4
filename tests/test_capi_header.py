"""include/zyg_capi.h against the reference's capi.zig: every su_* export is declared, with the line its citation names.
Runs only where /root/reference exists (this container); the GPU box has no reference tree."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAPI = "/root/reference/src/capi/capi.zig"

pytestmark = pytest.mark.skipif(not os.path.exists(CAPI), reason="reference tree not present")


def _exports():
    out = {}
    for i, line in enumerate(open(CAPI).read().split("\n")):
        m = re.match(r"export fn (\w+)", line)
        if m:
            out[m.group(1)] = i + 1
    return out


def test_every_export_is_declared_and_cited_at_its_line():
    header = open(os.path.join(ROOT, "include", "zyg_capi.h")).read()
    exports = _exports()
    assert len(exports) == 30
    for name, line in exports.items():
        m = re.search(r"int32_t " + name + r"\([^;]*;\s*/\* capi\.zig:(\d+)-(\d+)", header, re.S)
        assert m, f"{name}: no declaration with a capi.zig citation"
        assert int(m.group(1)) == line, f"{name}: cited at {m.group(1)}, defined at {line}"
        assert int(m.group(2)) > line


def test_library_exports_every_declared_symbol():
    import ctypes
    so = os.path.join(ROOT, "zyg_amd", "libzyg_amd.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    lib = ctypes.CDLL(so)
    header = open(os.path.join(ROOT, "include", "zyg_capi.h")).read() + open(os.path.join(ROOT, "include", "zygpu.h")).read()
    names = set(re.findall(r"\b((?:su|zygpu)_\w+)\s*\(", header))
    assert {"su_prop_set_transformation_frame", "su_image_update", "zygpu_render"} <= names
    for n in names:
        assert hasattr(lib, n), n


def _struct_fields(header, name):
    """Field names of `typedef struct <name> { ... } <name>;` in declaration order."""
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + r";", header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            fields.append(re.search(r"(\w+)(\[[^\]]*\])?$", decl).group(1))
    return fields


def test_integration_md_scene_desc_sketch_follows_the_header():
    """INTEGRATION.md section B shows the `extern struct SceneDesc` a Zig maintainer would copy: the same fields in the same order
    as include/zygpu.h's ZygpuSceneDesc, and the Python mirror zyg_amd.SceneDesc likewise."""
    header = open(os.path.join(ROOT, "include", "zygpu.h")).read()
    fields = _struct_fields(header, "ZygpuSceneDesc")
    assert fields[0] == "props" and "sun_radiance" in fields
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sketch = re.search(r"pub const SceneDesc = extern struct \{(.*?)\n\};", text, re.S).group(1)
    sketch = re.sub(r"//[^\n]*", "", sketch)
    zig = re.findall(r"(\w+)\s*:", sketch)
    assert zig == fields, (zig, fields)
    import zyg_amd
    assert [f[0] for f in zyg_amd.SceneDesc._fields_] == fields
