"""CPU tier: the ctypes mirror of every struct in include/tinyllm_engine.h has the C compiler's size and field offsets.

The reference-side binding a maintainer adds and this repo's own binding (extensions_hip/tiny_llm_ext_hip) pass
these structs by pointer across the C ABI; a field added on one side only shifts everything behind it silently.  gcc compiles a
probe against the header and prints sizeof / offsetof of every field; the ctypes classes must agree."""

import ctypes
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tiny-llm_amd" / "extensions_hip"))

PAIRS = {  # C struct -> ctypes class
    "tl_w4": "TlW4", "tl_layer_weights": "TlLayerWeights", "tl_moe_weights": "TlMoeWeights", "tl_engine_config": "TlEngineConfig",
    "tl_engine_stats": "TlEngineStats", "tl_step_profile": "TlStepProfile", "tl_linear_info": "TlLinearInfo",
    "tl_attention_info": "TlAttentionInfo", "tl_linear_ex": "TlLinearEx", "tl_step_check": "TlStepCheck",
}


def c_fields(header: str, struct: str) -> list[str]:
    """Field names of `typedef struct <struct> { ... } <struct>;` in declaration order (comments stripped; arrays keep their name)."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            name = re.sub(r"\[.*?\]", "", part.strip().split()[-1]).lstrip("*")
            names.append(name)
    return names


def test_ctypes_structs_match_the_c_layout(tmp_path):
    import tiny_llm_ext_hip as ext

    header = (ROOT / "include" / "tinyllm_engine.h").read_text()
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "tinyllm_engine.h"', "int main(void) {"]
    for struct in PAIRS:
        lines.append(f'    printf("{struct} size %zu\\n", sizeof({struct}));')
        for f in c_fields(header, struct):
            lines.append(f'    printf("{struct} {f} %zu\\n", offsetof({struct}, {f}));')
    lines += ["    return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    want: dict[str, dict[str, int]] = {}
    for line in out.splitlines():
        struct, field, value = line.split()
        want.setdefault(struct, {})[field] = int(value)
    for struct, cls_name in PAIRS.items():
        cls = getattr(ext, cls_name)
        assert ctypes.sizeof(cls) == want[struct]["size"], f"{cls_name}: ctypes size {ctypes.sizeof(cls)}, C size {want[struct]['size']}"
        names = [n for n, *_ in cls._fields_]
        assert names == [f for f in want[struct] if f != "size"], f"{cls_name}: field names / order differ from {struct}"
        for n in names:
            assert getattr(cls, n).offset == want[struct][n], f"{cls_name}.{n}: offset {getattr(cls, n).offset}, C offset {want[struct][n]}"
