"""CPU tier: the device-only code objects of the AQL replay route (csrc/aql.h) as the build leaves them next to the library --
tl_kernels_{engine,qmv3,attn_mfma,qmm6,qmm7,qmm3}.hsaco + tl_kernels.meta (tools/kernel_meta.py).  No GPU: the files are read with llvm-readelf /
llvm-objdump.  What the route relies on and a toolchain change could silently break:
  * every kernel a single-sequence decode step launches is in the code objects, under the name the HIP fat binary uses, with its
    explicit arguments and -- where it reads gridDim -- the code-object-v5 implicit block at the offsets csrc/aql.cpp fills;
  * the code objects are the TL_COHERENT build (common.h): stores of values another launch reads are device-scope write-through
    (`sc1`), loads are plain -- and nothing in them is a `volatile` access (flat_* sc0 sc1 + s_waitcnt vmcnt(0) after every one:
    measured 0.99 -> 1.31 ms per step)."""

import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tiny-llm_amd" / "extensions_hip" / "tiny_llm_ext_hip"
LLVM = Path("/opt/rocm/lib/llvm/bin")
STEP_KERNELS = ["qmv3_kernelILi1E", "attn_decode_fused_kernelILi8ELi4ELi1E", "attn_decode_mfma_kernel", "attn_merge_cols_kernel", "attn_merge_kernelILi",
                "step_end_kernel",
                # the batched-matmul step of 5 .. 64 sequences (round 5)
                "qmm6_kernel", "qmm3_kernel", "qmm3_reduce_kernel", "weight_rows_kernel",
                # the row-streaming matmul of gate|up / qkv (round 6)
                "qmm7_kernel"]


@pytest.fixture(scope="module")
def meta(built_libs):
    path = OUT / "tl_kernels.meta"
    assert path.exists(), "make -C tiny-llm_amd/csrc builds tl_kernels.meta next to the library"
    rows = {}
    for line in path.read_text().splitlines():
        f = line.split()
        rows[f[0]] = dict(kernarg=int(f[1]), hidden=int(f[2]), args=[tuple(int(v) for v in t.split(":")) for t in f[4:4 + int(f[3])]])
    return rows


def test_code_objects_and_layouts_are_built(meta):
    for name in ("engine", "qmv3", "attn_mfma", "qmm6", "qmm7", "qmm3"):
        assert (OUT / f"tl_kernels_{name}.hsaco").stat().st_size > 10_000
    for part in STEP_KERNELS:
        assert any(part in k for k in meta), f"no kernel matching {part} in tl_kernels.meta"
    for name, k in meta.items():
        assert k["args"], name
        end = max(o + s for o, s in k["args"])
        if k["hidden"] >= 0:
            # the implicit block starts behind the explicit arguments (8-byte aligned) and the segment holds all of it up to grid_dims
            assert k["hidden"] == (end + 7) // 8 * 8 and k["kernarg"] >= k["hidden"] + 66, (name, k)
        else:
            assert k["kernarg"] >= end, (name, k)
    # the kernels of a decode step take ONE struct by value (what csrc/aql.cpp copies from the captured node), the merge launches scalars
    one = [k for n, k in meta.items() if "qmv3_kernel" in n or "attn_decode" in n or "step_end_kernel" in n]
    assert one and all(len(k["args"]) == 1 and k["args"][0][0] == 0 for k in one)


def test_kernel_meta_refuses_moved_implicit_arguments(tmp_path):
    """tools/kernel_meta.py fails the BUILD when an implicit argument is not at its code-object-v5 offset."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("kernel_meta", ROOT / "tools" / "kernel_meta.py")
    km = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(km)
    assert km.HIDDEN_REL["hidden_block_count_x"] == 0 and km.HIDDEN_REL["hidden_group_size_x"] == 12 and km.HIDDEN_REL["hidden_grid_dims"] == 64
    good = list(km.kernels_of(OUT / "tl_kernels_attn_mfma.hsaco"))
    assert good and all(size > 0 for _, size, _ in good)


def test_the_code_objects_are_the_write_through_build(built_libs):
    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("no llvm tools")
    asm = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(OUT / "tl_kernels_qmv3.hsaco")], check=True, capture_output=True, text=True).stdout
    body = re.search(r"<_ZN2tl11qmv3_kernelILi1ELi2ELi4ELi1ELi0ELi10ELi0EEEvNS_8Qmv3ArgsE>:(.*?)s_endpgm", asm, re.S)
    assert body, "the qkv GEMV of one row (KS 2, 4 waves, RMSNorm prologue, store epilogue, 10 groups per wave) is not in the code object"
    text = body.group(1)
    stores = [l for l in text.splitlines() if "global_store" in l or "flat_store" in l]
    assert stores and all(" sc1" in l for l in stores if "store_short" in l), "the output rows must be written through (device-scope stores)"
    loads = [l for l in text.splitlines() if re.search(r"\b(global|flat)_load", l)]
    assert loads and not any(" sc0 sc1" in l or " sc1" in l for l in loads), "loads stay plain: every hand-over address is written once per step"
    assert "flat_load" not in text and "flat_store" not in text, "no volatile accesses"
    # the batched-decode matmuls store through buffer resources: every buffer store of their code objects carries sc1
    for obj, kern in (("qmm6", "qmm6_kernel"), ("qmm7", "qmm7_kernel"), ("qmm3", "qmm3_kernel")):
        a6 = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(OUT / f"tl_kernels_{obj}.hsaco")], check=True, capture_output=True, text=True).stdout
        bs = [l for l in a6.splitlines() if "buffer_store" in l]
        assert bs and all(" sc1" in l for l in bs), f"{kern}: a buffer store without sc1 in the write-through build"
        assert not any(" sc1" in l for l in a6.splitlines() if "buffer_load" in l), f"{kern}: loads stay plain"
    # the fat binary inside the library is the PLAIN build: no write-through stores there
    lib_asm_dir = Path(subprocess.run(["mktemp", "-d"], check=True, capture_output=True, text=True).stdout.strip())
    try:
        import shutil

        shutil.copy(OUT / "libtinyllm_hip.so", lib_asm_dir / "lib.so")
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(lib_asm_dir / "lib.so")], check=True, capture_output=True, cwd=lib_asm_dir)
        found = False
        for co in sorted(lib_asm_dir.glob("lib.so.*gfx950")):
            a = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(co)], check=True, capture_output=True, text=True).stdout
            m = re.search(r"<_ZN2tl11qmv3_kernelILi1ELi2ELi4ELi1ELi0ELi10ELi0EEEvNS_8Qmv3ArgsE>:(.*?)s_endpgm", a, re.S)
            if m:
                found = True
                assert not any(" sc1" in l for l in m.group(1).splitlines() if "global_store" in l), "the library's own kernels keep plain stores"
        assert found
    finally:
        shutil.rmtree(lib_asm_dir, ignore_errors=True)
