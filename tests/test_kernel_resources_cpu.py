"""CPU tier: static resource check of every gfx950 kernel in the built library -- no GPU needed.

A kernel that spills registers to scratch still computes the right answer, so no parity test notices; it just runs several times
slower (one wide decode-attention variant did exactly that in round 2 and was re-planned onto 8 waves, csrc/engine.hip
pick_decode_splits).  The code objects bundled in libtinyllm_hip.so carry the compiler's per-kernel metadata
(private_segment_fixed_size, vgpr / sgpr spill counts): this test reads it with llvm-objdump / llvm-readelf and holds every
kernel to "no scratch, no VGPR spills" (SGPR spills go to VGPR lanes and cost a v_readlane, not memory), except the ones listed
below with the reason they are allowed to."""

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "tiny-llm_amd" / "extensions_hip" / "tiny_llm_ext_hip" / "libtinyllm_hip.so"
LLVM = Path("/opt/rocm/lib/llvm/bin")

# kernels allowed to use scratch: (substring of the mangled name, reason)
ALLOWED: list[tuple[str, str]] = []  # round 3 removed both former entries (the wide attention kernel and the spilling skinny-matmul candidate)


def kernel_metadata():
    if not LIB.exists():
        pytest.skip("library not built")
    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("no llvm tools")
    out = []
    tmp = Path(subprocess.run(["mktemp", "-d"], check=True, capture_output=True, text=True).stdout.strip())
    try:
        copy = tmp / LIB.name
        shutil.copy(LIB, copy)  # --offloading writes the extracted bundles next to its input
        subprocess.run([str(LLVM / "llvm-objdump"), "--offloading", str(copy)], check=True, capture_output=True, cwd=tmp)
        for co in sorted(tmp.glob(LIB.name + ".*gfx950")):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
            for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                field = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", block).group(1))
                out.append({"name": name.group(1), "agprs": int(block.split()[0]), "scratch": field("private_segment_fixed_size"), "vgpr_spills": field("vgpr_spill_count"),
                            "sgpr_spills": field("sgpr_spill_count"), "vgprs": field("vgpr_count"), "lds": field("group_segment_fixed_size")})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def test_no_kernel_uses_scratch_or_spills_vector_registers():
    kernels = kernel_metadata()
    assert len(kernels) > 500, f"only {len(kernels)} kernels found in {LIB.name}: the metadata parse is off"
    offenders = []
    for k in kernels:
        if k["scratch"] == 0 and k["vgpr_spills"] == 0:
            continue
        if any(tag in k["name"] for tag, _ in ALLOWED):
            continue
        offenders.append(k)
    assert not offenders, "kernels with scratch / spills:\n" + "\n".join(
        f"  {k['name']}: scratch {k['scratch']} B, vgpr spills {k['vgpr_spills']}, sgpr spills {k['sgpr_spills']}" for k in offenders)


def test_the_hot_kernels_are_in_the_library_with_the_expected_footprint():
    """The kernels the bench line runs through, by name: present, within the 512-register file of a gfx950 lane, static LDS
    under the 160 KiB of a CU (dynamic LDS is sized at launch and checked by the planners)."""
    kernels = {k["name"]: k for k in kernel_metadata()}
    for tag in ("qmv3_kernelILi1E", "qmm3_kernelILi1E", "qmm3p_kernelILi4E", "qmm6_kernelILi4ELi5E", "qmm6_kernelILi1ELi5E", "qmm7_kernelILi4ELi5ELi5E", "qmm7_kernelILi3ELi2ELi5E", "attn_decode_fused_kernelILi8ELi4ELi1E",
                "attn_decode_fused_kernelILi8ELi4ELi4E", "paged_fa_bf16_d128_kernel", "qmm_mfma_kernel", "step_end_kernel"):
        found = [k for n, k in kernels.items() if tag in n]
        assert found, f"no kernel matching {tag}"
        for k in found:
            assert k["vgprs"] <= 512 and k["lds"] <= 160 * 1024, k
    # the matrix-core attention walk (csrc/attn_mfma.h) is planned at TWO workgroups per CU: 256 registers per lane at most, MFMA
    # results in VGPRs (the softmax rescales every accumulator; its object is compiled with -amdgpu-mfma-vgpr-form)
    walk = [k for n, k in kernels.items() if "attn_decode_mfma_kernel" in n]
    assert len(walk) == 4, [k["name"] for k in walk]  # {bf16 rows, fp32 slice partials} x {bf16 pages, FP8 pages}
    for k in walk:
        assert k["vgprs"] <= 256 and k["agprs"] == 0, k  # (.vgpr_count is the unified count: architectural + accumulation)
