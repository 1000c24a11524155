#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch footprint of the built library (compiler metadata, no GPU): the table behind
tests/test_kernel_resources_cpu.py.  `python tools/kernel_resources_report.py [substring ...]` (default: the decode-step kernels)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import test_kernel_resources_cpu as T  # noqa: E402

DEFAULT = ["qmv3_kernelILi1ELi4ELi4ELi1ELi0ELi5", "qmv3_kernelILi1ELi4ELi4ELi0ELi1ELi8", "qmv3_kernelILi1ELi4ELi4ELi2ELi1ELi8", "qmv3_kernelILi1ELi4ELi4ELi1ELi2ELi5",
           "qmv3_kernelILi1ELi8ELi8ELi0ELi1ELi10", "qmv3_kernelILi1ELi16ELi16", "attn_decode_fused_kernelILi8ELi4ELi1E", "attn_decode_fused_kernelILi8ELi4ELi4E",
           "attn_merge_kernel", "attn_merge_cols_kernel", "step_end_kernel", "qmm3_kernelILi1E", "qmm3_kernelILi4ELi1E", "qmm3p_kernel", "qmm3_reduce_kernel",
           "paged_fa_bf16_d128_kernel", "qmm_mfma_kernel"]


def main():
    tags = sys.argv[1:] or DEFAULT
    ks = T.kernel_metadata()
    print(f"{len(ks)} kernels in {T.LIB.name}; with scratch or VGPR spills: {sum(1 for k in ks if k['scratch'] or k['vgpr_spills'])}")
    print(f"{'kernel':100s} {'vgpr':>5s} {'sgpr-spill':>10s} {'static LDS':>10s} {'scratch':>7s}")
    for tag in tags:
        for k in sorted((k for k in ks if tag in k["name"]), key=lambda k: k["name"]):
            print(f"{k['name'][6:106]:100s} {k['vgprs']:5d} {k['sgpr_spills']:10d} {k['lds']:10d} {k['scratch']:7d}")


if __name__ == "__main__":
    main()
