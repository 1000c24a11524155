#!/usr/bin/env python3
"""Are the kernels of the working tree the SAME MACHINE CODE as those of an earlier commit?

    python tools/compare_kernel_isa.py <git-rev> [file.hip ...]        (default files: engine.hip qmv3.hip)

Compiles csrc/<file> for gfx950 (device side only, to assembly) from the working tree and from <git-rev>, and compares every
kernel that exists in both, instruction by instruction (labels renumbered; comments dropped).  Scalar loads from the kernel-
argument segment may differ in their OFFSET only (a struct that grew at its end moves the implicit arguments behind it).  Kernel
templates that gained a trailing defaulted parameter are matched through --alias OLD_SUFFIX=NEW_SUFFIX.

Use: opt-in code paths added without device time (round 2: TL_ATTN_QKV_PARTIALS, TL_WO_MERGES_ATTN) must leave the measured
default kernels untouched -- this shows it without a GPU.  Build-container tool, not part of the product."""
import argparse
import difflib
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = "tiny-llm_amd/csrc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-amdgpu-kernarg-preload-count=16", "--cuda-device-only", "-S", "-x", "hip"]


def kernels(asm: str) -> dict:
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", asm, re.M):
        end = asm.find("s_endpgm", m.end())
        if end < 0:
            continue
        body = re.sub(r";.*", "", asm[m.end():end])
        body = re.sub(r"\.LBB\d+_", ".LBB_", body)
        out[m.group(1)] = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(".")]
    return out


def compile_to_asm(src_dir: Path, name: str, out: Path) -> str:
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, name, "-o", str(out)], cwd=src_dir, check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("rev")
    ap.add_argument("files", nargs="*", default=["engine.hip", "qmv3.hip"])
    ap.add_argument("--alias", action="append", default=[], help="OLD_SUFFIX=NEW_SUFFIX of mangled kernel names")
    args = ap.parse_args()
    aliases = [a.split("=", 1) for a in args.alias]
    worst = 0
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        tar = subprocess.run(["git", "archive", args.rev, CSRC, "include"], cwd=ROOT, check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", str(tmp)], input=tar, check=True)
        for name in args.files:
            new = kernels(compile_to_asm(ROOT / CSRC, name, tmp / f"new_{name}.s"))
            old = kernels(compile_to_asm(tmp / CSRC, name, tmp / f"old_{name}.s"))
            same = offsets_only = different = missing = 0
            for k, body in old.items():
                k2 = k
                for a, b in aliases:
                    if k.endswith(a) and k2 not in new:
                        k2 = k[: -len(a)] + b
                if k2 not in new:
                    missing += 1
                    continue
                if body == new[k2]:
                    same += 1
                    continue
                d = [l for l in difflib.unified_diff(body, new[k2], lineterm="", n=0) if l[0] in "+-" and not l.startswith(("+++", "---"))]
                real = [l for l in d if not re.search(r"s_load_dword\w* s(\[\d+:\d+\]|\d+), s\[\d+:\d+\], 0x", l)]
                if real:
                    different += 1
                    print(f"  DIFFERENT {k}: {len(real)} lines, e.g. {real[:2]}")
                else:
                    offsets_only += 1
            print(f"{name}: {len(old)} kernels at {args.rev}: {same} identical, {offsets_only} identical but for kernel-argument offsets, "
                  f"{different} different, {missing} no longer present; {len(new) - (len(old) - missing)} new")
            worst = max(worst, different)
    return 1 if worst else 0


if __name__ == "__main__":
    sys.exit(main())
