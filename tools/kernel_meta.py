#!/usr/bin/env python3
"""Build step (tiny-llm_amd/csrc/Makefile): kernel-argument layouts of the device-only code objects the AQL replay path loads.

    python tools/kernel_meta.py OUT.meta  A.hsaco B.hsaco ...

The decode engine can replay a captured step as hand-written AQL dispatch packets on its own HSA queue (csrc/aql.h).  A packet needs
the kernel's argument segment laid out by the builder: the explicit arguments at their offsets and -- where the kernel reads
gridDim / blockDim (code object v5: `hidden_block_count_x` ...) -- the implicit block behind them.  HSA reports only the segment's
size; the layout is in the code object's metadata note, which this script reads with llvm-readelf and flattens into one line per
kernel:

    <mangled name> <kernarg bytes> <hidden base or -1> <n explicit> <offset>:<size> ...

`hidden base` is the offset of hidden_block_count_x; the script REFUSES a code object whose implicit arguments do not sit at the
code-object-v5 offsets the runtime fills (block counts +0/+4/+8, group sizes +12/+14/+16, remainders +18/+20/+22, global offsets
+40/+48/+56, grid dims +64), so a toolchain that moves them fails the build instead of a launch."""

import re
import subprocess
import sys
from pathlib import Path

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
HIDDEN_REL = {"hidden_block_count_x": 0, "hidden_block_count_y": 4, "hidden_block_count_z": 8, "hidden_group_size_x": 12,
              "hidden_group_size_y": 14, "hidden_group_size_z": 16, "hidden_remainder_x": 18, "hidden_remainder_y": 20,
              "hidden_remainder_z": 22, "hidden_global_offset_x": 40, "hidden_global_offset_y": 48, "hidden_global_offset_z": 56,
              "hidden_grid_dims": 64}
# implicit arguments a kernel of this library may declare but never needs filled (no printf, no device enqueue, no dynamic-LDS query)
IGNORED = {"hidden_printf_buffer", "hidden_hostcall_buffer", "hidden_multigrid_sync_arg", "hidden_heap_v1", "hidden_default_queue",
           "hidden_completion_action", "hidden_none", "hidden_dynamic_lds_size", "hidden_private_base", "hidden_shared_base", "hidden_queue_ptr"}


def kernels_of(path: Path):
    notes = subprocess.run([READELF, "--notes", str(path)], check=True, capture_output=True, text=True).stdout
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        size = re.search(r"\.kernarg_segment_size:\s+(\d+)", block)
        if not name or not size:
            continue
        args = []
        body = block.split(".args:")[1].split(".group_segment_fixed_size")[0] if ".args:" in block else ""
        for arg in re.split(r"\n\s*- ", body):
            off = re.search(r"\.offset:\s+(\d+)", arg)
            sz = re.search(r"\.size:\s+(\d+)", arg)
            kind = re.search(r"\.value_kind:\s+(\S+)", arg)
            if off and sz and kind:
                args.append((int(off.group(1)), int(sz.group(1)), kind.group(1)))
        yield name.group(1), int(size.group(1)), args


def main() -> int:
    out, objs = Path(sys.argv[1]), [Path(p) for p in sys.argv[2:]]
    lines, seen = [], set()
    for obj in objs:
        for name, size, args in kernels_of(obj):
            if name in seen:  # a kernel of a shared header compiled into two code objects: same layout, one entry
                continue
            seen.add(name)
            explicit = [(o, s) for o, s, k in args if not k.startswith("hidden_")]
            hidden = {k: o for o, s, k in args if k.startswith("hidden_")}
            base = hidden.get("hidden_block_count_x", -1)
            for kind, off in hidden.items():
                if kind in IGNORED:
                    continue
                if kind not in HIDDEN_REL or base < 0 or off - base != HIDDEN_REL[kind]:
                    print(f"kernel_meta: {name}: implicit argument {kind} at {off} (base {base}) is not at its code-object-v5 offset", file=sys.stderr)
                    return 1
            lines.append(" ".join([name, str(size), str(base), str(len(explicit))] + [f"{o}:{s}" for o, s in explicit]))
    out.write_text("\n".join(lines) + "\n")
    print(f"kernel_meta: {len(lines)} kernels -> {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
