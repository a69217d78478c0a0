"""Build gate of the column kernel, run by mckpp_f90_amd/csrc/Makefile on the device assembly of every library build
(the product's and the EXTRA-flag variants of tools/): a library whose kernel fails it is not linked.

  1. no instruction touches a VGPR whose asm-issued LDS read has not been waited for (tools/check_inflight.py) -
     always fatal: such a build computes timing-dependent numbers;
  2. every k_column_ps variant fits 128 VGPRs with nothing spilled and no scratch (16 waves per CU is what the
     launcher's geometry assumes) - fatal for the product build, a warning for builds with EXTRA flags (profiling
     stamps need registers of their own).

    python tools/check_build.py kernel.s [--budget-warn-only]
"""
import re
import sys

import check_inflight


def resources(text):
    out, name = {}, None
    for line in text.splitlines():
        m = re.match(r"\s*\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "name":
            name = val
            out[name] = {}
        elif name is not None:
            out[name][key] = int(val)
    return {n: k for n, k in out.items() if "k_column_ps" in n}


def main(argv):
    text = open(argv[1]).read()
    warn_only = "--budget-warn-only" in argv
    bad = check_inflight.check(text)
    for k, i, s, ln in bad[:40]:
        print(f"check_build: {(k or '')[:48]} line {i}: {s}   <- ds_read of line {ln} still outstanding", file=sys.stderr)
    rc = 1 if bad else 0
    res = resources(text)
    if not res:
        print("check_build: no k_column_ps variant in the assembly", file=sys.stderr)
        return 1
    for n, k in sorted(res.items()):
        over = k.get("vgpr_spill_count", 0) != 0 or k.get("private_segment_fixed_size", 0) != 0 or k.get("vgpr_count", 0) > 128
        if over:
            print(f"check_build: {'warning' if warn_only else 'error'}: {n}: {k}", file=sys.stderr)
            if not warn_only:
                rc = 1
    print(f"check_build: {len(res)} kernel variants, {len(bad)} in-flight violations, VGPRs "
          + " ".join(str(k.get('vgpr_count')) for _, k in sorted(res.items())) + (" - FAILED" if rc else " - ok"))
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
