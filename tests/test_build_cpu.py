"""Build guards that need no GPU: the register budget of the column kernel.

The launcher's geometry (16 waves per CU) assumes at most 128 VGPRs per lane, and the kernel's speed depends on
nothing being spilled to scratch memory - it is one 1,500-line persistent loop compiled with
-mllvm -disable-machine-licm, close to the limit, so a compiler update can tip it over.  `make resources` recompiles
the kernel with the library's own flags and prints the code-object metadata."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mckpp_f90_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_column_kernel_fits_its_register_budget():
    out = subprocess.run(["make", "-s", "-C", CSRC, "resources"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels = {}
    name = None
    for line in out.stdout.splitlines():
        m = re.match(r"\s*\.(\w+):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.groups()
        if key == "name":
            name = val
            kernels[name] = {}
        elif name is not None:
            kernels[name][key] = int(val)
    variants = {n: k for n, k in kernels.items() if "k_column_ps" in n}
    assert len(variants) == 3, f"expected the three variants of k_column_ps, found {sorted(kernels)}"
    for n, k in variants.items():
        assert k["vgpr_spill_count"] == 0, f"{n}: {k['vgpr_spill_count']} VGPRs spilled to scratch"
        assert k["private_segment_fixed_size"] == 0, f"{n}: uses {k['private_segment_fixed_size']} B of scratch per lane"
        assert k["vgpr_count"] <= 128, f"{n}: {k['vgpr_count']} VGPRs - 16 waves per CU need <= 128"
