"""Build guards that need no GPU: the register budget of the column kernel, and its hand-scheduled LDS reads.

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
    # 3 physics x 2 solver modes, and the default and the optional physics again with the level count as a literal (63, 72, 103
    # items per column)
    assert len(variants) == 18, f"expected the eighteen variants of k_column_ps, found {sorted(kernels)}"
    general = [k for n, k in variants.items() if "ELi0EEEv" in n]
    literal = [k for n, k in variants.items() if "ELi0EEEv" not in n]
    assert len(general) == 6 and len(literal) == 12
    # what the literal buys: about half the spilled SGPRs of the general default-physics kernels
    assert max(k["sgpr_spill_count"] for k in literal) < 0.7 * min(k["sgpr_spill_count"] for k in general)
    for n, k in variants.items():
        assert k["vgpr_spill_count"] == 0, f"{n}: {k['vgpr_spill_count']} VGPRs spilled to scratch"
        assert k["private_segment_fixed_size"] == 0, f"{n}: uses {k['private_segment_fixed_size']} B of scratch per lane"
        assert k["vgpr_count"] <= 128, f"{n}: {k['vgpr_count']} VGPRs - 16 waves per CU need <= 128"


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_no_instruction_touches_a_register_with_an_lds_read_in_flight():
    """The serial sweeps issue their LDS reads through asm statements one trip ahead and wait for them with explicit
    counts (ps_lds_read2 / ps_lds_wait).  In between the value is an ordinary variable for the compiler; a copy it
    inserts there (where paths merge, or when it splits a live range) copies what the register held before the read
    lands - the hardware does not interlock LDS returns - and the result depends on timing.  tools/check_inflight.py
    follows every asm-issued read through the generated code of the three kernel variants."""
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight

    text = check_inflight.isa_text()
    assert text.count("ds_read2_b64") > 50, "the kernel's assembly does not look like k_column_ps"
    bad = check_inflight.check(text)
    assert not bad, "\n".join(f"{k[:40]} line {i}: {s} (read of line {ln})" for k, i, s, ln in bad[:20])
    # the checker itself: a copy of a register between its read and the wait is reported
    probe = """_Zprobe:
\t;;#ASMSTART
\tds_read2_b64 v[2:5], v1 offset0:0 offset1:9
\t;;#ASMEND
\ts_cbranch_scc1 .LBB9_2
.LBB9_1:
\tv_mov_b64_e32 v[6:7], v[2:3]
.LBB9_2:
\t;;#ASMSTART
\ts_waitcnt lgkmcnt(0)
\t;;#ASMEND
\tv_add_f64 v[8:9], v[2:3], v[4:5]
\ts_endpgm
.Lfunc_end9:
"""
    found = check_inflight.check(probe)
    assert [f[1] for f in found] == [7], found


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_library_build_runs_the_gate_and_records_its_compiler(built):
    """`make` links the library only behind tools/check_build.py (both checks above on the very assembly of the build),
    and the library says which compiler that was."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert "$(OUT): mckpp_kernels.o mckpp_kernels_ps.o mckpp_runtime.o .kernel_checked" in mk
    assert os.path.exists(os.path.join(CSRC, ".kernel_checked")), "the in-tree library was linked without the build gate"
    from mckpp_f90_amd import api

    assert "-unchecked" not in api.build_id()
    assert "version" in api.build_compiler().lower()
    # the gate itself: a spilled register fails the product build, an in-flight copy fails any build
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_build

    ok = "_ZN1k_column_psILi0ELi0EE:\n\ts_endpgm\n.Lfunc_end0:\n    .name: k_column_ps0\n    .vgpr_count: 120\n    .vgpr_spill_count: 0\n    .private_segment_fixed_size: 0\n"
    tmp = os.path.join(ROOT, "gpurun_out", "_gate_probe.s")
    os.makedirs(os.path.dirname(tmp), exist_ok=True)
    open(tmp, "w").write(ok)
    assert check_build.main(["check_build", tmp]) == 0
    open(tmp, "w").write(ok.replace("vgpr_spill_count: 0", "vgpr_spill_count: 3"))
    assert check_build.main(["check_build", tmp]) == 1
    assert check_build.main(["check_build", tmp, "--budget-warn-only"]) == 0
    os.remove(tmp)
