"""conv_halo2_kernel's inline-asm loads (weight fragments by buffer_load_dwordx4, pixel fragments by ds_read_b128) are invisible to
hipcc's waitcnt bookkeeping: every wait is hand-counted, and "an asm load's VGPR destination counts as written at the statement, so the
compiler may read, copy, spill or reuse it before the data lands" (cdna_hip_programming.md section 5.7).  Round 6 shipped exactly that
bug for a few hours -- registers of loads still in flight at the K loop's exit re-used by the epilogue's first temporaries: a few wrong
elements per launch whenever the memory system was slow, found by tests/test_gpu_determinism.py.  This test compiles the kernels to
ISA (hipcc cross-compiles without a GPU) and runs scripts/h2_audit.py over every instantiation: both queues simulated through the K
loop (twice: the back edge) and the drain behind it -- no instruction may touch a register whose load has not been waited for, and no
LDS-DMA piece may be outstanding at a workgroup barrier."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "yolo-v4-tf.keras_amd", "csrc")


@pytest.mark.parametrize("dtype", ["bf16"])
def test_halo2_isa_has_no_premature_register_use(dtype, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / f"h2_{dtype}.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                    os.path.join(CSRC, f"conv_halo2_{dtype}.hip"), "-o", str(out)], check=True, capture_output=True, cwd=CSRC)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "h2_audit.py"), str(out)], capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("_ZN2y4")]
    assert len(lines) >= 6, r.stdout + r.stderr                      # every instantiation of the tile table was looked at
    assert all("MFMAs, 0 finding(s)" in l for l in lines) and r.returncode == 0, r.stdout


def test_the_audit_sees_a_planted_bug(tmp_path):
    """The auditor itself: a listing in which a register is re-used while its load is in flight, and one with an LDS-DMA piece
    outstanding at a barrier, must both be reported."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import h2_audit
    ok = ["buffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen", "s_waitcnt vmcnt(0)", "v_mfma_f32_32x32x16_bf16 a[0:15], v[4:7], v[8:11], a[0:15]"]
    bad = ["buffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen", "v_add_u32_e32 v5, 1, v2", "s_waitcnt vmcnt(0)"]
    dma = ["buffer_load_dwordx4 v9, s[4:7], s3 offen lds", "buffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen", "s_waitcnt vmcnt(1)", "s_barrier"]
    late = ["buffer_load_dwordx4 v9, s[4:7], s3 offen lds", "buffer_load_dwordx4 v[4:7], v1, s[8:11], 0 offen", "s_waitcnt vmcnt(2)", "s_barrier"]
    num = lambda ls: list(enumerate(ls, 1))
    assert h2_audit.audit(num(ok), "ok") == []
    assert len(h2_audit.audit(num(bad), "bad")) == 1
    assert h2_audit.audit(num(dma), "dma") == []                     # vmcnt(1): the older piece has landed
    assert len(h2_audit.audit(num(late), "late")) == 1               # vmcnt(2): it may not have
