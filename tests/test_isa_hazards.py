"""CPU (hipcc cross-compiles): the inline-asm AccVGPR MFMAs of k_mlp_jtj are invisible to the compiler's hazard recogniser; the
wait states between an MFMA and the first non-matrix read of its result are placed by hand (mfma_acc_settle, csrc/sdf_mlp.hpp).
A runtime test cannot prove their presence (the hazard is timing dependent), the assembly can."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_accumulator_is_read_before_its_mfma_wait_states():
    import check_mfma_hazards as chk
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "sdf_refine.s")
        chk.compile_isa(path)
        n, bad = chk.check(path)
        bad2 = chk.check_valu_def_before_mfma(path)
    assert not bad2, bad2[:5]
    assert n > 1000, "expected the asm MFMAs of k_mlp_jtj in the ISA, found %d" % n
    assert not bad, bad[:5]
