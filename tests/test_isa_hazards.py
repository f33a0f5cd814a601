"""CPU (hipcc cross-compiles): the inline-asm AccVGPR MFMAs of k_mlp_jtj are invisible to the compiler's hazard recogniser; the
wait states between an MFMA and the first non-matrix read of its result are placed by hand (mfma_acc_settle, csrc/sdf_mlp.hpp).
A runtime test cannot prove their presence (the hazard is timing dependent), the assembly can."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_no_accumulator_is_read_before_its_mfma_wait_states(sdf_isa):
    import check_mfma_hazards as chk
    n, bad = chk.check(sdf_isa)
    bad2 = chk.check_valu_def_before_mfma(sdf_isa)
    assert not bad2, bad2[:5]
    assert n > 1000, "expected the asm MFMAs of k_mlp_jtj in the ISA, found %d" % n
    assert not bad, bad[:5]
