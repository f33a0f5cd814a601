"""Static check of the compiled ISA: the AccVGPR form of the MFMA in `k_mlp_jtj` is inline asm (csrc/sdf_mlp.hpp, mfma32t), so the
compiler inserts none of the software wait states an MFMA result needs before a non-matrix instruction reads it.  This script
compiles sdf_refine.hip to assembly and verifies that no v_accvgpr_read of a register written by an asm MFMA appears between that
MFMA and the next `s_nop 15` (mfma_acc_settle).  Exit code 0 = clean.  Usage: python tools/check_mfma_hazards.py [file.s]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_isa(out):
    src = os.path.join(ROOT, "qsp_slam_amd", "csrc", "sdf_refine.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "-S",
                           "--cuda-device-only", "-o", out, src], cwd=os.path.dirname(src))


def check(path):
    lines = open(path).read().split("\n")
    mf = re.compile(r"v_mfma_f32_32x32x2_f32 a\[(\d+):(\d+)\]")
    rd = re.compile(r"v_accvgpr_read_b32 v\d+, a(\d+)")
    n_mfma, bad, pending, in_asm = 0, [], set(), False
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        m = mf.search(l)
        if m and in_asm:                      # builtin MFMAs are outside ASM blocks: the compiler handles their hazards
            n_mfma += 1
            pending.update(range(int(m.group(1)), int(m.group(2)) + 1))
            continue
        if "s_nop 15" in l:
            pending = set()
            continue
        m = rd.search(l)
        if m and int(m.group(1)) in pending:
            bad.append((i + 1, l.strip()))
    return n_mfma, bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "sdf_refine.s")
        compile_isa(path)
    n, bad = check(path)
    print("%d asm MFMAs, %d reads of in-flight accumulators without wait states" % (n, len(bad)))
    for ln, txt in bad[:10]:
        print("  line %d: %s" % (ln, txt))
    sys.exit(1 if bad or n == 0 else 0)
