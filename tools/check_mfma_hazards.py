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


def check_valu_def_before_mfma(path):
    """Second software-only rule: a VALU instruction that writes a register (VGPR operand A / B, or an accumulator AGPR through
    v_accvgpr_write) must be at least 2 wait states ahead of the MFMA that reads it."""
    ins = []
    in_asm = False
    for i, l in enumerate(open(path).read().split("\n")):
        t = l.strip()
        if "#ASMSTART" in t:
            in_asm = True
            continue
        if "#ASMEND" in t:
            in_asm = False
            continue
        if not t or t[0] in ";./" or t.endswith(":"):
            continue
        ins.append((i + 1, t, in_asm))
    mf = re.compile(r"v_mfma_f32_32x32x2_f32 a\[(\d+):(\d+)\], v(\d+), v(\d+), a\[")

    def defs(t):
        if t.startswith("v_accvgpr_write"):
            m = re.match(r"v_accvgpr_write_b32 a(\d+)", t)
            return ("a", {int(m.group(1))}) if m else ("a", set())
        if not t.startswith("v_") or t.startswith("v_mfma") or t.startswith("v_cmp"):
            return ("v", set())
        m = re.match(r"v_\S+\s+v\[(\d+):(\d+)\]", t)
        if m:
            return ("v", set(range(int(m.group(1)), int(m.group(2)) + 1)))
        m = re.match(r"v_\S+\s+v(\d+)", t)
        return ("v", {int(m.group(1))}) if m else ("v", set())

    bad = []
    for k, (ln, t, asm) in enumerate(ins):
        m = mf.search(t)
        if not (m and asm):
            continue
        use_a = set(range(int(m.group(1)), int(m.group(2)) + 1))
        use_v = {int(m.group(3)), int(m.group(4))}
        ws, j = 0, k - 1
        while j >= 0 and ws < 2:
            pt = ins[j][1]
            mm = re.match(r"s_nop (\d+)", pt)
            if mm:
                ws += int(mm.group(1)) + 1
            else:
                kind, regs = defs(pt)
                if regs & (use_a if kind == "a" else use_v):
                    bad.append((ln, t, ins[j][0], pt))
                    break
                ws += 1
            j -= 1
    return bad


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
    bad2 = check_valu_def_before_mfma(path)
    print("%d MFMAs read a register a VALU instruction wrote fewer than 2 wait states earlier" % len(bad2))
    for b in bad2[:10]:
        print("  line %d: %s   <- line %d: %s" % b)
    sys.exit(1 if bad or bad2 or n == 0 else 0)
