"""CPU (hipcc cross-compiles): register-spill budget of the decoder kernels, read from the code object metadata of the compiled
ISA.  A tile kernel of this library owns a whole SIMD's register file (one wave of 512 registers on the split-fp16 pipe, two of 256
on the others); what the allocator cannot place goes to scratch memory -- private_segment_fixed_size bytes per lane, written and
read back once per tile.  Round 2 shipped k_mlp_jtj_h2<2> with 274 spilled VGPRs / 516 B per lane (5.2 GB of scratch writes per C4
launch) without anybody noticing; this test fails when a change pushes a kernel over its budget.  Budgets = what the current source
compiles to (profiles/r04_isa_budget.txt) plus a little slack; lower them when a kernel improves."""
import re
import subprocess

# kernel (demangled prefix) -> (max spilled VGPRs, max scratch bytes per lane, max spilled SGPRs)
BUDGET = {
    "qsp::k_mlp_jtj_h2<2, 4, false>": (32, 128, 2),      # (round 4: one SGPR spilled to a VGPR lane by the packed-triangle epilogue)
    "qsp::k_mlp_jtj_h2<1, 4, false>": (0, 0, 2),
    "qsp::k_mlp_jtj_h2<2, 8, false>": (32, 128, 0),
    "qsp::k_mlp_jtj_h2<1, 8, false>": (0, 0, 0),
    "qsp::k_mlp_jtj_h2<2, 8, true>": (72, 200, 0),       # the NARROW forms: runtime slab counts and skipped slots cost the allocator
    "qsp::k_mlp_jtj_h2<1, 8, true>": (8, 32, 0),         # some of its footing; they run a fraction of the full shape's work
    "qsp::k_mlp_fwd_h2<2, false, 4>": (24, 96, 0),
    "qsp::k_mlp_fwd_h2<2, true, 8>": (64, 160, 0),
    "qsp::k_mlp_fwd_h1<4>": (0, 0, 0),
    "qsp::k_mlp_fwd_h1<8>": (20, 80, 0),
    "qsp::k_decode_screen<4>": (0, 0, 0),
    "qsp::k_decode_screen<8>": (20, 80, 0),
    "qsp::k_decode_h2<false, false>": (24, 96, 0),
    "qsp::k_decode_h2<true, false>": (40, 160, 0),
    "qsp::k_decode_h2<false, true>": (64, 160, 0),
    "qsp::k_decode_h2<true, true>": (80, 200, 4),
    "qsp::k_mlp_fwd<false>": (0, 0, 0),
    "qsp::k_mlp_fwd<true>": (48, 192, 0),
    "qsp::k_mlp_jtj<false>": (72, 232, 184),             # (round 4: the packed-triangle store of the partial sums walks ONE running
    "qsp::k_mlp_jtj<true>": (76, 308, 172),              #  index -- sixteen independent per-lane offsets cost 25 more spilled VGPRs here)
    "qsp::k_decode<false, false>": (0, 0, 0),
    "qsp::k_decode<true, false>": (0, 0, 0),
    "qsp::k_decode<false, true>": (36, 136, 0),
    "qsp::k_decode<true, true>": (40, 160, 0),
}


def kernel_metadata(path):
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
        body = m.group(2)
        short = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        short = short.replace("void ", "", 1).split("(")[0]

        def g(k):
            return int(re.search(r"\.%s:\s+(\d+)" % k, body).group(1))
        out[short] = dict(vgpr=g("vgpr_count"), vspill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"),
                          scratch=g("private_segment_fixed_size"))
    return out


def test_decoder_kernels_stay_inside_their_spill_budget(sdf_isa):
    meta = kernel_metadata(sdf_isa)
    missing = [k for k in BUDGET if k not in meta]
    assert not missing, (missing, sorted(meta))
    over = {k: (meta[k]["vspill"], meta[k]["scratch"], meta[k]["sspill"]) for k, b in BUDGET.items()
            if meta[k]["vspill"] > b[0] or meta[k]["scratch"] > b[1] or meta[k]["sspill"] > b[2]}
    assert not over, "over budget (spilled VGPRs, scratch bytes per lane, spilled SGPRs): %r" % over
    # every k_mlp_* / k_decode* kernel of the file has a budget: a new tile kernel must be entered here
    unbudgeted = [k for k in meta if re.match(r"qsp::k_(mlp|decode)", k) and k not in BUDGET]
    assert not unbudgeted, unbudgeted
