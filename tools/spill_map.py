"""Where a kernel's register spills sit: scratch stores / loads and MFMAs per window of its ISA listing, with the basic-block
labels of each window.  Usage: python tools/spill_map.py file.s <mangled-name-prefix> [window]"""
import re
import sys


def body_of(lines, prefix):
    start = [i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l and not l.startswith("\t")][0]
    end = [i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm")][0]
    return lines[start:end + 1]


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    body = body_of(open(path).read().split("\n"), prefix)
    st = [i for i, l in enumerate(body) if "scratch_store" in l]
    ld = [i for i, l in enumerate(body) if "scratch_load" in l]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    labels = [(i, l.split(":")[0]) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    print(len(body), "lines;", len(st), "scratch stores,", len(ld), "scratch loads,", len(mf), "MFMAs")
    for b in range(0, len(body), B):
        s = sum(1 for i in st if b <= i < b + B)
        l = sum(1 for i in ld if b <= i < b + B)
        m = sum(1 for i in mf if b <= i < b + B)
        lab = [x[1] for x in labels if b <= x[0] < b + B]
        if s or l or lab:
            print("%6d  st %3d  ld %3d  mfma %3d  %s" % (b, s, l, m, " ".join(lab)[:120]))


if __name__ == "__main__":
    main()
