"""From a rocprofv3 --kernel-trace directory of tools/ba_only.py: the device timeline of the LAST local joint BA -- busy time,
idle gaps between consecutive kernels (and which kernel follows the longest ones), per-kernel totals.
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/ba_only.py c4 3 ;  python3 tools/ba_timeline.py DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in csv.DictReader(open(f))]
rows.sort()
# one BA = from a k_errors that follows a long pause to the next long pause (> 2 ms: host-side set_state / python)
groups, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 1500000:
        groups.append(cur); cur = []
    cur.append(r)
groups.append(cur)
g = [x for x in groups if len(x) > 50][-1]
span = max(e for _, e, _ in g) - g[0][0]
# kernels of two streams overlap (the chain factorisation and its tile workgroups): busy = the union of the intervals, a gap
# = from the latest end so far to the next start
gaps = collections.Counter(); gapn = collections.Counter()
busy, end = 0, g[0][0]
for s_, e_, k_ in g:
    gap = max(0, s_ - end)
    if s_ > g[0][0]:
        gaps[k_] += gap; gapn[k_] += 1
    busy += max(0, e_ - max(end, s_))
    end = max(end, e_)
print("kernels %d   span %.3f ms   busy %.3f ms   idle %.3f ms" % (len(g), span / 1e6, busy / 1e6, (span - busy) / 1e6))
print("idle time in front of each kernel (total us, count, mean us):")
for k, v in gaps.most_common(12):
    print("  %-28s %8.1f %4d %6.1f" % (k, v / 1e3, gapn[k], v / 1e3 / gapn[k]))
tot = collections.Counter()
for s, e, k in g: tot[k] += e - s
print("busy by kernel (us):", ", ".join("%s %.0f" % (k, v / 1e3) for k, v in tot.most_common(10)))
if len(sys.argv) > 2:      # the kernels of one LM trial in order: start (us from the trial's first kernel), duration, gap in front
    i0 = [i for i, r in enumerate(g) if r[2] == "k_trial_stage1"][int(sys.argv[2])]
    t0 = g[i0][0]
    for j in range(i0 - 3, i0 + 16):
        s_, e_, k_ = g[j]
        print("  %-24s start %8.1f  dur %6.1f  gap %6.1f" % (k_, (s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - g[j - 1][1]) / 1e3))
