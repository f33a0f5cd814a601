"""From a rocprofv3 --kernel-trace directory of tools/lat_calls.py: the device timeline of ONE one-object call (the 10th of the
timed ones) -- every kernel and copy with its start, duration and the idle gap in front of it; busy and idle totals.
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/lat_calls.py fp16x2 ;  python3 tools/call_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in csv.DictReader(open(f))]
rows.sort()
# a call of the one-object pattern = 5 k_solve launches with no screening pass (k_mlp_fwd_h1) between them
solves = [i for i, r in enumerate(rows) if r[2].startswith("k_solve")]
h1 = [i for i, r in enumerate(rows) if "fwd_h1" in r[2]]
first_h1 = h1[0] if h1 else len(rows)
one = [i for i in solves if i < first_h1]
n_calls = len(one) // 5
c = min(12, n_calls - 2)
i_end = one[5 * c + 4]                       # last k_solve of call c
i_prev = one[5 * c - 1]                      # last k_solve of the call before
i_next = one[5 * (c + 1) + 4]
g = rows[i_prev + 1:i_end + 1]
t0 = rows[i_prev][1]
print("one-object call: from the end of the previous call's last kernel to the end of this call's last kernel: %.3f ms" % ((g[-1][1] - t0) / 1e6))
busy = sum(e - s for s, e, _ in g)
print("device busy %.3f ms, idle %.3f ms (of which in front of the first kernel: %.3f ms -- the host between two calls)" % (
    busy / 1e6, (g[-1][1] - t0 - busy) / 1e6, (g[0][0] - t0) / 1e6))
end = t0
for s, e, k in g:
    print("  %-34s start %8.1f  dur %6.1f  gap %6.1f" % (k[:34], (s - t0) / 1e3, (e - s) / 1e3, (s - end) / 1e3))
    end = e
