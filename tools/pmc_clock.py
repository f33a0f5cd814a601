"""From a rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ... run: per kernel mean duration, shader clock
(GRBM_GUI_ACTIVE / 8 XCDs / duration) and matrix-pipe busy fraction (MFMA busy cycles / 1024 SIMDs / active cycles)."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(acc.items()):
    if "GRBM_GUI_ACTIVE" not in c:
        continue
    n = len(c["GRBM_GUI_ACTIVE"])
    dur = sum(c["_dur_GRBM_GUI_ACTIVE"]) / n
    act = sum(c["GRBM_GUI_ACTIVE"]) / n / 8.0
    mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), 1)
    if dur > 1e6:
        print("%-28s n=%3d  %.2f ms  clock %.3f GHz  MFMA busy %.1f %%" % (k[-28:], n, dur / 1e6, act / dur, 100 * mf / 1024 / max(act, 1)))
