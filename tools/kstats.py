"""prints the per-kernel summary of a rocprofv3 --stats output directory"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-42s calls=%5s avg_us=%10.1f total_ms=%9.2f" % (r["Name"][:42], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
