"""Print the top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run (kernel_stats.csv), per step."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%6.2f%% %7.1f calls/step avg %9.1f us  %7.3f ms/step  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]) / steps,
          float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps, r["Name"][:100]))
print("total kernel ms/step", tot / 1e6 / steps)
