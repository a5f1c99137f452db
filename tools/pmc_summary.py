"""Per-kernel sums of rocprofv3 --pmc CSV output: python tools/pmc_summary.py <dir> <launches per kernel>"""
import csv, glob, collections, sys
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in sorted(glob.glob(sys.argv[1] + "/*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-56:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k, {a: round(b / div) for a, b in sorted(v.items())})
