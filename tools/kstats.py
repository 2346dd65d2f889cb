"""Per-kernel calls / average / total from a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [steps] [min_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0
floor = float(sys.argv[3]) if len(sys.argv) > 3 else 0
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    avg = float(r['AverageNs']) / 1e3
    per = float(r['TotalDurationNs']) / 1e3 / steps if steps else avg
    tot += float(r['TotalDurationNs']) / 1e3 / (steps or 1)
    if per >= floor:
        extra = f' {per:9.1f} us/step {int(r["Calls"]) / steps:5.1f} calls/step' if steps else ''
        print(f'{avg:9.1f} us avg x{int(r["Calls"]):6d}{extra}  {r["Name"][:120]}')
print(f'total {tot:.1f} us' + ('/step' if steps else ''))
