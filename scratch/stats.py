import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms  ({tot/1e6/steps:.3f} ms/step over {steps} steps)")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r["Name"]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = n.split("(")[0][:58]
    print(f"{n:58s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/1e6:8.3f} ms  per-step {float(r['TotalDurationNs'])/1e6/steps:7.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
