"""Timeline analysis of a rocprofv3 kernel trace: last `adam_pack`-to-`adam_pack` window = one step."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if (sys.argv[2] if len(sys.argv) > 2 else "adam_pack") in r["Kernel_Name"]]
k = int(sys.argv[4]) if len(sys.argv) > 4 else 1
per = int(sys.argv[5]) if len(sys.argv) > 5 else 1          # marker occurrences per step
marks = marks[per - 1::per]
a, b = marks[-1 - k], marks[-k]
win = rows[a + 1:b + 1]
t0, t1 = win[0]["s"], win[-1]["e"]
print(f"step window {(t1 - t0) / 1e3:.1f} us, {len(win)} kernels")
# union busy
ev = sorted([(r["s"], 1) for r in win] + [(r["e"], -1) for r in win])
busy = 0; depth = 0; last = t0; conc = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    conc[depth] += t - last
    depth += d; last = t
print(f"union busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us; time by concurrency:", {k: round(v / 1e3, 1) for k, v in sorted(conc.items())})
byq = collections.defaultdict(float)
for r in win: byq[r["Queue_Id"]] += r["e"] - r["s"]
print("busy per queue (us):", {k: round(v / 1e3, 1) for k, v in byq.items()})
# gaps on the main queue
mainq = max(byq, key=byq.get)
mq = [r for r in win if r["Queue_Id"] == mainq]
gaps = []
for p, n in zip(mq, mq[1:]):
    g = n["s"] - p["e"]
    if g > 0: gaps.append((g, p["Kernel_Name"][:40], n["Kernel_Name"][:40]))
print(f"main queue: {len(mq)} kernels, sum gaps {sum(g for g, *_ in gaps) / 1e3:.1f} us, mean gap {sum(g for g, *_ in gaps) / max(1, len(gaps)) / 1e3:.2f} us")
for g, p, n in sorted(gaps, reverse=True)[:12]:
    print(f"  gap {g / 1e3:7.1f} us after {p} before {n}")
if len(sys.argv) > 3:
    for r in win:
        print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r['Queue_Id']} {re.sub(r'_ZN12_GLOBAL__N_1..', '', r['Kernel_Name'])[:60]}")
