"""Ad-hoc: timeline of a block-mode run from a rocprofv3 --kernel-trace csv: the persistent launches and what lies between them.
python tests/trace_timeline.py <dir with *_kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
t0 = rows[0][0]
multi = [r for r in rows if "pipeline2" in r[2]]
print(f"{len(rows)} kernels, {len(multi)} persistent launches, span {(rows[-1][1] - t0) / 1e6:.1f} ms")
prev_end = t0
for s, e, n in multi:
    between = [r for r in rows if r[0] >= prev_end and r[1] <= s and "pipeline2" not in r[2]]
    busy = sum(r[1] - r[0] for r in between)
    print(f"gap {(s - prev_end) / 1e6:8.1f} ms ({len(between)} kernels, sum of their durations {busy / 1e6:8.1f} ms) | persistent launch {(e - s) / 1e6:9.1f} ms")
    prev_end = e
tail = [r for r in rows if r[0] >= prev_end]
print(f"after the last: {(rows[-1][1] - prev_end) / 1e6:.1f} ms, {len(tail)} kernels")
tot = {}
for s, e, n in rows:
    a = tot.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:42s} {c:6d} launches {d / 1e6:10.1f} ms")
