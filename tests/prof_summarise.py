"""Condense gpurun_out/prof_<round>/ (written by tests/prof_run.sh on the GPU box) into profiles/."""
import csv, glob, json, os, shutil, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{R}"
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]
ks = newest(f"{src}/trace/**/*_kernel_stats.csv")
shutil.copy(ks, f"profiles/{R}_kernel_stats.csv")
out = {"round": R, "command": "python3 bench.py --no-cpu --block-streams 0", "kernel": "pipeline2_kernel"}
for name in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = newest(f"{src}/{name}/**/*_counter_collection.csv")
    agg, n = {}, {}
    for r in csv.DictReader(open(f)):
        if "pipeline2_kernel" in r["Kernel_Name"]:
            k = r["Counter_Name"]
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
            n[k] = n.get(k, 0) + 1
    for k, v in agg.items():
        out[k] = {"sum": v, "launches": n[k], "per_launch": v / n[k]}
for line in open(f"{src}/bench_trace.log"):
    if line.startswith('{"metric"'):
        out["bench_line_under_profiler"] = json.loads(line)
fs, ws = out.get("FETCH_SIZE"), out.get("WRITE_SIZE")
if fs and ws:
    # MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads exactly half of
    # a wide coalesced streaming read.  This kernel's reads are mostly narrow random accesses (uncalibrated), so both
    # the raw and the doubled figure are kept; WRITE_SIZE is exact for 16-B streaming stores.
    out["traffic_bytes_per_launch"] = {"fetch_raw": fs["per_launch"] * 1024, "fetch_x2": fs["per_launch"] * 2048,
                                       "write": ws["per_launch"] * 1024,
                                       "total_with_fetch_x2": fs["per_launch"] * 2048 + ws["per_launch"] * 1024}
json.dump(out, open(f"profiles/{R}_pmc_summary.json", "w"), indent=1)
print(open(f"profiles/{R}_kernel_stats.csv").read()[:1500])
print(json.dumps(out.get("traffic_bytes_per_launch")))
