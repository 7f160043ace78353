"""Condense gpurun_out/prof_<round>/ (written by tests/prof_run.sh on the GPU box) into profiles/."""
import csv, glob, hashlib, json, os, shutil, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{R}"
os.makedirs("profiles", exist_ok=True)
newest = lambda pat: sorted(glob.glob(pat, recursive=True), key=os.path.getmtime)[-1]
ks = newest(f"{src}/trace/**/*_kernel_stats.csv")
shutil.copy(ks, f"profiles/{R}_kernel_stats.csv")
def csrc_sha():
    """what the library was built from: bench.py refuses a summary whose kernels are not the tree's"""
    h = hashlib.sha256()
    for f in sorted(glob.glob("nlzm_amd/csrc/*.h") + glob.glob("nlzm_amd/csrc/*.hip")):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


out = {"round": R, "command": "python3 bench.py --no-cpu --block-streams 0 --no-workloads", "kernel": "pipeline2_kernel", "csrc_sha16": csrc_sha()}
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
# calibration (tests/microbench/random_access.hip): what the counters report for N random 8-byte reads / 4-byte stores in 4 GiB
try:
    cal = {}
    n_acc = None
    for line in open(f"{src}/cal_fetch.log"):
        if line.startswith('{"accesses_per_kernel"'):
            n_acc = json.loads(line)["accesses_per_kernel"]
    for name, ctr, kern in (("cal_fetch", "FETCH_SIZE", "random_reads8"), ("cal_write", "WRITE_SIZE", "random_writes4")):
        f = newest(f"{src}/{name}/**/*_counter_collection.csv")
        v = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr)
        cal[ctr] = {"kernel": kern, "accesses": n_acc, "counter_KiB": v, "counter_bytes": v * 1024,
                    "bytes_per_access": v * 1024 / n_acc if n_acc else None}
    out["calibration_random_access_4GiB"] = cal
    # which reading of FETCH_SIZE applies to this kernel: if a random 8-byte read counts ~64 B the raw figure is whole lines
    # (no doubling); ~32 B would mean the guide's x2 applies to narrow reads as well
    bpa = cal["FETCH_SIZE"]["bytes_per_access"]
    if bpa and fs and ws:
        factor = 64.0 / bpa
        out["traffic_bytes_per_launch"]["fetch_calibrated"] = fs["per_launch"] * 1024 * factor
        out["traffic_bytes_per_launch"]["total_calibrated"] = fs["per_launch"] * 1024 * factor + ws["per_launch"] * 1024
        out["traffic_bytes_per_launch"]["calibration_note"] = (f"a random 8-byte read in 4 GiB reads {bpa:.1f} B of FETCH_SIZE: line-granular traffic = raw x {factor:.2f}; "
                                                              "this kernel's reads are such reads, so total_calibrated is the figure stood behind")
except Exception as e:      # (no calibration pass in this round's directory)
    out["calibration_random_access_4GiB"] = {"error": str(e)}
json.dump(out, open(f"profiles/{R}_pmc_summary.json", "w"), indent=1)
print(open(f"profiles/{R}_kernel_stats.csv").read()[:1500])
print(json.dumps(out.get("traffic_bytes_per_launch")))
