"""Same input through several builds of the library (NLZM_LIB), one process each: python tests/gpu_variants.py <size> <window> <kind> [key=value ...] -- <lib>..."""
import os, subprocess, sys
args = sys.argv[1:]
cut = args.index("--")
head, libs = args[:cut], args[cut + 1:]
for lib in libs:
    env = dict(os.environ, NLZM_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "tests/gpu_opt.py"] + head, env=env, capture_output=True, text=True)
    keep = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(("cycles/position", "finder: blocks that", "worker lanes: ", "hot bins' waves: ")) or l[:1].isdigit()]
    print("==", os.path.basename(lib), " ".join(head))
    print("\n".join(keep), flush=True)
