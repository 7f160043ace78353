"""Same input through several builds of the library (NLZM_LIB), one process each: python tests/gpu_variants.py <size> <window> <kind> <lib>..."""
import os, subprocess, sys
size, hb, kind = sys.argv[1:4]
for lib in sys.argv[4:]:
    env = dict(os.environ, NLZM_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "tests/gpu_opt.py", size, hb, kind], env=env, capture_output=True, text=True)
    keep = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith(("cycles/position", "finder: blocks that", "worker lanes: ", "hot bins' waves: ")) or l[:1].isdigit()]
    print("==", os.path.basename(lib), size, hb, kind)
    print("\n".join(keep), flush=True)
