"""Ad-hoc probe (not a test): SHA-256 and size of every root of corpus.real_text (and of the sub-directories of the last one), to see which of
them are the same on two boxes."""
import hashlib, os, sys
sys.path.insert(0, '.')
from nlzm_amd import corpus
def digest(root, exts):
    h, n, nf = hashlib.sha256(), 0, 0
    for d, dirs, files in os.walk(root):
        dirs.sort()
        for f in sorted(files):
            p = os.path.join(d, f)
            if f.endswith(exts) and not os.path.islink(p):
                try:
                    b = open(p, 'rb').read()
                except OSError:
                    continue
                h.update(b); n += len(b); nf += 1
    return nf, n, h.hexdigest()[:16]
for root, exts in corpus.REAL_ROOTS:
    if os.path.isdir(root):
        print(root, *digest(root, exts), flush=True)
root, exts = corpus.REAL_ROOTS[-1]
for sub in sorted(os.listdir(root)):
    p = os.path.join(root, sub)
    if os.path.isdir(p):
        nf, n, hx = digest(p, exts)
        if n > 3_000_000:
            print("  ", sub, nf, n, hx, flush=True)
