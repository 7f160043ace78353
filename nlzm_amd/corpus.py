"""Seeded synthetic corpora: stand-ins for enwik8/enwik9/calgary.tar.

None of the real corpora exist on the build or GPU boxes (no network), so every
benchmark and test input comes from here.  The generators are deterministic for a
given (kind, size, seed) on any machine (numpy PCG64 streams only).

``syn_text``   Zipf(s=1.1) draws over a 50,000-word vocabulary of seeded 2-12 letter
               lowercase words; separators {" " 80 %, ", " 8 %, ". " 8 %, "\\n" 4 %};
               then spans of 300-200,000 bytes (log-uniform) are overwritten by copies
               of earlier spans (about 8 % of the bytes: long-range / RK256 regime) and
               runs of 70-5,000 identical bytes are written over about 1 % of them
               (nice-length skip, BT4 early-return path).  SURVEY.md's per-word
               probabilities (0.002 / 0.0005) would make the file >90 % copies, so the
               rates are scaled to those coverages instead.
``random``     uniform random bytes (expansion path).
``runs``       long byte runs and short periodic patterns.
``mixed``      text, binary records, runs and duplicated blocks interleaved (the
               calgary.tar stand-in).

If ``$NLZM_CORPUS_DIR`` holds the real file (enwik8, enwik9, calgary.tar) ``load()``
returns that instead.
"""
from __future__ import annotations

import os

import numpy as np

SEED = 0x4E4C5A4D

SIZES = {
    "calgary": 3_152_896,
    "enwik8": 100_000_000,
    "enwik9": 1_000_000_000,
}


def _vocab(rng: np.random.Generator, nwords: int = 50_000):
    lens = rng.integers(2, 13, size=nwords)
    # letter frequencies roughly English-like so hashes of 2/3/4-grams are skewed
    letters = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    w = 1.0 / np.arange(1, 27) ** 0.8
    w /= w.sum()
    flat = letters[rng.choice(26, size=int(lens.sum()), p=w)]
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    return flat, starts.astype(np.int64), lens.astype(np.int64)


_SEPS = [b" ", b", ", b". ", b"\n"]
_SEP_P = np.array([0.80, 0.08, 0.08, 0.04])


def _text_block(rng, vocab, cdf, nbytes: int) -> np.ndarray:
    """About nbytes of words+separators (>= nbytes, caller trims)."""
    flat, starts, lens = vocab
    ntok = int(nbytes / 6.5) + 64
    tok = np.searchsorted(cdf, rng.random(ntok), side="right").astype(np.int32)
    sep = rng.choice(4, size=ntok, p=_SEP_P)
    sep_len = np.where((sep == 0) | (sep == 3), 1, 2).astype(np.int32)
    wl = lens[tok].astype(np.int32)
    tot = wl + sep_len
    ends = np.cumsum(tot, dtype=np.int64)
    total = int(ends[-1])
    begs = ends - tot
    out = np.empty(total, dtype=np.uint8)
    # word bytes: out[begs[i] + k] = flat[starts[tok[i]] + k], k < wl[i]
    nw = int(wl.sum())
    wbeg = np.cumsum(wl, dtype=np.int64) - wl
    k = np.arange(nw, dtype=np.int64) - np.repeat(wbeg, wl)
    out[np.repeat(begs, wl) + k] = flat[np.repeat(starts[tok], wl) + k]
    sp = begs + wl
    out[sp] = np.array([0x20, 0x2C, 0x2E, 0x0A], dtype=np.uint8)[sep]
    two = sep_len == 2
    out[sp[two] + 1] = 0x20
    return out


def syn_text(size: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.default_rng(seed)
    vocab = _vocab(rng)
    p = 1.0 / np.arange(1, len(vocab[1]) + 1) ** 1.1
    cdf = np.cumsum(p / p.sum())
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    blk = 8 << 20
    while pos < size:
        b = _text_block(rng, vocab, cdf, min(blk, size - pos))
        k = min(len(b), size - pos)
        out[pos:pos + k] = b[:k]
        pos += k
    # overwrite events, in increasing position order
    nwords = size / 7.5
    ncopy = int(nwords * 2e-5) + 1
    nrun = int(nwords * 3e-5) + 1
    ev_pos = np.sort(rng.integers(0, max(size, 1), size=ncopy + nrun))
    ev_kind = rng.random(ncopy + nrun) < (ncopy / max(1, ncopy + nrun))
    # span lengths: log-uniform so that most copies are short and a few are huge
    ev_len_copy = np.exp(rng.uniform(np.log(300), np.log(200_000), size=ncopy + nrun)).astype(np.int64)
    ev_len_run = rng.integers(70, 5001, size=ncopy + nrun)
    ev_src = rng.random(ncopy + nrun)
    ev_byte = rng.integers(0, 256, size=ncopy + nrun)
    for i in range(ncopy + nrun):
        q = int(ev_pos[i])
        if ev_kind[i]:
            ln = int(min(ev_len_copy[i], size - q, q))
            if ln < 300:
                continue
            src = int(ev_src[i] * (q - ln + 1))
            out[q:q + ln] = out[src:src + ln]
        else:
            ln = int(min(ev_len_run[i], size - q))
            out[q:q + ln] = ev_byte[i]
    return out


def random_bytes(size: int, seed: int = SEED) -> np.ndarray:
    return np.random.default_rng(seed ^ 0x5EED).integers(0, 256, size=size, dtype=np.uint8)


def runs(size: int, seed: int = SEED) -> np.ndarray:
    rng = np.random.default_rng(seed ^ 0x52554E)
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    while pos < size:
        kind = rng.integers(0, 3)
        ln = int(min(size - pos, rng.integers(1, 3000)))
        if kind == 0:
            out[pos:pos + ln] = rng.integers(0, 256)
        elif kind == 1:
            per = int(rng.integers(2, 9))
            pat = rng.integers(0, 256, size=per, dtype=np.uint8)
            out[pos:pos + ln] = np.resize(pat, ln)
        else:
            out[pos:pos + ln] = rng.integers(97, 123, size=ln, dtype=np.uint8)
        pos += ln
    return out


def mixed(size: int, seed: int = SEED) -> np.ndarray:
    """calgary.tar stand-in: text files, binary-ish records, runs, duplicated members."""
    rng = np.random.default_rng(seed ^ 0xCA16A7)
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    member = 0
    while pos < size:
        ln = int(min(size - pos, rng.integers(20_000, 400_000)))
        kind = member % 5
        if kind in (0, 1):
            b = syn_text(ln, seed + 17 * member + 1)
        elif kind == 2:
            # fixed-width binary records with small-alphabet fields
            rec = rng.integers(0, 16, size=(ln // 16 + 1, 16), dtype=np.uint8)
            rec[:, 0] = 0xFF
            rec[:, 4:8] = (np.arange(rec.shape[0])[:, None] >> (8 * np.arange(4))) & 0xFF
            b = rec.reshape(-1)[:ln]
        elif kind == 3:
            b = runs(ln, seed + 31 * member)
        else:
            src = int(rng.integers(0, max(1, pos - ln))) if pos > ln else 0
            b = out[src:src + ln].copy() if pos > ln else random_bytes(ln, seed + member)
        out[pos:pos + ln] = b[:ln]
        # tar-like 512-byte header of zeros and a name
        hdr = min(512, size - pos)
        out[pos:pos + hdr] = 0
        name = (b"member%04d.dat" % member)[:hdr]
        out[pos:pos + len(name)] = np.frombuffer(name, dtype=np.uint8)
        pos += ln
        member += 1
    return out


def dups(size: int, seed: int = SEED) -> np.ndarray:
    """Long duplicated spans: incompressible blocks repeated at distances that exercise the
    RK256 finder (>= 256 B), its carried match and the uint16 truncation of its length cap
    (spans >= 65,536 B), separated by text."""
    rng = np.random.default_rng(seed ^ 0xD0B5)
    out = np.empty(size, dtype=np.uint8)
    blocks = [random_bytes(int(k), seed + i + 1) for i, k in enumerate((70_000, 300, 5_000, 66_000, 1_000))]
    pos = 0
    i = 0
    while pos < size:
        if i % 2 == 0:
            b = syn_text(int(rng.integers(2_000, 40_000)), seed + 100 + i)
        else:
            b = blocks[int(rng.integers(0, len(blocks)))]
            off = int(rng.integers(0, 3)) * 7
            b = b[off:]
        k = min(len(b), size - pos)
        out[pos:pos + k] = b[:k]
        pos += k
        i += 1
    return out


def chains(size: int, seed: int = SEED) -> np.ndarray:
    """Parse segments that run into the forced cut at 4,096 positions (NLZM.cpp:1469) or stop just short of it.

    A segment only ends where no match crosses (or at the forced cut), so each unit is: a pool of 40-byte pieces of a
    random string B taken every 30 bytes (every position of B is then inside some match: the segment cannot end), B
    itself, of a length a little below / at / above 4,096, two fresh bytes (a natural end inside the last 264
    positions before the forced cut when B is short of it), and a 300-byte copy of earlier bytes (matches of the full
    264 that the segment before them may only see cut short, the next one whole)."""
    rng = np.random.default_rng(seed ^ 0xC4A1)
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    unit = 0
    while pos < size:
        nb = (3_900, 3_990, 4_050, 4_096, 4_200, 3_840, 8_100)[unit % 7] + int(rng.integers(0, 40))
        b = rng.integers(0, 256, nb, dtype=np.uint8)
        parts = []
        for k in range(0, nb - 10, 30):
            parts.append(b[k:k + 40])
            parts.append(rng.integers(0, 256, 3, dtype=np.uint8))
        src = rng.integers(0, 256, 300, dtype=np.uint8)
        parts += [src, rng.integers(0, 256, 5, dtype=np.uint8), b, rng.integers(0, 256, 2, dtype=np.uint8), src,
                  rng.integers(0, 256, 7, dtype=np.uint8)]
        u = np.concatenate(parts)
        k = min(u.size, size - pos)
        out[pos:pos + k] = u[:k]
        pos += k
        unit += 1
    return out


def cutnice(size: int, seed: int = SEED) -> np.ndarray:
    """A nice region (carried match >= 64, NLZM.cpp:1514) that starts exactly where a segment is cut at 4,096 positions
    (:1469), one position before it, one after it, ...

    Each unit: 300 fresh bytes `src`; a pool of 40-byte pieces of a random string b taken every 30 bytes (so inside b every
    position lies in some short match and the segment that starts with b cannot end); b itself, 4,095 + off bytes long,
    followed directly by a copy of `src`: the first match of 65 bytes or more is found at b's position 4,095 + off, i.e.
    the nice region starts at the segment's position 4,096 + off.  With off = 0 no edge of the segment spans its cut and
    the next segment starts inside a region whose phase (:1529) depends on where it started."""
    rng = np.random.default_rng(seed ^ 0xC07A1CE)
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    unit = 0
    while pos < size:
        nb = 4_095 + (0, -1, 1, 0, 2, 0, -2)[unit % 7]
        src = rng.integers(0, 256, 300, dtype=np.uint8)
        b = rng.integers(0, 256, nb, dtype=np.uint8)
        bb = np.concatenate([b, src])
        parts = [src, rng.integers(0, 256, 4, dtype=np.uint8)]
        for k in range(0, nb, 30):
            parts.append(bb[k:k + 40])
            parts.append(rng.integers(0, 256, 3, dtype=np.uint8))
        parts += [rng.integers(0, 256, 5, dtype=np.uint8), b, src, rng.integers(0, 256, 7, dtype=np.uint8)]
        u = np.concatenate(parts)
        k = min(u.size, size - pos)
        out[pos:pos + k] = u[:k]
        pos += k
        unit += 1
    return out


def xml_like(size: int, seed: int = SEED) -> np.ndarray:
    """Markup shaped like a wiki dump (what enwik8 / enwik9 are): <page> records with a title, numeric ids, a timestamp, a contributor and
    a text body of Zipf words with [[links]], ''emphasis'' and {{templates}} -- fixed tag strings that repeat every few hundred bytes at
    slowly varying distances (rep-distance commands, NLZM.cpp:1253-1272, 1344-1367), runs of digits, and free text in between."""
    rng = np.random.default_rng(seed ^ 0x3C3F786D)
    flat, starts, lens = _vocab(rng, 8000)
    words = [bytes(flat[s:s + l]) for s, l in zip(starts, lens)]
    p = 1.0 / np.arange(1, len(words) + 1) ** 1.05
    cdf = np.cumsum(p / p.sum())
    parts, total, pid = [], 0, 1000
    while total < size:
        nw = int(rng.integers(20, 400))
        tok = np.searchsorted(cdf, rng.random(nw), side="right")
        body = []
        for t in tok:
            w = words[min(int(t), len(words) - 1)]
            r = rng.random()
            if r < 0.04: w = b"[[" + w + b"]]"
            elif r < 0.06: w = b"''" + w + b"''"
            elif r < 0.07: w = b"{{" + w + b"|" + words[int(rng.integers(0, 50))] + b"}}"
            elif r < 0.09: w = w + b".\n"
            body.append(w)
        pid += int(rng.integers(1, 40))
        ts = b"20%02d-%02d-%02dT%02d:%02d:%02dZ" % (int(rng.integers(1, 7)), int(rng.integers(1, 13)), int(rng.integers(1, 29)),
                                                    int(rng.integers(0, 24)), int(rng.integers(0, 60)), int(rng.integers(0, 60)))
        rec = (b"  <page>\n    <title>" + b" ".join(words[int(x)] for x in rng.integers(0, 3000, size=int(rng.integers(1, 4)))) + b"</title>\n    <id>%d</id>\n"
               b"    <revision>\n      <id>%d</id>\n      <timestamp>" % (pid, pid * 7 + int(rng.integers(0, 1000))) + ts + b"</timestamp>\n"
               b"      <contributor>\n        <username>" + words[int(rng.integers(0, 500))] + b"</username>\n        <id>%d</id>\n      </contributor>\n"
               b"      <text xml:space=\"preserve\">" % int(rng.integers(1, 99999)) + b" ".join(body) + b"</text>\n    </revision>\n  </page>\n")
        parts.append(rec)
        total += len(rec)
    return np.frombuffer(b"".join(parts), dtype=np.uint8)[:size].copy()


# ---- real text: source files that are part of the image (the same on the build container and on the GPU boxes) ------------------
REAL_ROOTS = [("/usr/lib/python3.10", (".py",)), ("/usr/lib/python3/dist-packages", (".py",)), ("/opt/rocm/include", (".h", ".hpp")),
              ("/usr/include", (".h", ".hpp"))] + [
    # (the installed packages that the build container and the GPU boxes both have, named one by one: the container holds more)
    ("/usr/local/lib/python3.10/dist-packages/" + pkg, (".py", ".pyi", ".h", ".hpp", ".rst"))
    for pkg in ("torch", "sympy", "pandas", "scipy", "plotly", "triton", "numpy", "sqlalchemy", "matplotlib", "networkx", "pygments", "fontTools", "libcst")]


def real_text(size: int, seed: int = SEED) -> np.ndarray:
    """The first `size` bytes of the image's own source files, concatenated: Python's standard library, the distribution's Python
    packages, the ROCm and system C / C++ headers, the sources of the installed Python packages (PyTorch and the rest) -- directory by directory in the order of REAL_ROOTS,
    inside a directory by sorted path, every file once.  Real text with real skew (indentation runs, licence headers repeated in every
    file, templated C++), where the synthetic stand-ins are Zipf words: the hottest BT4 head of Python source holds 14 % of the
    positions.  The bytes depend on the image: callers pin them by SHA-256 (tests/golden/real.json) and
    skip, loudly, where the files differ.  `seed` is ignored."""
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    for root, exts in REAL_ROOTS:
        if pos >= size or not os.path.isdir(root):
            continue
        for d, dirs, files in os.walk(root):
            dirs.sort()
            for f in sorted(files):
                if not f.endswith(exts):
                    continue
                path = os.path.join(d, f)
                if os.path.islink(path):
                    continue
                try:
                    b = np.fromfile(path, dtype=np.uint8)
                except OSError:
                    continue
                k = min(b.size, size - pos)
                out[pos:pos + k] = b[:k]
                pos += k
                if pos >= size:
                    return out
    if pos < size:
        raise RuntimeError(f"real_text: the image holds only {pos} bytes of source files, {size} were asked for")
    return out


def block_set(size: int, seed: int = SEED) -> np.ndarray:
    """The input of bench.py's block-mode leg in small: 32 independent texts of size/32 bytes each (block i is
    syn_text(size/32, seed + 100 + i), as there), back to back."""
    per = size // 32
    return np.concatenate([syn_text(per, seed + 100 + i) for i in range(32)])


def dense_text(size: int, seed: int = SEED) -> np.ndarray:
    """Text over a vocabulary of 40 words: after a few kilobytes every position lies inside some match, so that nearly every
    parse segment runs into the forced cut at 4,096 positions (NLZM.cpp:1469) -- what 300 MB of ordinary text look like to the
    parser at -window:28, in a few hundred kilobytes (the helper parser's cases: DESIGN.md section 11)."""
    rng = np.random.default_rng(seed ^ 0xDE75E)
    vocab = _vocab(rng, 40)
    p = 1.0 / np.arange(1, 41) ** 0.9
    cdf = np.cumsum(p / p.sum())
    out = np.empty(size, dtype=np.uint8)
    pos = 0
    while pos < size:
        b = _text_block(rng, vocab, cdf, min(1 << 20, size - pos))
        k = min(len(b), size - pos)
        out[pos:pos + k] = b[:k]
        pos += k
    return out


def dense_breaks(size: int, seed: int = SEED) -> np.ndarray:
    """dense_text with a natural segment end every 1,500 - 9,000 bytes: three bytes that occur nowhere else (no match spans them), so
    that segments of every length follow segments that were cut -- among them ones that end inside the last 264 + 64 positions in front
    of a forced cut, where records were re-listed for a cut that then does not happen (the helper parser's give-up and drop paths)."""
    rng = np.random.default_rng(seed ^ 0xB4EA45)
    out = dense_text(size, seed)
    pos = int(rng.integers(1500, 9000))
    k = 0
    while pos + 3 < size:
        out[pos] = 0x80 + (k % 120); out[pos + 1] = 0x80 + ((k // 120) % 120); out[pos + 2] = 0x80 + ((k * 7) % 120)
        k += 1
        # (often close to a multiple of 4,096 behind the last break: a segment that ends just in front of, or just behind, a forced cut)
        step = int(rng.integers(1500, 9000))
        if rng.random() < 0.5: step = 4096 * int(rng.integers(1, 3)) + int(rng.integers(-330, 40))
        pos += max(step, 200)
    return out


def u16_cut(size: int, seed: int = SEED) -> np.ndarray:
    """Matches that RK256 finds where its compare is ended by the uint16 length parameter (NLZM.cpp:760, :1096) after 90 - 257 bytes:
    `la_end - p` is then just above 65,536, which happens at the chunk offsets 56,840 - 57,007 of every chunk (chunk_size 122,368, feed
    122,633).  Such an entry is shorter than the match it stands for, and when it is the table's longest entry the next position extends
    it again (:1503-1512) -- real text does this once in ~100 MB.  For RK256 to be the finder that meets the match, BT4
    must be off (a nice region of ANOTHER match A that is still >= 64 long where the match B begins) and the HT rows stale (the matches
    are over four symbols only: every 2- and 3-gram was seen a few bytes ago).  Per trial t: source B = Y + Z in chunk 0 at a place that
    puts one of its 256-aligned RK windows on the wanted position, source A = X + Y in chunk 1 (placed so that none of ITS aligned windows
    lies inside the match: RK256 would carry A over the hit and not probe, :1090), target X + Y + Z in chunk t + 2 so that B's window falls on
    an offset of the zone, 12 / 20 / 28 bytes behind the start of Y: A has 88 / 80 / 72 bytes left there (nice, and shorter than what the
    uint16 leaves of B), and the position is a multiple of 8 from the target's start, where its segment begins (between the matches the
    bytes match nothing, every segment is one literal) -- nice positions call the finders at every eighth position of the segment (:1514)."""
    return _u16_cut(size, seed, tie=False)


def _u16_cut(size: int, seed: int, tie: bool) -> np.ndarray:
    C, FEED = 122_368, 122_633
    rng = np.random.default_rng(seed ^ 0x16C07)
    sym = np.frombuffer(b"acgt", dtype=np.uint8)
    out = rng.integers(0, 256, size=size, dtype=np.uint8)
    trials = min(max(0, size // C - 3), 48)
    for t in range(trials):
        # tie: every second trial with B the NEARER source (it wins the table's end and grows).  B then lies behind A in the file, its Y is a
        # nice region of its own, and RK256 stores B's window under its own start only if that is a multiple of 8 from the start of Y (:1084-1087
        # store the CALLING position): lead 16 / 24 there, and X four bytes shorter so that the hit is a calling position of the target too
        swap = tie and t % 2 == 1
        xlen = 296 if swap else 300
        X = sym[rng.integers(0, 4, size=300)][:xlen]; Y = sym[rng.integers(0, 4, size=100)]; Z = sym[rng.integers(0, 4, size=900)]
        lead = 16 + 8 * ((t // 2) % 2) if swap else 12 + 8 * (t % 3)   # the hit lies that many bytes behind the start of Y in the target
        cap = 100 - lead if tie else 90 + (t * 37) % 168    # what the uint16 leaves of the lookahead at the hit (tie: exactly what A has left)
        hit = (t + 2) * C + (FEED - 65_536 - cap)           # absolute position of the RK256 hit
        tgt = hit - lead - xlen
        # B's window: 256-aligned start, in a chunk where every position calls; A placed so that its aligned windows start 200 bytes
        # into the match (none has 256 bytes of it)
        b_chunk, a_chunk = (1, 0) if swap else (0, 1)
        win = b_chunk * C + 256 * (40 + 8 * t) + (-(b_chunk * C)) % 256
        src_b = win - lead
        src_a = a_chunk * C + 2048 + 1024 * t + 56 + (-(a_chunk * C)) % 256
        out[src_b:src_b + 1000] = np.concatenate([Y, Z])
        out[src_a:src_a + xlen + 100] = np.concatenate([X, Y])
        out[tgt:tgt + xlen + 1000] = np.concatenate([X, Y, Z])
    return out


def u16_tie(size: int, seed: int = SEED) -> np.ndarray:
    """u16_cut with the cut-short RK256 entry ending exactly where the nice match A ends: both are the table's longest entries, and the
    one with the smaller distance is what the table's end holds (NLZM.cpp:835-852) -- only then does the next position extend the RK256
    entry.  Trials alternate: A the nearer source (the entry stays as it is), B the nearer source (it grows)."""
    return _u16_cut(size, seed, tie=True)


def splice(size: int, seed: int = SEED) -> np.ndarray:
    """Fuzz input for the parity harness: pieces of every kind the finders treat differently, spliced at random -- incompressible bytes, text over 4 - 16
    symbols, copies of earlier spans (distance and length log-uniform: 4 bytes to 3,000, one byte back to the start), near-copies (a copy with a few bytes
    changed: matches that end and resume), runs, and copies forced across the chunk offsets where RK256's uint16 length parameter wraps (56,800 - 57,100 of every
    122,368-byte chunk: section 12 of DESIGN.md).  No structure is intended; every seed is another arrangement."""
    rng = np.random.default_rng(seed ^ 0x5B11CE)
    out = np.empty(size + 8192, dtype=np.uint8)       # (a piece may start just below `size`: 3,000 bytes of padding and a copy of 2,500 at most)
    n = 0
    C = 122_368
    while n < size:
        kind = rng.random()
        if n < 64 or kind < 0.15:
            l = int(rng.integers(1, 200)); out[n:n + l] = rng.integers(0, 256, size=l, dtype=np.uint8)
        elif kind < 0.35:
            l = int(rng.integers(8, 400)); k = int(rng.integers(2, 17)); base = int(rng.integers(32, 200))
            out[n:n + l] = (base + rng.integers(0, k, size=l)).astype(np.uint8)
        elif kind < 0.45:
            l = int(rng.integers(2, 600)); out[n:n + l] = int(rng.integers(0, 256))
        else:
            l = int(np.exp(rng.uniform(np.log(4), np.log(3000))))
            d = int(np.exp(rng.uniform(0, np.log(n))))
            d = max(1, min(d, n))
            if kind > 0.9:      # put the copy across the zone of the nearest chunk, if there is room behind us for its source
                zone = (n // C) * C + int(rng.integers(56_700, 57_100))
                if zone > n and zone - n < 3000:
                    pad = zone - n - int(rng.integers(0, 300))
                    if pad > 0: out[n:n + pad] = rng.integers(0, 256, size=pad, dtype=np.uint8); n += pad
                    l = int(rng.integers(300, 2500))
            for i in range(0, l, max(d, 1)):        # (overlapping copies repeat, as LZ copies do)
                m = min(d, l - i); out[n + i:n + i + m] = out[n + i - d:n + i - d + m]
            if 0.75 < kind <= 0.9 and l > 16:
                for _ in range(int(rng.integers(1, 4))): out[n + int(rng.integers(0, l))] ^= int(rng.integers(1, 256))
        n += l
    return out[:size].copy()


def spines(size: int, seed: int = SEED) -> np.ndarray:
    """Positions with MANY BT4 record-setters of their own (NLZM.cpp:996-998): groups of K = 36 .. 120 strings that share ever longer prefixes with a
    query that follows them -- the farthest shares the most -- so that the query's descent meets them by increasing distance and every one of them is
    a longer match than the one before (none lies between the query and a later one in the tree's order: all are on the path).  A block set reserves
    32 pairs per position and takes the rest from an arena per launch (nlzm_core.h, bt_pair_ptr): these are the positions that use it, and with a
    small arena (option block_ext_blocks) the ones that use it up.  Between the groups: text over sixteen symbols."""
    rng = np.random.default_rng(seed ^ 0x5917E5)
    out = np.empty(size + 32768, dtype=np.uint8)
    n = 0
    while n < size:
        l = int(rng.integers(200, 3000))
        out[n:n + l] = (97 + rng.integers(0, 16, size=l)).astype(np.uint8); n += l
        K = int(rng.integers(36, 121))
        q = (97 + rng.integers(0, 26, size=K + 40)).astype(np.uint8)
        for k in range(K, 0, -1):                   # the farthest first: it shares 3 + K bytes with the query
            m = 3 + k
            out[n:n + m] = q[:m]; out[n + m] = 65 + (int(q[m]) - 97 + 1 + int(rng.integers(0, 24))) % 26; n += m + 1    # (an upper-case byte: never the query's)
            f = int(rng.integers(8, 40))
            out[n:n + f] = (48 + rng.integers(0, 10, size=f)).astype(np.uint8); n += f
        out[n:n + q.size] = q; n += q.size
    return out[:size].copy()


_GENS = {"spines": spines, "splice": splice, "u16_cut": u16_cut, "u16_tie": u16_tie, "xml_like": xml_like, "real_text": real_text, "dense_breaks": dense_breaks, "dense_text": dense_text, "syn_text": syn_text, "cutnice": cutnice, "block_set": block_set, "random": random_bytes, "runs": runs, "mixed": mixed, "dups": dups, "chains": chains}


def make(kind: str, size: int, seed: int = SEED) -> np.ndarray:
    return _GENS[kind](size, seed)


def load(name: str) -> tuple[np.ndarray, str]:
    """Return (bytes, provenance) for 'enwik8' | 'enwik9' | 'calgary'."""
    d = os.environ.get("NLZM_CORPUS_DIR")
    fname = {"calgary": "calgary.tar"}.get(name, name)
    if d and os.path.exists(os.path.join(d, fname)):
        return np.fromfile(os.path.join(d, fname), dtype=np.uint8), f"file:{fname}"
    size = SIZES[name]
    if name == "calgary":
        return mixed(size), "synthetic:mixed"
    return syn_text(size), "synthetic:syn_text"
