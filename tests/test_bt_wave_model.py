"""A model of the scheme the hot bins' waves run (nlzm_amd/csrc/nlzm_kernels.hip, worker_role_hot):
consecutive BT4 calls of one hash head in flight at the same time, each trailing the calls before it down the tree, under
assumptions about undecided positions -- checked against the serial order of MatchFinderBT::FindAndUpdate (NLZM.cpp:978-1022).

The device code itself is checked on the GPU (tests/test_gpu_parity.py: bytes and counters against the reference, the
threshold forced down so that nearly every bin is run by a wave).  This file pins the ARGUMENT, on the CPU, in plain Python:
  * a call assigns every slot at most once and reads a slot only on its way down, so a later call may read every slot the
    earlier calls have finished with; a call marks the slot it takes, a call that needs a marked slot repeats the step;
  * one call starts per step; an undecided position that is assumed to be skipped is NOT made (round 6: no descent without
    stores) -- it holds a lane as a call that has ended, with no result and no notes;
  * results behind an undecided position are held; a wrong assumption takes every call behind the position back, latest
    first, from the (slot, value replaced) notes; a position that was assumed to be called has its own stores taken back and
    the bin goes on behind it, one that was assumed to be skipped is made now: the bin goes on AT it;
  * a decision may come before the position's result (the finder stage says "call" as soon as the positions in front are settled).
A step of the model is a step of the wave: every lane loads from the memory as the step finds it, then the stores of the
step are made (marks before links).  Lanes take steps in lockstep; which free lane a call starts on is random.
"""
import random

NONE, PENDING = -1, -2
MAXLEN, MAXTESTS = 24, 256


def match_len(buf, s, a, init, cap):
    l = init
    while l < cap and buf[s + l] == buf[a + l]:
        l += 1
    return l


def serial(buf, positions, called):
    """The reference order: position after position; a position that is not called reports its matches and stores nothing."""
    tree, head, results = {}, [NONE], {}
    for a in positions:
        stores, res = {}, []
        sp = head[0]
        stores["h"] = a
        pend_l, pend_r, len_l, len_r, tests, best = 2 * a, 2 * a + 1, 0, 0, 0, 1
        done = False
        while sp != NONE and tests < MAXTESTS:
            tests += 1
            pl, pr = tree.get(2 * sp, NONE), tree.get(2 * sp + 1, NONE)
            init = min(len_l, len_r)
            l = match_len(buf, sp, a, init, MAXLEN)
            if l > best:
                res.append((a - sp, l)); best = l
            if l >= MAXLEN:
                stores[pend_l], stores[pend_r] = pl, pr
                done = True
                break
            if buf[sp + l] < buf[a + l]:
                stores[pend_l] = sp; pend_l = 2 * sp + 1; len_r = l; sp = pr
            else:
                stores[pend_r] = sp; pend_r = 2 * sp; len_l = l; sp = pl
        if not done:
            stores[pend_r] = NONE; stores[pend_l] = NONE
        results[a] = (tuple(res), tests)
        if called[a]:
            head[0] = stores.pop("h")
            tree.update(stores)
    return tree, head[0], results


class Lane:
    def __init__(self):
        self.state = "idle"


def wave(buf, positions, marked, called, assume_skip, decide_at, rng, lanes_n=8):
    """The wave: returns the tree, the head, the results as they were published, and how many calls were made again."""
    tree, head = {}, [NONE]
    lanes = [Lane() for _ in range(lanes_n)]
    published, redone = {}, 0
    nxt, seq_next, step, recovering = 0, 1, 0, False
    # (a decision may precede the position's result: the finder stage does not wait for a result to say "call")
    decided = lambda a: step >= decide_at[a]

    def replay(notes):
        for slot, v in reversed(notes):
            if slot == "h":
                head[0] = v
            else:
                tree[slot] = v

    while nxt < len(positions) or any(L.state != "idle" for L in lanes):
        step += 1
        assert step < 200000, "the wave does not move"
        und = [L for L in lanes if L.state != "idle" and L.und]
        oseq = min((L.seq for L in und), default=1 << 60)
        wrong = [L for L in lanes if L.state != "idle" and L.wrong]
        if wrong:
            recovering = True
        if recovering and not any(L.state in ("start", "run") for L in lanes):
            rseq = min(L.seq for L in wrong)
            for L in sorted((L for L in lanes if L.state == "held" and L.seq > rseq), key=lambda L: -L.seq):
                if not L.dry:
                    replay(L.notes)
                L.state = "idle"; redone += 1
            P = next(L for L in lanes if L.state == "held" and L.seq == rseq)
            replay(P.notes)                  # assumed to be called, and skipped: the values its stores replaced (assumed to be skipped: no notes)
            P.state = "idle"
            nxt = P.index if P.dry else P.index + 1     # (assumed to be skipped, and called: made now)
            recovering = False
            continue
        # ---- one call starts per step (not behind a call without its stores that is on its way, not while recovering)
        free = [L for L in lanes if L.state == "idle"]
        if not recovering and free and nxt < len(positions):
            L = rng.choice(free)
            a = positions[nxt]
            L.__dict__.update(state="start", a=a, index=nxt, seq=seq_next, und=False, wrong=False, dry=False, notes=[],
                              res=[], best=1, tests=0, out=False)
            nxt += 1; seq_next += 1
        # ---- loads of this step (from the memory as the step finds it)
        loaded = {}
        for L in lanes:
            if L.state == "start":
                loaded[id(L)] = head[0]
            elif L.state == "run" and L.sp != NONE and L.tests < MAXTESTS:
                loaded[id(L)] = (tree.get(2 * L.sp, NONE), tree.get(2 * L.sp + 1, NONE))
        # ---- decisions that have come in
        for L in lanes:
            if L.state != "idle" and L.und and not L.wrong and decided(L.a):
                if called[L.a] == (not L.dry):
                    L.und = False
                else:
                    L.wrong = True
        # ---- what the loads say; the stores of the step (marks before links: the order matters only inside one lane)
        for L in lanes:
            if L.state == "start":
                a = L.a
                if marked[a] and decided(a) and not called[a]:
                    L.state = "idle"
                    continue
                if marked[a] and not decided(a):
                    L.und, L.dry = True, assume_skip[a]
                if L.dry:                                           # nothing is made of it: a call that has ended, without a result
                    L.state, L.out = "held", True
                    continue
                L.sp = loaded[id(L)]
                L.pend_l, L.pend_r, L.len_l, L.len_r = 2 * a, 2 * a + 1, 0, 0
                L.notes.append(("h", L.sp))
                tree[2 * a] = PENDING; tree[2 * a + 1] = PENDING
                head[0] = a
                L.state = "run"
            elif L.state == "run":
                a, sp = L.a, L.sp
                if sp == NONE or L.tests >= MAXTESTS:
                    fin = (NONE, NONE)
                else:
                    pl, pr = loaded[id(L)]
                    init = min(L.len_l, L.len_r)
                    l = match_len(buf, sp, a, init, MAXLEN)
                    full = l >= MAXLEN
                    right = (not full) and buf[sp + l] < buf[a + l]
                    if (full and PENDING in (pl, pr)) or (not full and (pr if right else pl) == PENDING):
                        continue                                    # held by an earlier call: the step is repeated
                    L.tests += 1
                    if l > L.best:
                        L.res.append((a - sp, l)); L.best = l
                    if full:
                        fin = (pl, pr)
                    else:
                        slot, taken, old = (L.pend_l, 2 * sp + 1, pr) if right else (L.pend_r, 2 * sp, pl)
                        L.notes.append((taken, old))
                        tree[taken] = PENDING
                        tree[slot] = sp
                        if right:
                            L.pend_l, L.len_r, L.sp = taken, l, pr
                        else:
                            L.pend_r, L.len_l, L.sp = taken, l, pl
                        continue
                tree[L.pend_l], tree[L.pend_r] = fin
                L.state = "held"
        # ---- a call that has ended: its result goes out when no undecided position stands before it
        for L in lanes:
            if L.state == "held" and not L.wrong:
                if not L.out and L.seq <= oseq and not L.dry:
                    assert L.a not in published or published[L.a] == (tuple(L.res), L.tests), "a published result changed"
                    published[L.a] = (tuple(L.res), L.tests); L.out = True
                if L.out and not L.und and L.seq <= oseq:
                    L.state = "idle"
    return tree, head[0], published, redone


def make_case(rng, n, alphabet, repeat):
    body = [rng.randrange(alphabet) for _ in range(n)]
    for _ in range(repeat):                                         # long repeats: deep common prefixes, full-length matches
        src, dst, ln = rng.randrange(n), rng.randrange(n), rng.randrange(8, 60)
        for k in range(ln):
            if src + k < n and dst + k < n:
                body[dst + k] = body[src + k]
    return body + [255] * (MAXLEN + 1)                              # (padding that matches nothing)


def run(seed, n=260, alphabet=2, repeat=6, accuracy=0.8, p_marked=0.25, lanes=8):
    rng = random.Random(seed)
    buf = make_case(rng, n, alphabet, repeat)
    positions = list(range(n))
    marked = {a: rng.random() < p_marked for a in positions}
    called = {a: (not marked[a]) or rng.random() < 0.5 for a in positions}
    assume_skip = {a: (not called[a]) if rng.random() < accuracy else called[a] for a in positions}
    t, decide_at = 0, {}
    for a in positions:                                             # decisions come in position order, some early, some late
        t += rng.choice((0, 1, 2, 3, 5, 9, 30))
        decide_at[a] = t
    want_tree, want_head, want_res = serial(buf, positions, called)
    tree, head, published, redone = wave(buf, positions, marked, called, assume_skip, decide_at, rng, lanes)
    assert head == want_head
    for a in positions:
        if called[a]:                                               # the slots of a position that was not inserted belong to no tree
            assert (tree.get(2 * a, NONE), tree.get(2 * a + 1, NONE)) == (want_tree.get(2 * a, NONE), want_tree.get(2 * a + 1, NONE)), a
    for a in positions:
        if a in published:                                          # (a position known to be skipped before its turn reports nothing)
            assert published[a] == want_res[a], a
        if called[a]:
            assert a in published, a
    return redone


def test_trailing_calls_give_the_serial_tree_and_results():
    """No undecided positions: pure pipelining (marks, repeated steps, one start per step)."""
    for seed in range(40):
        run(seed, p_marked=0.0, lanes=2 + seed % 7)


def test_assumptions_held_results_and_take_backs():
    """Undecided positions under right and wrong assumptions, early and late decisions."""
    redone = 0
    for seed in range(120):
        redone += run(1000 + seed, accuracy=0.5 + 0.5 * (seed % 3) / 2, p_marked=0.1 + 0.2 * (seed % 4), lanes=2 + seed % 9,
                      alphabet=2 + seed % 2)
    assert redone > 100                                             # (the take-back path was exercised)


def test_deep_trees_and_the_test_cap():
    """One symbol: every call runs into full-length matches; two symbols with long repeats: chains up to the cap."""
    for seed in range(10):
        run(2000 + seed, n=200, alphabet=1, repeat=0, p_marked=0.2)
        run(3000 + seed, n=400, alphabet=2, repeat=20, p_marked=0.2, lanes=16)
