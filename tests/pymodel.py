"""Pure-Python, string-level model of the reflexible extend path (small cases only).

An independent second restatement used to cross-check oracle/reflexiv_oracle.c:
records are (key_str, marker, ext_str, left, right) with plain ACGT strings, so
none of the 2-bit word packing of the C oracle is shared.  Follows SURVEY.md
Appendix B (B.0 order contract, B.3-B.6) and the same reference lines
(P/ReflexivMain.java / P/ReflexivDSMain.java) as the C oracle.
"""
from __future__ import annotations

NUC = "ACGT"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}
BLOCK = object()


def decode_kmer(x: int, k: int) -> str:
    return "".join(NUC[(x >> (2 * (k - 1 - i))) & 3] for i in range(k))


def revcomp(s: str) -> str:
    return "".join(COMP[c] for c in reversed(s))


def partition_starts(keys, P):
    n = len(keys)
    st, prev = [], 0
    for p in range(P):
        s = max((p * n) // P, prev)
        while 0 < s < n and keys[s] == keys[s - 1]:
            s += 1
        st.append(s)
        prev = s
    st.append(n)
    return st


def stable_sort(recs):
    return sorted(recs, key=lambda r: r[0])          # Python's sort is stable


def rc_expand(kmers, counts, k):
    out = []
    for x, c in zip(kmers, counts):
        s = decode_kmer(x, k)
        for t in (s, revcomp(s)):
            out.append((t[:-1], 1, t[-1], c, c))
    return out


def fork_forward(recs, starts, sub, min_err, ds):
    free = (lambda cov: -1 - cov) if (ds and min_err) else (lambda cov: -1)
    out, ostarts = [], []
    for p in range(len(starts) - 1):
        ostarts.append(len(out))
        first = len(out)
        for key, mk, ext, left, right in recs[starts[p]:starts[p + 1]]:
            if len(out) == first or out[-1][0] != key:
                out.append([key, mk, ext, left, free(left)])
                continue
            h = out[-1]
            cs, ch = left, h[3]
            if cs > ch:
                err = min_err and ch <= min_err and cs >= 2 * ch
                out[-1] = [key, mk, ext, cs, free(cs) if err else sub]
            elif cs == ch:
                if NUC.index(ext) > NUC.index(h[2]):
                    out[-1] = [key, mk, ext, cs, sub]
                else:
                    h[4] = sub
            else:
                err = min_err and cs <= min_err and ch >= 2 * cs
                h[4] = free(ch) if err else sub
    ostarts.append(len(out))
    return [tuple(r) for r in out], ostarts


def reflect(recs):
    return [(key[1:] + ext, 2, key[0], left, right) for key, mk, ext, left, right in recs]


def fork_reflected(recs, starts, sub, min_err, ds):
    ds_ec = ds and bool(min_err)
    out, ostarts = [], []
    for p in range(len(starts) - 1):
        ostarts.append(len(out))
        first = len(out)
        last_cov = 0
        for key, mk, ext, left, right in recs[starts[p]:starts[p + 1]]:
            cs = left
            if len(out) == first or out[-1][0] != key:
                last_cov = cs
                out.append([key, mk, ext, (-1 - cs) if ds_ec else -1, right])
                continue
            h = out[-1]
            if cs > last_cov:
                err = min_err and last_cov <= min_err and cs >= 2 * last_cov
                last_cov = cs
                out[-1] = [key, mk, ext, ((-1 - cs) if ds_ec else -1) if err else sub, right]
            elif cs == last_cov:
                # a tie always goes to the later record (4|base vs 1): ReflexivMain.java:2648-2653
                out[-1] = [key, mk, ext, sub, right]
            else:
                err = min_err and cs <= min_err and last_cov >= 2 * cs
                if err:
                    if not ds_ec:
                        h[3] = -1
                else:
                    h[3] = sub
    ostarts.append(len(out))
    return [tuple(r) for r in out], ostarts


def seq_of(r):
    return r[0] + r[2] if r[1] == 1 else r[2] + r[0]


def oriented(seq, m, sub, left, right):
    if m == 1:
        return (seq[:sub], 1, seq[sub:], left, right)
    return (seq[-sub:], 2, seq[:-sub], left, right)


def random_reflection(recs, starts, sub):
    out = []
    for p in range(len(starts) - 1):
        m = 2
        for r in recs[starts[p]:starts[p + 1]]:
            out.append(oriented(seq_of(r), m, sub, r[3], r[4]))
            m = 3 - m
    return out


def extend_pass(recs, starts, sub, rdd=False, start_marker=2):
    out, ostarts = [], []
    for p in range(len(starts) - 1):
        ostarts.append(len(out))
        m = start_marker
        holder = None

        def flip(r):
            nonlocal m
            out.append(oriented(seq_of(r), m, sub, r[3], r[4]))
            m = 3 - m
        for s in recs[starts[p]:starts[p + 1]]:
            if holder is None:
                holder = s
                continue
            if s[0] != holder[0]:
                flip(holder)
                holder = s
                continue
            if s[1] == holder[1]:
                flip(s)
                continue
            F, R = (s, holder) if s[1] == 1 else (holder, s)
            a, b = F[3], R[4]
            lf, lr = len(F[2]), len(R[2])
            if (a < 0 and b < 0) or (a >= 0 and b >= 0):
                d = -1
            elif s is F:
                d = a - lr if (a >= 0 and a - lr >= 0) else (b - lf if (b >= 0 and b - lf >= 0) else BLOCK)
            else:
                if b >= 0 and b - lf >= 0:
                    d = b - lf
                elif rdd:
                    d = F[4] - lr if (a >= 0 and F[4] - lr >= 0) else BLOCK
                else:
                    d = a - lr if (a >= 0 and a - lr >= 0) else BLOCK
            if d is BLOCK:
                flip(s)
                continue
            seq = R[2] + R[0] + F[2]
            if d < 0:
                L, Rt = R[3], F[4]
            elif F[3] > 0:
                L, Rt = d, F[4]
            else:
                L, Rt = R[3], d
            out.append(oriented(seq, m, sub, L, Rt))
            m = 3 - m
            holder = None
        if holder is not None:
            flip(holder)
    ostarts.append(len(out))
    return out, ostarts


def assemble(kmers, counts, k=31, P=4, min_err=8, min_iter=15, max_iter=150, ds=True, trace=None, coalesce=False):
    sub = k - 1
    part = [P]                      # the loop below may coalesce it (P/ReflexivMain.java:277-281)
    recs = stable_sort(rc_expand(kmers, counts, k))
    recs, _ = fork_forward(recs, partition_starts([r[0] for r in recs], P), sub, min_err, ds)
    recs = stable_sort(reflect(recs))
    recs, st = fork_reflected(recs, partition_starts([r[0] for r in recs], P), sub, min_err, ds)
    recs = random_reflection(recs, st, sub)

    def one_pass(recs):
        recs = stable_sort(recs)
        out, _ = extend_pass(recs, partition_starts([r[0] for r in recs], part[0]), sub, rdd=not ds)
        if trace is not None:
            trace.append(len(out))
        return out
    it = 0
    recs = one_pass(recs)
    for _ in range(3):
        it += 1
        recs = one_pass(recs)
    it += 1
    recs = one_pass(recs)
    last = 0
    partition_number = P
    while it <= max_iter:
        it += 1
        if it >= min_iter and it % 3 == 0:
            if last == len(recs):
                break
            last = len(recs)
            if coalesce and partition_number >= 16 and len(recs) // partition_number <= 20:
                partition_number = partition_number // 4 + 1
                part[0] = partition_number
        recs = one_pass(recs)
    return recs


# ---- k > 31: P/ReflexivDSMain64.java assemblyFromKmer (:374-826) without the extras of :584-619 / :672-712

def rc_expand_str(kmers, counts):
    """kmers as ACGT strings (the CSV rows KmerBinarizer reads)."""
    out = []
    for s, c in zip(kmers, counts):
        for t in (s, revcomp(s)):
            out.append((t[:-1], 1, t[-1], c, c))
    return out


def contigs_text_w(recs, min_contig):
    """DSKmerToContig + TagRowContigID of ReflexivDSMain64 (:830-892): '>Contig-<len>-<idx>', 100 columns."""
    out, idx = [], 0
    for r in recs:
        s = seq_of(r)
        if len(s) < min_contig:
            continue
        out.append(f">Contig-{len(s)}-{idx}\n" + "\n".join(s[i:i + 100] for i in range(0, len(s), 100)) + "\n")
        idx += 1
    return "".join(out), idx


# from-counts extras of ReflexivDSMain64 (:584-619, :672-712), one function per operator class; records are
# (key_str, marker, ext_str, left, right), inputs sorted by key, `starts` = logical partition starts

def double_w(recs, sub):
    """DSReflexivAndForwardKmer: every record, then the same sequence keyed at its other end"""
    out = []
    for r in recs:
        out.append(r)
        out.append(oriented(seq_of(r), 3 - r[1], sub, r[3], r[4]))
    return out


def _can_merge(s, h):
    """the four conditions under which the extend pass would join current `s` and holder `h` (opposite markers)"""
    a, b = (s[3], h[4]) if s[1] == 1 else (s[4], h[3])
    return (a < 0 and b < 0) or (a >= 0 and b >= 0) or (a >= 0 and a - len(h[2]) >= 0) or (b >= 0 and b - len(s[2]) >= 0)


def key_filter_w(op, recs, starts, sub):
    """op: 'pairs' DSFilterExtendableKmerPairs, 'unext' DSFilterUnExtendableKmer, 'first' ...StillExtendableKmerFromPairs,
    'longer' ...StillExtendableKmerEnds"""
    out = []
    fwd = lambda r: oriented(seq_of(r), 1, sub, r[3], r[4])
    for p in range(len(starts) - 1):
        holder = None
        for s in recs[starts[p]:starts[p + 1]]:
            if holder is None:
                holder = s
                continue
            h = holder
            if s[0] != h[0]:
                if op != "pairs":
                    out.append(h)
                holder = s
            elif op == "first":
                out.append(h)
                holder = None
            elif op == "longer":
                score = lambda r: len(r[2]) * 31 + ((len(r[2]) - 1) % 31 + 1)
                out.append(h if score(h) >= score(s) else s)
                holder = None
            elif op == "pairs":
                if s[1] != h[1] and _can_merge(s, h):
                    f, r = (s, h) if s[1] == 1 else (h, s)
                    out.append(f)               # the forward member as it is ...
                    out.append(fwd(r))          # ... the reflected one turned forward
                    holder = None
                else:
                    holder = s
            else:                               # 'unext'
                if s[1] != h[1] and _can_merge(s, h):
                    holder = None
                else:
                    out.append(fwd(h) if h[1] == 2 else h)
                    holder = s
        if holder is not None:
            out.append(holder)
    return out


def flip_all_w(recs, m, sub):
    return [oriented(seq_of(r), m, sub, r[3], r[4]) for r in recs]


def assemble_w(kmers, counts, k, P=4, min_err=8, min_iter=15, max_iter=150, trace=None, extras=False):
    """kmers: ascending ACGT strings.  Stop rule of :621-661: checks from min_iter + 3 on, the first repeat
    of the count switches param.scramble 2 -> 3 (every later pass starts its marker at 1), the second stops;
    the survivors are sorted by key before they become text (:714)."""
    sub = k - 1
    recs = stable_sort(rc_expand_str(kmers, counts))
    recs, _ = fork_forward(recs, partition_starts([r[0] for r in recs], P), sub, min_err, True)
    recs = stable_sort(reflect(recs))
    recs, st = fork_reflected(recs, partition_starts([r[0] for r in recs], P), sub, min_err, True)
    recs = random_reflection(recs, st, sub)

    def one_pass(recs, start=2):
        recs = stable_sort(recs)
        out, _ = extend_pass(recs, partition_starts([r[0] for r in recs], P), sub, rdd=False, start_marker=start)
        if trace is not None:
            trace.append(len(out))
        return out
    it = 0
    recs = one_pass(recs)
    for _ in range(3):
        it += 1
        recs = one_pass(recs)
    it += 1
    recs = one_pass(recs)
    last, scramble = 0, 2
    unext = None
    ps = lambda rr: partition_starts([r[0] for r in rr], P)
    while it <= max_iter:
        it += 1
        if extras and it == min_iter + 3:
            both = stable_sort(double_w(stable_sort(recs), sub))
            pe = stable_sort(key_filter_w("pairs", both, ps(both), sub))
            pu = stable_sort(key_filter_w("unext", both, ps(both), sub))
            recs = key_filter_w("first", pe, ps(pe), sub)
            unext = key_filter_w("first", pu, ps(pu), sub)
        if it >= min_iter + 3 and it % 3 == 0:
            if last == len(recs):
                if scramble == 2:
                    scramble = 3
                else:
                    break
            last = len(recs)
        recs = one_pass(recs, 1 if scramble == 3 else 2)
    if unext is not None:
        recs = recs + unext
        for m in (1, 2):
            recs = stable_sort(flip_all_w(recs, m, sub))
            recs = key_filter_w("longer", recs, ps(recs), sub)
    return stable_sort(recs)
