"""Idle time between consecutive kernels of each HIP queue in a rocprofv3 kernel trace (rocpd sqlite): per queue the
busy time, the summed gaps below a cut-off (dependent launches back to back) and the largest gaps.
usage: stream_gaps.py <results.db> [t_from_ms t_to_ms]   (window relative to the first kernel)"""
import sqlite3
import sys

def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for cut in ("(", "<"):
        if cut in n[1:]:
            n = n[:n.index(cut, 1)]
    return n[-44:]


con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
ncol = next((c for c in ("kernel_name", "name", "kernel") if c in cols), None)
if ncol is None:
    print("columns of `kernels`:", cols)
    sys.exit(1)
rows = list(con.execute(f"select start, end, {ncol}, {qcol or 0} from kernels order by start"))
t0 = rows[0][0]
lo, hi = (float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6) if len(sys.argv) > 3 else (0, 1e18)
rows = [(s - t0, e - t0, n, q) for s, e, n, q in rows if lo <= s - t0 < hi]
print(f"{len(rows)} kernels in the window, span {(rows[-1][1] - rows[0][0]) / 1e3:.1f} us, columns: {cols}")
by_q = {}
for r in rows:
    by_q.setdefault(r[3], []).append(r)
for q, ks in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in ks)
    gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
    small = [g for g in gaps if 0 <= g[0] < 30e3]
    print(f"queue {q}: {len(ks)} kernels, busy {busy / 1e3:.1f} us, gaps < 30 us: {len(small)} summing {sum(g[0] for g in small) / 1e3:.1f} us "
          f"(median {sorted(g[0] for g in small)[len(small) // 2] / 1e3 if small else 0:.2f} us), overlapping launches {sum(1 for g in gaps if g[0] < 0)}")
    for g, a, b in sorted(gaps, key=lambda x: -x[0])[:6]:
        print(f"     {g / 1e3:8.1f} us  after {a[:50]}  before {b[:50]}")

    import collections
    pair = collections.defaultdict(lambda: [0, 0.0])
    for g, a, b in small:
        k = (short(a), short(b))
        pair[k][0] += 1
        pair[k][1] += g
    print("   gaps < 30 us by kernel pair (count, total us, mean us):")
    for k, (c, t) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"     {c:4d} {t / 1e3:8.1f} {t / c / 1e3:6.2f}  {k[0]} -> {k[1]}")
    hist = collections.Counter(min(int(g[0] / 1e3), 29) for g in small)
    print("   histogram (us: count):", sorted(hist.items()))

if len(sys.argv) > 4 or len(sys.argv) == 3:
    pass
# the main queue's sequence inside the window, gaps above 4 us marked
mainq = max(by_q.items(), key=lambda kv: len(kv[1]))[1]
if len(sys.argv) > 3:
    prev = None
    for s_, e_, n_, q_ in mainq:
        gap = (s_ - prev) / 1e3 if prev is not None else 0.0
        print(f"{s_ / 1e3:10.1f} {(e_ - s_) / 1e3:8.1f} {'gap %6.1f' % gap if gap > 4 else '          '} {short(n_)}")
        prev = e_
