"""Step-time model from two kernel traces of the bench step (tests/probes/r06_first.sh: `trace_dump.py` CSVs):
  side   = the shipped schedule (weight-gradient / reduction work on two side queues),
  noside = the same kernels on ONE queue, i.e. every kernel alone on the chip: its ISOLATED duration.
Per optimiser step (boundaries = the embedding launch) it emits, for every queue, busy / idle time; for every main-queue kernel
its in-step and isolated duration and the side-queue kernels that ran during it; and predictions of the step time:
  identity   = sum(main in-step durations) + main-queue gaps                    (the decomposition itself)
  serial     = sum(isolated durations of ALL kernels) + main-queue gaps         (no overlap at all = the one-queue step)
  conserving = serial - hidden, hidden = side work that ran while the main queue was idle or beside kernels that did not slow down
usage: critical_path.py <side.csv.gz> <noside.csv.gz> <out.json>"""
import collections
import csv
import gzip
import json
import re
import sys


def load(path):
    rows = []
    with gzip.open(path, "rt") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["start_ns"]), int(r["end_ns"]), r["queue"], r["name"]))
    return rows


def short(n):
    n = n.replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:64]


def steps_of(rows):
    bounds = [i for i, r in enumerate(rows) if "embed_fwd_kernel" in r[3]]
    return [rows[a:b] for a, b in zip(bounds[:-1], bounds[1:])]


def span(sts):
    firsts = [st[0][0] for st in sts]
    return (firsts[-1] - firsts[0]) / (len(firsts) - 1)


side, noside = load(sys.argv[1]), load(sys.argv[2])
S, N = steps_of(side)[3:], steps_of(noside)[3:]          # (skip warm-up steps: allocator growth, first-use initialisation)
mainq = collections.Counter(r[2] for r in side).most_common(1)[0][0]

iso = collections.defaultdict(list)          # isolated duration of the k-th launch of a kernel name inside a step
for st in N:
    occ = collections.Counter()
    for s, e, q, n in st:
        iso[(short(n), occ[short(n)])].append(e - s)
        occ[short(n)] += 1
iso = {k: sum(v) / len(v) for k, v in iso.items()}

per_step = []
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, collections.Counter()])
sagg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for st in S:
    occ = collections.Counter()
    main = [r for r in st if r[2] == mainq]
    sides = [r for r in st if r[2] != mainq]
    qstat = {q: {"kernels": sum(1 for r in st if r[2] == q), "busy_us": sum(r[1] - r[0] for r in st if r[2] == q) / 1e3}
             for q in sorted(set(r[2] for r in st))}
    ident = sum(e - s for s, e, _, _ in main)
    gaps = sum(max(0, main[i + 1][0] - main[i][1]) for i in range(len(main) - 1))
    iso_main = iso_side = excess = 0.0
    for s, e, q, n in st:
        k = (short(n), occ[short(n)])
        occ[short(n)] += 1
        i_us = iso.get(k, e - s)
        if q != mainq:
            iso_side += i_us
            a = sagg[short(n)]
            a[0] += 1; a[1] += e - s; a[2] += i_us
            continue
        iso_main += i_us
        ov = collections.Counter()
        for s2, e2, q2, n2 in sides:
            o = min(e, e2) - max(s, s2)
            if o > 0:
                ov[short(n2)] += o
        a = agg[short(n)]
        a[0] += 1; a[1] += e - s; a[2] += i_us
        for kk, vv in ov.items():
            a[3][kk] += vv
        excess += (e - s) - i_us
    per_step.append({"queues": qstat, "main_in_step_us": ident / 1e3, "main_gaps_us": gaps / 1e3,
                     "main_isolated_us": iso_main / 1e3, "side_isolated_us": iso_side / 1e3, "main_excess_us": excess / 1e3})

nst = len(per_step)
mean = lambda k: sum(p[k] for p in per_step) / nst
measured = span(S) / 1e3
out = {
    "source": "rocprofv3 --kernel-trace of `bench.py --steps 6 --warmup 3 --no-graph` with and without --no-side-stream "
              "(tests/probes/r06_first.sh, tests/probes/critical_path.py)",
    "steps_averaged": nst, "main_queue": mainq,
    "measured_step_us": measured, "measured_step_one_queue_us": span(N) / 1e3,
    "queues": {q: {"kernels_per_step": sum(p["queues"].get(q, {}).get("kernels", 0) for p in per_step) / nst,
                   "busy_us": sum(p["queues"].get(q, {}).get("busy_us", 0.0) for p in per_step) / nst,
                   "idle_us": measured - sum(p["queues"].get(q, {}).get("busy_us", 0.0) for p in per_step) / nst}
               for q in sorted(set(r[2] for r in side))},
    "main_in_step_us": mean("main_in_step_us"), "main_gaps_us": mean("main_gaps_us"),
    "main_isolated_us": mean("main_isolated_us"), "side_isolated_us": mean("side_isolated_us"),
    "main_excess_us": mean("main_excess_us"),
}
out["predicted"] = {
    "identity_us": out["main_in_step_us"] + out["main_gaps_us"],
    "serial_us": out["main_isolated_us"] + out["side_isolated_us"] + out["main_gaps_us"],
}
out["hidden_side_work_us"] = out["predicted"]["serial_us"] - out["predicted"]["identity_us"]
rows = []
for n, (c, t_in, t_iso, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    rows.append({"kernel": n, "launches_per_step": c / nst, "in_step_us_per_step": t_in / 1e3 / nst, "isolated_us_per_step": t_iso / 1e3 / nst,
                 "in_step_avg_us": t_in / 1e3 / c, "isolated_avg_us": t_iso / 1e3 / c,
                 "overlapped_by_us_per_step": {k: round(v / 1e3 / nst, 1) for k, v in ov.most_common(3)}})
out["main_queue_kernels"] = rows
out["side_queue_kernels"] = [{"kernel": n, "launches_per_step": c / nst, "in_step_us_per_step": t_in / 1e3 / nst,
                              "isolated_us_per_step": t_iso / 1e3 / nst}
                             for n, (c, t_in, t_iso) in sorted(sagg.items(), key=lambda kv: -kv[1][1])]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.endswith("_kernels")}, indent=1))
print(f"{'kernel':58s} {'n':>4s} {'in-step':>9s} {'isolated':>9s} {'avg in':>8s} {'avg iso':>8s}  overlapped by")
for r in rows[:26]:
    print(f"{r['kernel'][:58]:58s} {r['launches_per_step']:4.0f} {r['in_step_us_per_step']:9.1f} {r['isolated_us_per_step']:9.1f} "
          f"{r['in_step_avg_us']:8.1f} {r['isolated_avg_us']:8.1f}  {r['overlapped_by_us_per_step']}")
print("side queues:")
for r in out["side_queue_kernels"][:10]:
    print(f"{r['kernel'][:58]:58s} {r['launches_per_step']:4.0f} {r['in_step_us_per_step']:9.1f} {r['isolated_us_per_step']:9.1f}")
