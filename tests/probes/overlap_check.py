"""From a rocprofv3 kernel trace: how much of the grouped weight-gradient GEMM's time runs beside other kernels, and the wall span
of a step (first kernel start to last kernel end), to compare eager launches with hipGraph replays."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
tn = [(s, e) for s, e, n in ev if "gemm_tn8" in n]
others = [(s, e) for s, e, n in ev if "gemm_tn8" not in n]
def overlap(a, bs):
    s0, e0 = a
    return sum(max(0, min(e0, e) - max(s0, s)) for s, e in bs if e > s0 and s < e0)
tot = sum(e - s for s, e in tn)
ov = sum(min(e - s, overlap((s, e), others)) for s, e in tn[-64:]) / max(1, sum(e - s for s, e in tn[-64:]))
print(f"gemm_tn8 launches {len(tn)}, mean {tot / max(1, len(tn)) / 1e3:.1f} us; fraction of its time with another kernel running (last 64): {ov:.2f}")
adam = [s for s, e, n in ev if "adam" in n]
if len(adam) >= 3:
    d = [(adam[i + 1] - adam[i]) / 1e6 for i in range(len(adam) - 1)]
    print("step periods (ms, adam to adam):", [round(x, 2) for x in d[-8:]])
