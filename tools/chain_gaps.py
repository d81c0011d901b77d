"""Critical-path accounting from a rocprofv3 --kernel-trace CSV: for one factorisation step (32 potrf_diag
launches at N=4096) print the kernels of the critical-path queue, their durations and the idle gaps between them.
    python tools/chain_gaps.py <kernel_trace.csv> [step_index]"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:34], r["Queue_Id"]) for r in rows)
pot = [i for i, k in enumerate(ks) if "potrf" in k[2]]
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else 32
step = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sel = pot[nblk * step: nblk * (step + 1)]
t0 = ks[sel[0]][0]
print("step span us", (ks[sel[-1]][1] - t0) / 1e3)
qid = ks[sel[0]][3]
q1 = [k for k in ks[sel[0]: sel[-1] + 1] if k[3] == qid]
gaps = [(q1[i + 1][0] - q1[i][1]) / 1e3 for i in range(len(q1) - 1)]
durs = {}
for k in q1:
    durs.setdefault(k[2], []).append((k[1] - k[0]) / 1e3)
print("chain kernels", len(q1), "sum dur", round(sum(sum(v) for v in durs.values()), 1), "sum gaps", round(sum(gaps), 1),
      "median gap", statistics.median(gaps))
for n, v in durs.items():
    print(" ", n, len(v), round(sum(v) / len(v), 2))
for i in range(sel[10], sel[12] + 1):
    s, e, n, q = ks[i]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {n} q={q}")
