"""Gaps between the kernels of a C3 step from a rocprofv3 kernel trace (tuning aid, not collected by pytest).
Usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv>   (trace made with: rocprofv3 --kernel-trace --output-format csv
-d DIR -- python3 tools/step_ab.py x, NDT_STEP_AB_DEFERRED=1)."""
import csv, glob, os, sys
import numpy as np
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
def short(n):
    for k in ("k_bucket_pass", "k_bucket_leaves"):
        if k in n: return k
    if "k_derivatives" in n:
        return "eval_prelaunched" if n.split("k_derivatives<")[1].split(">")[0].endswith("true") else "eval_ordinary"
    return "other"
ev = [(short(n), s, e) for n, s, e in rows]
# steps: from one k_bucket_pass to the next
idx = [i for i, r in enumerate(ev) if r[0] == "k_bucket_pass"]
steps = [ev[a:b] for a, b in zip(idx[:-1], idx[1:])]
steps = [s for s in steps if 20 < len(s) < 40][10:-2]
def med(v): return float(np.median(v)) / 1e3
out = {}
for s in steps:
    names = [r[0] for r in s]
    if names[1] != "k_bucket_leaves": continue
    out.setdefault("pass", []).append(s[0][2] - s[0][1])
    out.setdefault("pass -> leaves gap", []).append(s[1][1] - s[0][2])
    out.setdefault("leaves", []).append(s[1][2] - s[1][1])
    out.setdefault("leaves -> first evaluation gap", []).append(s[2][1] - s[1][2])
    out.setdefault("first evaluation (%s)" % s[2][0], []).append(s[2][2] - s[2][1])
    out.setdefault("first -> second evaluation: start to start", []).append(s[3][1] - s[2][1])
    out.setdefault("first end -> second end", []).append(s[3][2] - s[2][2])
    ends = [r[2] for r in s[2:] if r[0].startswith("eval")]
    out.setdefault("later evaluations: end to end", []).extend(np.diff(ends)[1:].tolist())
    out.setdefault("evaluations per step", []).append(len(ends) * 1000)
    out.setdefault("step: pass start -> last evaluation end", []).append(ends[-1] - s[0][1])
print("%d steps" % len(steps))
for k, v in out.items():
    print("  %-46s median %8.2f us   min %8.2f   max %8.2f" % (k, med(v), min(v) / 1e3, max(v) / 1e3))
