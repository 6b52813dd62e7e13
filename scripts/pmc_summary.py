"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel dispatch."""
import csv, glob, re, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
for f in sorted(glob.glob(root + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+(<[^>]*>)?|rocprim[^<]*<[^,]*|__amd_rocclr_\w+)", r["Kernel_Name"])
        name = m.group(1)[-52:] if m else r["Kernel_Name"][:52]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f)
    for k, d in acc.items():
        if k.startswith("k_"):
            print("  %-52s" % k, "  ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(d.items())))
