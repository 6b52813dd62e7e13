"""FETCH_SIZE per k_derivatives launch of a `tools/size_sweep.py` run collected under `rocprofv3 --kernel-trace --pmc FETCH_SIZE`:
    python tools/sweep_fetch.py <dir with *counter_collection.csv>
Launches are told apart by their grid (Grid_Size / Workgroup_Size); the bytes are FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM:
gfx950 counts half the bytes of wide coalesced reads; KB -> bytes)."""
import collections, csv, glob, sys
root = sys.argv[1]
acc = collections.OrderedDict()
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "k_derivatives" not in r["Kernel_Name"] or r["Counter_Name"] != "FETCH_SIZE":
            continue
        key = (int(r["Grid_Size"]), int(r["Workgroup_Size"]), r["Kernel_Name"].split("k_derivatives")[1][:24])
        acc.setdefault(key, []).append(float(r["Counter_Value"]))
for (grid, wg, tmpl), v in acc.items():
    v = v[len(v) // 4:]   # the first launches of a size warm the Infinity Cache
    print("grid %9d threads (%5d blocks x %4d) %-24s launches %3d  FETCH_SIZE x2 = %8.2f MB per launch"
          % (grid, grid // wg, wg, tmpl, len(v), 2.0 * 1024.0 * sum(v) / len(v) / 1e6))
