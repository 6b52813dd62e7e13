"""profiles/traffic_k_derivatives.json from the newest profiles/rNN_pmc_summary.txt.

    python scripts/traffic_from_pmc.py [--write]

HBM bytes per launch of the headline kernel = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes): on gfx950
FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, section HBM; calibrated in
round 1 on k_cell_keys, which reads 12,000,000 B and reports 5872 KB).  bench.py quotes the JSON as a
STATIC figure (the PMC pass is a separate rocprofv3 run); tests/test_abi_cpu.py checks that the JSON and
the newest summary agree, so the quotation cannot drift from the measurement it cites.
"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "k_derivatives<false, 1, 1, false>"


def newest_summary():
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.txt"))]
    if not files:
        raise SystemExit("no profiles/rNN_pmc_summary.txt")
    return max(files, key=lambda f: int(re.search(r"r(\d+)_pmc_summary", f).group(1)))


def counters(path, kernel=KERNEL):
    """Mean per-dispatch value of every counter the summary holds for `kernel`."""
    out = {}
    for line in open(path):
        if not line.startswith("  " + kernel + " "):
            continue
        for name, val, n in re.findall(r"(\w+)=([0-9.e+\-]+)\(n=(\d+)\)", line):
            if name not in out:  # the first section of a summary is the C3 headline run
                out[name] = float(val)
                out["n_" + name] = int(n)
    return out


def traffic(path):
    c = counters(path)
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        raise SystemExit("%s holds no FETCH_SIZE / WRITE_SIZE line for %s" % (path, KERNEL))
    return {
        "kernel": KERNEL,
        "workload": "C3 200k->1M, 0.5 m (bench.py under rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes)",
        "FETCH_SIZE_KB": c["FETCH_SIZE"],
        "WRITE_SIZE_KB": c["WRITE_SIZE"],
        "correction": "gfx950 FETCH_SIZE counts half the bytes of coalesced reads (MI355X_MICROARCH.md, HBM): doubled; "
                      "calibrated in round 1 on k_cell_keys which reads 12,000,000 B and reports 5872 KB",
        "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
        "source": "profiles/%s (%s, %d launches)" % (os.path.basename(path), KERNEL, c.get("n_FETCH_SIZE", 0)),
        "TCC_HIT": c.get("TCC_HIT_sum"),
        "TCC_MISS": c.get("TCC_MISS_sum"),
    }


if __name__ == "__main__":
    t = traffic(newest_summary())
    print(json.dumps(t, indent=1))
    if "--write" in sys.argv:
        json.dump(t, open(os.path.join(ROOT, "profiles", "traffic_k_derivatives.json"), "w"), indent=1)
