#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r04; mkdir -p $OUT; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/s2_pytest.txt; tail -6 $OUT/s2_pytest.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $OUT/s2_bench.json 2> $OUT/s2_bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r04/s2_bench.json"))
print({k:d[k] for k in ("value","ms_per_step","ms_target_build","ms_align")})
hc=d["host_cloud"]; print({k:v for k,v in hc.items() if k not in ("what","pcl_registration")})
for k,v in d["configs"].items(): print(k, json.dumps(v)[:1500])
P
