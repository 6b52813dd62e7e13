#!/bin/bash
# round 3: 48-byte packed voxel records -- tests first, then kernel time and the bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03s16
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_features.py tests/test_gpu_multigrid.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/t.log 2>&1; rc=$?
tail -12 $OUT/t.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tests/gpu_abl_bench.py "f64 records" 2>&1 | grep -v amdgpu.ids | tee $OUT/packed.txt
NDT_ABL_PACKED=1 timeout -k 10 200 python tests/gpu_abl_bench.py "packed 48-byte records" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/packed.txt
timeout -k 10 200 python tests/gpu_abl_bench.py "f64 records (again)" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/packed.txt
timeout -k 10 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ('value','ms_per_step','ms_target_build','ms_align')}); print(d['packed_records'])"
