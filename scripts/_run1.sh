cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
timeout -k 10 200 python tools/kernel_bench.py base 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/kb_base.txt; echo "kb rc=$?"
NDT_HIP_LIB=$GRAFT_REPO_ROOT/slam-sam_amd/libndt_hip_stamps.so timeout -k 10 300 python tools/stamps_per_wave.py > gpurun_out/r05/stamps_per_wave.txt 2>&1; echo "stamps rc=$?"
tail -5 gpurun_out/r05/kb_base.txt; tail -60 gpurun_out/r05/stamps_per_wave.txt
