set -e
# in-kernel stamps of the build kernels: a SEPARATE library (own object directory), the production
# libndt_hip.so is not touched
make -C $GRAFT_REPO_ROOT/slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_HIP_LIB=$GRAFT_REPO_ROOT/slam-sam_amd/libndt_hip_stamps.so python $GRAFT_REPO_ROOT/tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tee $GRAFT_REPO_ROOT/gpurun_out/build_stamps.txt
