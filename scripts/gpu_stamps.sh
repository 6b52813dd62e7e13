set -e
# in-kernel stamps of k_derivatives: a SEPARATE library (own object directory), the production
# libndt_hip.so is not touched
make -C $GRAFT_REPO_ROOT/slam-sam_amd/csrc VARIANT=stamps -j8 > /dev/null 2>&1
NDT_HIP_LIB=$GRAFT_REPO_ROOT/slam-sam_amd/libndt_hip_stamps.so python $GRAFT_REPO_ROOT/tests/gpu_stamps.py
