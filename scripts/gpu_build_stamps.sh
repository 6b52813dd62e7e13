set -e
cd $GRAFT_REPO_ROOT/slam-sam_amd/csrc
rm -f build/ndt_target.o build/ndt_api.o
make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -DNDT_STAMPS" > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tests/gpu_build_stamps.py 2>&1 | grep -v amdgpu.ids | tee $GRAFT_REPO_ROOT/gpurun_out/r02_build_stamps.txt
