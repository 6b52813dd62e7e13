set -e
cd $GRAFT_REPO_ROOT/slam-sam_amd/csrc
for W in 3 4 5 6; do
  rm -f build/ndt_derivs.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -DNDT_DERIV_WAVES_PER_SIMD=$W" > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py "W=$W"
done
