set -e
cd $GRAFT_REPO_ROOT/slam-sam_amd/csrc
for V in "$@"; do
  rm -f build/ndt_derivs.o
  make CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result $V" > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tests/gpu_kernel_bench.py "[$V]"
done
