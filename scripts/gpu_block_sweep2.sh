for cfg in "512 2048" "448 2048" "256 2048" "256 512" "192 512" "128 512" "128 256" "64 256"; do
  set -- $cfg
  echo "== BLOCK=$1 SINGLE_LEVEL_MAX=$2"
  NDT_DERIV_BLOCK=$1 NDT_DERIV_SINGLE_LEVEL_MAX=$2 python tests/gpu_kernel_bench.py "B=$1,S=$2" 2>&1 | tail -1
done
