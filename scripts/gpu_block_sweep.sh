for B in 128 192 256 320 384 512 640 832; do
  echo "== NDT_DERIV_BLOCK=$B"
  NDT_DERIV_BLOCK=$B python tests/gpu_size_sweep.py 2>&1 | grep -E "n=  200000|n=  800000|n= 4000000|n=   50000"
  NDT_DERIV_BLOCK=$B python tests/gpu_kernel_bench.py "B=$B" 2>&1 | tail -1
done
