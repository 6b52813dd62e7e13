// Launch + completion round-trip latency on one GPU: stream sync vs. host spin on a pinned flag.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <atomic>
#include <vector>
#include <algorithm>
struct Big { float v[120]; };
__global__ void k_empty(Big b, int* dummy) { if (b.v[0] == 123.f) dummy[0] = 1; }
__global__ void k_flag(Big b, volatile unsigned long long* flag, unsigned long long seq, double* out) {
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 32) out[threadIdx.x] = (double)seq + b.v[1];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((unsigned long long*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  if (argc > 1) hipSetDeviceFlags(hipDeviceScheduleSpin);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  int* d; hipMalloc(&d, 4);
  unsigned long long* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
  double* out; hipHostMalloc(&out, 256, hipHostMallocMapped);
  unsigned long long* dflag; hipHostGetDevicePointer((void**)&dflag, flag, 0);
  double* dout; hipHostGetDevicePointer((void**)&dout, out, 0);
  Big b{}; 
  for (int blocks : {1, 782}) {
    std::vector<double> t1, t2;
    for (int i = 0; i < 300; ++i) {
      double a = now();
      hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s, b, d);
      hipStreamSynchronize(s);
      t1.push_back(now() - a);
    }
    unsigned long long seq = *flag;
    for (int i = 0; i < 300; ++i) {
      ++seq;
      double a = now();
      hipLaunchKernelGGL(k_flag, dim3(blocks), dim3(256), 0, s, b, dflag, seq, dout);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      t2.push_back(now() - a);
    }
    hipStreamSynchronize(s);
    std::sort(t1.begin(), t1.end()); std::sort(t2.begin(), t2.end());
    printf("blocks %4d: launch+streamSync median %.2f us (min %.2f) | launch+spin-on-flag median %.2f us (min %.2f)\n", blocks, t1[150], t1[0], t2[150], t2[0]);
  }
  return 0;
}
