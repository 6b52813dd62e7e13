// Where do the ~7.7 us of launch + completion go?  Host time inside the launch call vs. the rest,
// for hipLaunchKernelGGL and for hipModuleLaunchKernel with a pre-packed argument buffer.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstring>
struct Big { float v[120]; };
extern "C" __global__ void k_flag(Big b, volatile unsigned long long* flag, unsigned long long seq, double* out) {
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 32) out[threadIdx.x] = (double)seq + b.v[1];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((unsigned long long*)flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStream_t sp; int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
  unsigned long long* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
  double* out; hipHostMalloc(&out, 256, hipHostMallocMapped);
  unsigned long long* dflag; hipHostGetDevicePointer((void**)&dflag, flag, 0);
  double* dout; hipHostGetDevicePointer((void**)&dout, out, 0);
  Big b{};
  hipFunction_t fn;
  hipGetFuncBySymbol(&fn, (const void*)k_flag);
  struct __attribute__((packed, aligned(8))) Args { Big b; volatile unsigned long long* flag; unsigned long long seq; double* out; } args;
  for (int mode = 0; mode < 3; ++mode) {
    hipStream_t st = mode == 2 ? sp : s;
    std::vector<double> tc, tt;
    unsigned long long seq = *flag;
    for (int i = 0; i < 400; ++i) {
      ++seq;
      double a = now();
      if (mode == 1) {
        args.b = b; args.flag = dflag; args.seq = seq; args.out = dout;
        size_t sz = sizeof(args);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        hipModuleLaunchKernel(fn, 391, 1, 1, 512, 1, 1, 0, st, nullptr, extra);
      } else {
        hipLaunchKernelGGL(k_flag, dim3(391), dim3(512), 0, st, b, dflag, seq, dout);
      }
      double c = now();
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      double e = now();
      if (i >= 50) { tc.push_back(c - a); tt.push_back(e - a); }
    }
    hipStreamSynchronize(st);
    printf("%-38s launch call %.2f us | launch -> flag visible %.2f us\n",
           mode == 0 ? "hipLaunchKernelGGL" : mode == 1 ? "hipModuleLaunchKernel(packed args)" : "hipLaunchKernelGGL, high-priority stream", med(tc), med(tt));
  }
  return 0;
}
