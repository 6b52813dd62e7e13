// H2D probe (tuning aid): the copy engine (hipMemcpyAsync from pinned memory) against a kernel that PULLS the same
// bytes out of mapped pinned host memory over PCIe, as one piece and in chunks.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
__global__ void __launch_bounds__(256) k_pull(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main(int argc, char** argv) {
  const size_t N = 12u << 20;
  float *h = nullptr, *hd = nullptr, *d = nullptr;
  CK(hipHostMalloc((void**)&h, N, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  CK(hipMalloc((void**)&d, N));
  memset(h, 1, N);
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto best = [&](auto&& f) { double b = 1e9; for (int r = 0; r < 30; ++r) { auto t0 = std::chrono::steady_clock::now(); f(); hipStreamSynchronize(s);
      b = std::min(b, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } return b; };
  for (int nch : {1, 2, 4, 8, 16}) {
    const size_t sz = N / nch;
    double tc = best([&] { for (int c = 0; c < nch; ++c) hipMemcpyAsync((char*)d + c * sz, (char*)h + c * sz, sz, hipMemcpyHostToDevice, s); });
    printf("copy engine %2d x %5zu KB: %.3f ms (%.1f GB/s)\n", nch, sz >> 10, tc, N / tc / 1e6);
    for (int blocks : {64, 256, 1024}) {
      double tk = best([&] { for (int c = 0; c < nch; ++c) hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s, (const float4*)((char*)hd + c * sz), (float4*)((char*)d + c * sz), sz / 16); });
      printf("   pull kernel %2d x %5zu KB, %4d blocks: %.3f ms (%.1f GB/s)\n", nch, sz >> 10, blocks, tk, N / tk / 1e6);
    }
  }
  return 0;
}
