// Feasibility probe for a pre-launched evaluation kernel that waits for its pose in a mailbox:
// round trip host write -> spinning kernel sees it -> kernel writes a pinned flag -> host sees it,
// for a mailbox in pinned host memory and in fine-grained device memory written through the BAR,
// against the plain launch -> flag round trip.  Every spin loop has a 20 ms s_memrealtime limit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <x86intrin.h>

__global__ void k_wait(const unsigned long long* mbox, unsigned long long seq, unsigned int* arrive,
                       unsigned long long* host_flag, int all_blocks_poll) {
  __shared__ int ok;
  if (threadIdx.x == 0) {
    ok = 1;
    if (all_blocks_poll || blockIdx.x == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        const unsigned long long v = __hip_atomic_load(mbox, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == seq) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) { ok = 0; break; }  // 20 ms at 100 MHz
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int need = all_blocks_poll ? gridDim.x : 1u;
    const bool mine = all_blocks_poll ? true : blockIdx.x == 0;
    if (mine) {
      const unsigned int t = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == need - 1u) {
        __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(host_flag, ok ? seq : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__global__ void k_flag(unsigned long long seq, unsigned long long* host_flag) {
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    __hip_atomic_store(host_flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned long long* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
  unsigned long long* dflag; hipHostGetDevicePointer((void**)&dflag, flag, 0);
  unsigned long long* hbox; hipHostMalloc(&hbox, 64, hipHostMallocMapped); *hbox = 0;
  unsigned long long* dhbox; hipHostGetDevicePointer((void**)&dhbox, hbox, 0);
  unsigned int* arrive; hipMalloc(&arrive, 64); hipMemset(arrive, 0, 64);
  unsigned long long* fbox = nullptr;
  hipError_t fe = hipExtMallocWithFlags((void**)&fbox, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s ptr=%p\n", hipGetErrorString(fe), (void*)fbox);
  hipPointerAttribute_t at;
  if (fe == hipSuccess && hipPointerGetAttributes(&at, fbox) == hipSuccess)
    printf("  attributes: type=%d device=%d hostPointer=%p devicePointer=%p\n", (int)at.type, at.device, at.hostPointer, at.devicePointer);
  int largebar = -1; hipDeviceGetAttribute(&largebar, hipDeviceAttributeIsLargeBar, 0);
  printf("hipDeviceAttributeIsLargeBar=%d\n", largebar);
  unsigned long long seq = 0;
  const int blocks = 241, threads = 832;
  // (0) plain launch -> flag
  {
    std::vector<double> t;
    for (int i = 0; i < 300; ++i) {
      ++seq; const double a = now_us();
      hipLaunchKernelGGL(k_flag, dim3(blocks), dim3(threads), 0, s, seq, dflag);
      while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
      if (i >= 50) t.push_back(now_us() - a);
    }
    hipStreamSynchronize(s);
    printf("plain launch -> flag visible                      %.2f us\n", med(t));
  }
  auto run = [&](const char* name, unsigned long long* host_ptr, const unsigned long long* dev_ptr, int all) {
    std::vector<double> t; int fails = 0;
    for (int i = 0; i < 300; ++i) {
      ++seq;
      hipLaunchKernelGGL(k_wait, dim3(blocks), dim3(threads), 0, s, dev_ptr, seq, arrive, dflag, all);
      const double w = now_us(); while (now_us() - w < 25.0) {}   // the kernel is resident and spinning
      const double a = now_us();
      __atomic_store_n(host_ptr, seq, __ATOMIC_RELEASE);
      _mm_sfence();
      unsigned long long v;
      while ((v = __atomic_load_n(flag, __ATOMIC_ACQUIRE)) != seq && v != ~0ull) {}
      if (v == ~0ull) ++fails;
      if (i >= 50) t.push_back(now_us() - a);
    }
    hipStreamSynchronize(s);
    printf("%-50s %.2f us  (timeouts %d)\n", name, med(t), fails);
  };
  run("mailbox in pinned host memory, block 0 polls", hbox, dhbox, 0);
  run("mailbox in pinned host memory, all 241 blocks poll", hbox, dhbox, 1);
  if (fe == hipSuccess && largebar == 1) {
    run("mailbox in fine-grained DEVICE memory (BAR), block 0", fbox, fbox, 0);
    run("mailbox in fine-grained DEVICE memory (BAR), all blocks", fbox, fbox, 1);
  }
  return 0;
}
