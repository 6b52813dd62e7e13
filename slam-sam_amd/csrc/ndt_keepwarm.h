// ndt_keepwarm.h -- an optional low-rate heartbeat that keeps an otherwise idle device from dropping its clocks between
// two scans (ndt_set_keepwarm, include/ndt_hip.h; internal).  A driver at the reference's 10-20 Hz keyframe rate leaves the
// device idle for 50-100 ms between aligns; the first evaluations of the next align then run on a device that is ramping
// back up ("idle-device latency").  Default off: it costs power, and a device shared with other work
// does not need it.
#pragma once

#include <atomic>
#include <chrono>
#include <thread>

#include <hip/hip_runtime.h>

namespace ndt {

class KeepWarm {
 public:
  ~KeepWarm() { stop(); }
  // period_us <= 0: off.  One small kernel (one block of 256 threads per compute unit, a few microseconds of FMAs) every
  // period_us while the engine has been idle for at least one period.
  int start(int device, int compute_units, int period_us);
  void stop();
  // the engine is busy (called at the start and the end of every align / evaluation batch / build): no beat needed
  void touch() { last_activity_.store(now_ns(), std::memory_order_relaxed); }
  int period_us() const { return period_us_; }
  long long beats() const { return beats_.load(std::memory_order_relaxed); }

 private:
  static long long now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
  void run(int device, int compute_units);
  std::thread th_;
  std::atomic<bool> quit_{false};
  std::atomic<long long> last_activity_{0};
  std::atomic<long long> beats_{0};
  int period_us_ = 0;
};

}  // namespace ndt
