// ndt_repack_pool.h -- persistent host worker threads of the upload path (internal; plain C++ so that
// tests/cpp can drive it under the sanitizers without a GPU).
#pragma once

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace ndt {

// A few persistent host threads for the AoS -> SoA repack of large uploads (spawning six std::threads
// per upload cost more than 0.1 ms of a 0.6 ms hand-over).  Idle workers sleep on a condition variable.
struct RepackPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, done_cv;
  std::function<void()> job;
  unsigned long gen = 0;
  int pending = 0;   // workers that have not finished the current generation's job yet
  bool stop = false;
  // A worker only ever runs the job of a generation that was published AFTER it started: it is born
  // with `seen` = the generation current at that moment (read under the mutex).  A worker born with
  // seen = 0 into a pool whose gen was already > 0 used to wake at once and run the PREVIOUS upload's
  // job -- a lambda over a dead stack frame (tests/cpp/test_repack_pool.cpp).
  void ensure(unsigned n) {
    std::lock_guard<std::mutex> lk(m);
    while (th.size() < n) {
      const unsigned long born = gen;
      th.emplace_back([this, born] {
        unsigned long seen = born;
        for (;;) {
          std::function<void()> f;
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            f = job;
          }
          if (f) f();
          {
            std::lock_guard<std::mutex> lk(m);
            if (--pending == 0) done_cv.notify_all();
          }
        }
      });
    }
  }
  // every worker runs f once (f claims chunks from a shared counter); returns at once.  Must be
  // followed by wait() before the next run() and before f's captures go out of scope.
  void run(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f);
    ++gen;
    pending = (int)th.size();  // every existing worker sees this generation exactly once
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return pending == 0; });
    job = nullptr;  // no callable outlives the frame it captured
  }
  ~RepackPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
      cv.notify_all();
    }
    for (auto& t : th) t.join();
  }
};

}  // namespace ndt
