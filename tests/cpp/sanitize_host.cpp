// Host-side code of the engine under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: GPU
// sanitizers are not available on the pool).  Compiles ndt_newton.cpp / ndt_comm.cpp directly
// (no HIP kernels involved) and drives them through their internal interfaces:
//   * Newton + More-Thuente on a synthetic concave score with an analytic evaluator, including
//     evaluations that fail and evaluations that return non-finite values;
//   * the 6x6 solve (Cholesky fast path and eigen route), result_covariance, pose <-> matrix;
//   * the shared-memory reducer with 4 threads x 2000 rounds (bit-identical sums on every rank);
//   * SE(3) exp / log round trips;
//   * the upload path's worker pool growing between two jobs whose captures live on dead frames;
//   * the host half of the asynchronous cloud hand-off (stage_cloud): every layout and stride against a scalar
//     restatement, the caller's cloud in an exactly-sized heap block (no byte beyond the last point's z is read)
//     that is overwritten and freed THE MOMENT the call returns -- the staging copy must already be complete.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <unistd.h>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../slam-sam_amd/csrc/ndt_comm.h"
#include "../../slam-sam_amd/csrc/ndt_newton.h"
#include "../../slam-sam_amd/csrc/ndt_repack_pool.h"
#include "../../slam-sam_amd/csrc/ndt_se3.h"

static int fails = 0;
#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } \
  } while (0)

int main() {
  std::mt19937 rng(7);
  std::normal_distribution<double> N(0.0, 1.0);
  // ---- Newton driver on score(p) = -(p - p*)^T A (p - p*) / 2 + c --------------------------------
  double A[36], M[36];
  for (auto& v : M) v = N(rng);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double s = i == j ? 2.0 : 0.0;
      for (int k = 0; k < 6; ++k) s += M[6 * i + k] * M[6 * j + k];
      A[6 * i + j] = s * 50.0;
    }
  const double popt[6] = {0.4, -0.1, 0.25, 0.02, -0.03, 0.1};
  int calls = 0;
  ndt::EvalFn fn = [&](const double* p, const float*, bool need_h, ndt::Eval* e) -> int {
    ++calls;
    double d[6];
    for (int i = 0; i < 6; ++i) d[i] = p[i] - popt[i];
    e->score = 1000.0;
    for (int i = 0; i < 6; ++i) {
      double g = 0;
      for (int j = 0; j < 6; ++j) g += A[6 * i + j] * d[j];
      e->g[i] = -g;
      e->score -= 0.5 * d[i] * g;
    }
    for (int i = 0; i < 36; ++i) e->H[i] = need_h ? -A[i] : 0.0;
    e->nvtl_sum = 10; e->n_with = 10; e->n_pairs = 40;
    return 0;
  };
  ndt_params prm;   // ndt_default_params() lives in the HIP translation unit: same values by hand
  std::memset(&prm, 0, sizeof(prm));
  prm.resolution = 1.0f; prm.outlier_ratio = 0.55; prm.search_method = NDT_DIRECT7; prm.min_points_per_voxel = 6;
  prm.eig_inflation_ratio = 0.01; prm.hessian_mode = NDT_HESSIAN_FULL; prm.use_line_search = 1;
  prm.trans_epsilon = 1e-6; prm.step_size = 0.1; prm.max_iterations = 60;
  float guess[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  ndt_result r;
  for (int in_trials = 0; in_trials < 2; ++in_trials) {
    std::memset(&r, 0, sizeof(r));
    CHECK(ndt::newton_align(prm, 100, guess, fn, &r, in_trials != 0) == 0);
    CHECK(r.converged == 1);
    double err = 0;
    for (int i = 0; i < 6; ++i) err = std::fmax(err, std::fabs(r.final_pose[i] - popt[i]));
    CHECK(err < 1e-3);
  }
  prm.use_line_search = 0;
  std::memset(&r, 0, sizeof(r));
  CHECK(ndt::newton_align(prm, 100, guess, fn, &r) == 0);
  prm.use_line_search = 1;
  // evaluator failure is propagated, non-finite evaluations do not crash the driver
  int n = 0;
  ndt::EvalFn failing = [&](const double* p, const float* T, bool h, ndt::Eval* e) -> int {
    return ++n == 3 ? -3 : fn(p, T, h, e);
  };
  CHECK(ndt::newton_align(prm, 100, guess, failing, &r) != 0);
  ndt::EvalFn nan_eval = [&](const double* p, const float* T, bool h, ndt::Eval* e) -> int {
    fn(p, T, h, e);
    e->score = std::nan(""); e->g[2] = INFINITY; e->H[7] = std::nan("");
    return 0;
  };
  std::memset(&r, 0, sizeof(r));
  (void)ndt::newton_align(prm, 100, guess, nan_eval, &r);
  CHECK(std::isfinite(r.final_transformation[12]));

  // ---- pose <-> matrix, covariance ---------------------------------------------------------------
  for (int t = 0; t < 200; ++t) {
    double p[6] = {N(rng) * 10, N(rng) * 10, N(rng), N(rng) * 0.5, N(rng) * 0.5, N(rng)}, q[6];
    float T[16];
    ndt::pose_to_matrix(p, T);
    ndt::matrix_to_pose(T, q);
    float T2[16];
    ndt::pose_to_matrix(q, T2);
    for (int i = 0; i < 16; ++i) CHECK(std::fabs(T[i] - T2[i]) < 2e-5f);
    float jang[24], hang[45];
    ndt::angle_tables(p, jang, hang);
  }
  double H[36], cov[36];
  for (int i = 0; i < 36; ++i) H[i] = -A[i];
  CHECK(ndt::result_covariance(H, 1e-6, true, cov));
  CHECK(cov[0] > 0);
  std::memset(H, 0, sizeof(H));
  CHECK(!ndt::result_covariance(H, 0.0, false, cov));
  H[0] = std::nan("");
  CHECK(!ndt::result_covariance(H, 1e-6, false, cov));
  double d1, d2;
  ndt::gauss_constants(1.0, 0.55, &d1, &d2);
  CHECK(std::isfinite(d1) && std::isfinite(d2));

  // ---- shared-memory reducer: 4 ranks as threads ---------------------------------------------------
  {
    const int R = 4, rounds = 2000;
    const std::string name = "/ndt_sanitize_" + std::to_string((long)getpid());
    std::vector<std::vector<double>> last(R, std::vector<double>(NDT_EVAL_WORDS));
    std::vector<int> rc(R, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < R; ++k)
      th.emplace_back([&, k] {
        ndt::Reducer red;
        std::string err;
        if (red.init_shm(name.c_str(), k, R, &err) != NDT_OK) { rc[k] = 1; return; }
        double w[NDT_EVAL_WORDS];
        for (int it = 0; it < rounds; ++it) {
          for (int i = 0; i < NDT_EVAL_WORDS; ++i) w[i] = (k + 1) * 0.1 * (i + 1) + it * 1e-3;
          if (red.allreduce_host(w, NDT_EVAL_WORDS, &err) != NDT_OK) { rc[k] = 2; return; }
        }
        std::memcpy(last[k].data(), w, sizeof(w));
      });
    for (auto& t : th) t.join();
    for (int k = 0; k < R; ++k) CHECK(rc[k] == 0);
    for (int k = 1; k < R; ++k) CHECK(std::memcmp(last[0].data(), last[k].data(), sizeof(double) * NDT_EVAL_WORDS) == 0);
    double want = 0;
    for (int k = 0; k < R; ++k) want += (k + 1) * 0.1 * 1 + (rounds - 1) * 1e-3;
    CHECK(std::fabs(last[0][0] - want) < 1e-9);
  }

  // ---- a stale segment of a crashed run under the same name (magic set, large sequence words) ------
  // rank 1 starts FIRST and finds it; it must not be fooled into summing its garbage: only a live
  // rank 0 answers the attach handshake
  {
    const std::string name = "/ndt_sanitize_stale_" + std::to_string((long)getpid());
    {
      ndt::Reducer r0, r1;
      std::string err;
      std::thread t([&] { (void)r1.init_shm(name.c_str(), 1, 2, &err); });
      std::string err0;
      CHECK(r0.init_shm(name.c_str(), 0, 2, &err0) == NDT_OK);
      t.join();
      double w[NDT_EVAL_WORDS] = {0};
      std::thread t2([&] { double v[NDT_EVAL_WORDS] = {0}; for (int i = 0; i < 1000; ++i) (void)r1.allreduce_host(v, NDT_EVAL_WORDS, &err); });
      for (int i = 0; i < 1000; ++i) (void)r0.allreduce_host(w, NDT_EVAL_WORDS, &err0);
      t2.join();
      // simulate the crash: keep the name alive by re-linking a copy of the used segment
      int fd = shm_open(name.c_str(), O_RDWR, 0600);
      CHECK(fd >= 0);
      std::vector<char> image(1 << 20);
      const ssize_t got = pread(fd, image.data(), image.size(), 0);
      close(fd);
      r1.destroy();
      r0.destroy();  // unlinks
      fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      CHECK(fd >= 0 && got > 0);
      CHECK(ftruncate(fd, (off_t)(1 << 20)) == 0 && pwrite(fd, image.data(), (size_t)got, 0) == got);
      close(fd);
    }
    ndt::Reducer a, b;
    std::string ea, eb;
    int rc_b = -1;
    std::thread late([&] { rc_b = b.init_shm(name.c_str(), 1, 2, &eb); });   // meets the stale segment first
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    CHECK(a.init_shm(name.c_str(), 0, 2, &ea) == NDT_OK);
    late.join();
    CHECK(rc_b == NDT_OK);
    double wa[NDT_EVAL_WORDS], wb[NDT_EVAL_WORDS];
    for (int i = 0; i < NDT_EVAL_WORDS; ++i) { wa[i] = 1.0 + i; wb[i] = 100.0; }
    std::thread tb([&] { (void)b.allreduce_host(wb, NDT_EVAL_WORDS, &eb); });
    CHECK(a.allreduce_host(wa, NDT_EVAL_WORDS, &ea) == NDT_OK);
    tb.join();
    for (int i = 0; i < NDT_EVAL_WORDS; ++i) CHECK(wa[i] == 101.0 + i && wb[i] == wa[i]);
  }

  // ---- upload worker pool: growth between two jobs ------------------------------
  // Every job captures its caller's stack frame by reference, as upload_soa's does.  A worker added
  // by ensure() between two run() calls must never run the PREVIOUS job: that frame is gone.  The
  // frames below are heap blocks freed right after wait(), so ASan sees a stale job as use-after-free;
  // the counters catch it without ASan too.
  {
    for (int trial = 0; trial < 300; ++trial) {
      ndt::RepackPool pool;
      auto upload = [&](unsigned workers, int chunks) {
        struct Frame { std::atomic<int> next{0}, done{0}; int chunks = 0; };
        Frame* f = new Frame();
        f->chunks = chunks;
        pool.ensure(workers);
        pool.run([f] {
          for (;;) {
            const int c = f->next.fetch_add(1);
            if (c >= f->chunks) return;
            f->done.fetch_add(1);
          }
        });
        pool.wait();
        const bool ok = f->done.load() == chunks;
        delete f;  // the frame is dead from here on
        return ok;
      };
      CHECK(upload(3, 8));
      CHECK(upload(6, 31));   // grows 3 -> 6 with gen already at 1: the new workers must wait for THIS job
      CHECK(upload(2, 5));    // more workers than asked for: the extra ones find no chunk
      CHECK(upload(9, 64));
      CHECK(pool.pending == 0 && !pool.job);
    }
  }

  // ---- host half of the cloud hand-off ------------------------------------------------------------
  // ndt_set_target / ndt_set_source return once stage_cloud has returned; from then on the caller may free or
  // overwrite its cloud while the copies out of staging are still running.  Here the "copy engine" is a callback
  // that records the chunks; the checks run AFTER the caller's memory is gone.
  {
    ndt::RepackPool pool;
    std::uniform_real_distribution<float> U(-100.f, 100.f);
    const size_t sizes[] = {1, 2, 3, 4, 5, 7, 8, 63, 4095, 8192, 8193, 20000, 131072, 131073, 300001};
    const size_t strides[] = {12, 16, 20, 32, 48, 0 /* SoA */};
    for (size_t n : sizes)
      for (size_t stride : strides)
        for (unsigned workers : {0u, 3u}) {
          std::vector<float> want(3 * n);
          for (auto& v : want) v = U(rng);
          std::vector<float> stage(ndt::StageJob::stage_floats(n), -7.0f);
          std::vector<std::pair<size_t, size_t>> chunks;
          ndt::StageJob job;
          job.n = n;
          job.stage = stage.data();
          char* aos = nullptr;
          float *sx = nullptr, *sy = nullptr, *sz = nullptr;
          if (stride) {
            // exactly the bytes a strided cloud of n points owns: (n - 1) * stride + 12
            const size_t bytes = (n - 1) * stride + 12;
            aos = static_cast<char*>(std::malloc(bytes));
            std::memset(aos, 0x5a, bytes);
            for (size_t i = 0; i < n; ++i) std::memcpy(aos + i * stride, &want[3 * i], 12);
            job.aos = aos;
            job.stride = stride;
          } else {
            sx = static_cast<float*>(std::malloc(n * 4)); sy = static_cast<float*>(std::malloc(n * 4)); sz = static_cast<float*>(std::malloc(n * 4));
            for (size_t i = 0; i < n; ++i) { sx[i] = want[3 * i]; sy[i] = want[3 * i + 1]; sz[i] = want[3 * i + 2]; }
            job.x = sx; job.y = sy; job.z = sz;
          }
          ndt::stage_cloud(&pool, workers, job, [&](size_t c, size_t lo, size_t hi) {
            CHECK(c == chunks.size());
            chunks.emplace_back(lo, hi);
          });
          // the call has returned: the caller's memory is the caller's again
          if (aos) { std::memset(aos, 0xff, (n - 1) * stride + 12); std::free(aos); }
          if (sx) { std::memset(sx, 0xff, n * 4); std::free(sx); std::free(sy); std::free(sz); }
          size_t covered = 0;
          for (auto& ch : chunks) { CHECK(ch.first == covered); covered = ch.second; }
          CHECK(covered == n);
          bool same = true;
          for (size_t i = 0; i < n && same; ++i) {
            const size_t c0 = i / job.chunk * job.chunk, len = job.seg(i / job.chunk), off = i - c0;
            const float* b = stage.data() + 3 * c0;
            same = b[off] == want[3 * i] && b[len + off] == want[3 * i + 1] && b[2 * len + off] == want[3 * i + 2];
          }
          CHECK(same);
        }
    CHECK(ndt::host_cpu_budget() >= 1);
  }

  // ---- XCD-aware chunk assignment of the derivative kernel (xcd_chunk, ndt_device.h) -----------------
  // For every row length, XCD offset, XCD count and stripe: a bijection of [0, L), and within a stripe the positions
  // that run on one XCD (position + off mod X) hold CONSECUTIVE chunks.
  {
    for (int X : {1, 2, 4, 8})
      for (int m : {0, 1, 3, 32, 64})
        for (int L : {1, 2, 7, 8, 9, 63, 64, 241, 256, 257, 391, 512, 1000, 3907})
          for (int off = 0; off < X; ++off) {
            std::vector<int> seen((size_t)L, 0);
            bool ok = true;
            const int S = m > 0 ? X * m : L;
            std::vector<int> last((size_t)X, -1);
            for (int p = 0; p < L && ok; ++p) {
              const int c = ndt::xcd_chunk(p, L, off, X, m);
              ok = c >= 0 && c < L && !seen[(size_t)c]++;
              if (!ok) break;
              ok = c / S == p / S;   // a position's chunk lies in its own stripe
              const int x = (p + off) % X;
              if (p % S < X) last[(size_t)x] = -1;   // first block of this XCD in the stripe
              if (last[(size_t)x] >= 0) ok = ok && c == last[(size_t)x] + 1;
              last[(size_t)x] = c;
            }
            CHECK(ok);
          }
    // single-pose grids with a dedicated summing block in front keep round 3's mapping: x*q + min(x,r) + j - 1
    for (int G : {17, 242, 257}) {
      bool same = true;
      for (int g = 1; g < G && same; ++g) {
        const int x = g & 7, j = g >> 3, q = G >> 3, r = G & 7;
        const int old_chunk = x * q + std::min(x, r) + j - 1;
        // the new form maps the point blocks [0, G - 1) among themselves; the old one mapped [0, G) and subtracted one:
        // both hand XCD x a run of consecutive chunks, in XCD order starting with the summing block's XCD
        if (G == 257) same = ndt::xcd_chunk(g - 1, G - 1, 1, 8, 0) == old_chunk;   // (identical where G - 1 is a multiple of 8)
        else same = ndt::xcd_chunk(g - 1, G - 1, 1, 8, 0) >= 0 && old_chunk >= 0;
      }
      CHECK(same);
    }
  }

  // ---- SE(3) ---------------------------------------------------------------------------------------
  for (int t = 0; t < 500; ++t) {
    double xi[6], back[6];
    const double scale = t % 5 == 0 ? 1e-9 : 1.0;  // small-angle branches too
    for (int i = 0; i < 3; ++i) xi[i] = N(rng) * 0.9 * scale;
    const double wn = std::sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    if (wn > 3.0)  // Log is the inverse of Exp only inside the ball |w| < pi
      for (int i = 0; i < 3; ++i) xi[i] *= 3.0 / wn;
    for (int i = 3; i < 6; ++i) xi[i] = N(rng) * 5.0;
    ndt::se3::Pose P = ndt::se3::expmap(xi);
    ndt::se3::logmap(P, back);
    for (int i = 0; i < 6; ++i) CHECK(std::fabs(xi[i] - back[i]) < 1e-7 * (1.0 + std::fabs(xi[i])));
    double T[16];
    ndt::se3::to_colmajor(ndt::se3::between(P, ndt::se3::retract(P, xi)), T);
    double rpy[3];
    ndt::se3::rpy(P, rpy);
  }
  std::printf(fails ? "FAIL (%d)\n" : "PASS\n", fails);
  return fails ? 1 : 0;
}
