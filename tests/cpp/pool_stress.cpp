// Stress of the planner's fork-join pool (path_planner_amd/csrc/pp_pool.h): several threads call
// run() on one pool at the same time, as the parts of a large batch do; every item of every job
// has to run exactly once, and run() may only return when its own job is complete.
// usage: pool_stress [callers] [workers] [seconds]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../path_planner_amd/csrc/pp_pool.h"

int main(int argc, char** argv) {
  const int callers = argc > 1 ? atoi(argv[1]) : 4;
  const int workers = argc > 2 ? atoi(argv[2]) : 6;
  const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
  std::atomic<long> bad{0}, jobs{0}, items{0};
  {
    Pool pool((unsigned)workers);
    auto caller = [&](int id) {
      std::mt19937 rng(1234 + id);
      std::vector<unsigned char> hits;
      const auto t0 = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        const size_t n = rng() % 8 == 0 ? rng() % 5000 : rng() % 40;
        const size_t chunk = 1 + rng() % 20, serial_below = rng() % 3 == 0 ? 2 : 64;
        hits.assign(n, 0);
        std::atomic<long> sum{0};
        pool.run(n, [&](size_t i) {
          if (++hits[i] != 1) ++bad;  // an item nobody else may touch
          if ((i * 2654435761u >> 7) % 64 == 0 && id == 0) std::this_thread::yield();
          sum += (long)i + 1;
        }, chunk, serial_below);
        // run() has returned: everything is in
        if (sum.load() != (long)(n * (n + 1) / 2)) ++bad;
        for (size_t i = 0; i < n; ++i)
          if (hits[i] != 1) ++bad;
        ++jobs;
        items += (long)n;
        if (rng() % 64 == 0 && n > 3) {  // an item that throws: every other item still runs, run() reports it
          std::atomic<long> ran{0};
          bool reported = false;
          try {
            pool.run(n, [&](size_t i) {
              if (i == n / 2) throw 1;
              ++ran;
            }, chunk, serial_below);
          } catch (const std::runtime_error&) {
            reported = true;
          }
          if (!reported || ran.load() != (long)n - 1) ++bad;
        }
        if (rng() % 4 == 0)  // a caller between its own jobs lends a hand, as a part's thread does while it waits for the device
          for (int k = 0; k < 8; ++k) pool.help_once();
        if (rng() % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(rng() % 700));  // lets workers fall asleep
      }
    };
    std::vector<std::thread> th;
    for (int c = 1; c < callers; ++c) th.emplace_back(caller, c);
    caller(0);
    for (auto& t : th) t.join();
  }  // the pool's destructor joins the workers
  printf("%d callers, %d workers: %ld jobs, %ld items, %ld errors\n", callers, workers, jobs.load(), items.load(), bad.load());
  return bad.load() ? 1 : 0;
}
