// pp_pool.h -- fork-join pool for the per-query host work of the planner's rounds (host code only).
//
// run(n, fn) calls fn(i) for every i in [0, n) on the workers and on the calling thread and returns
// when all of them are done; if fn threw for some item, run() throws std::runtime_error then (the
// other items still ran).  Several threads may call run() on one pool at the same time (the
// parts of a large batch, each driven by a thread of its own, share the workers): a worker takes
// items from whichever job has some left, so a part whose rounds are short, or that waits for the
// device, leaves its share of the workers to the others.
//
// A round is short and the next one follows within a few hundred microseconds, so a worker spins
// for a while before it goes to sleep on the condition variable (waking a sleeper costs more than a
// small round's work), and a caller never waits for a worker to show up: it works through its own
// job and waits only for the items other threads have taken.
#ifndef PP_POOL_H_
#define PP_POOL_H_

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#if defined(__x86_64__) || defined(__i386__)
#define PP_POOL_PAUSE() __builtin_ia32_pause()
#else
#define PP_POOL_PAUSE() std::this_thread::yield()
#endif

class Pool {
 public:
  explicit Pool(unsigned workers) {
    th_.reserve(workers);
    for (unsigned i = 0; i < workers; ++i) {
      try {
        th_.emplace_back([this] { loop(); });
      } catch (...) {
        break;  // no more threads to be had: the pool works with the ones it has
      }
    }
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> l(m_);
      stop_.store(true, std::memory_order_seq_cst);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  Pool(const Pool&) = delete;
  Pool& operator=(const Pool&) = delete;
  unsigned workers() const { return (unsigned)th_.size(); }

  // `chunk` = items a thread takes at a time: small for items that are work, large for trivial ones;
  // fewer than `serial_below` items run on the calling thread
  void run(size_t n, const std::function<void(size_t)>& fn, size_t chunk = 16, size_t serial_below = 64) {
    if (n == 0) return;
    Job* job = nullptr;
    if (!th_.empty() && n >= serial_below && n < kClosed) {
      for (Job& j : jobs_) {
        bool free_slot = false;
        if (j.busy.compare_exchange_strong(free_slot, true, std::memory_order_acquire)) {
          job = &j;
          break;
        }
      }
    }
    if (!job) {  // nothing to share the work with, or more callers at once than the pool has job slots
      bool threw = false;
      for (size_t i = 0; i < n; ++i) {
        try {
          fn(i);
        } catch (...) {
          threw = true;
        }
      }
      if (threw) throw std::runtime_error("a pool item threw");
      return;
    }
    job->fn.store(&fn, std::memory_order_relaxed);
    job->n.store((uint32_t)n, std::memory_order_relaxed);
    job->chunk.store((uint32_t)(chunk ? chunk : 1), std::memory_order_relaxed);
    job->done.store(0, std::memory_order_relaxed);
    job->threw.store(false, std::memory_order_relaxed);
    const uint64_t gen = (job->state.load(std::memory_order_relaxed) >> 32) + 1;
    job->state.store(gen << 32, std::memory_order_release);  // open: item 0 is next
    // (the epoch and the sleeper count are each written by one side and read by the other: both
    // sides use sequentially consistent accesses, so one of them sees the other's write)
    epoch_.fetch_add(1, std::memory_order_seq_cst);
    if (sleepers_.load(std::memory_order_seq_cst) != 0) {
      std::lock_guard<std::mutex> l(m_);
      cv_.notify_all();
    }
    work(*job);
    // items other threads took are still running
    for (unsigned spins = 0; job->done.load(std::memory_order_acquire) != (uint32_t)n; ++spins) {
      if (spins < 4096)
        PP_POOL_PAUSE();
      else
        std::this_thread::yield();
    }
    const bool threw = job->threw.load(std::memory_order_relaxed);  // (ordered by the acquire on done)
    job->state.store((gen << 32) | kClosed, std::memory_order_release);
    job->busy.store(false, std::memory_order_release);
    if (threw) throw std::runtime_error("a pool item threw");
  }

  // For a thread that is waiting for something else (a caller between its own jobs): runs one
  // chunk of whichever job has items left; false if there was nothing to do.
  bool help_once() {
    for (Job& j : jobs_)
      if (work(j, true)) return true;
    return false;
  }

 private:
  static constexpr double kSpinMs = 0.3;
  static constexpr uint64_t kClosed = 0xFFFFFFFFull;  // in the low half of a job's state: no job here
  static constexpr int kJobs = 8;
  // state = generation << 32 | next item.  The fields below it are written while the slot is
  // claimed and closed, before the new generation is published; a thread that reads them while
  // they change holds a state value that no longer matches, so its claim fails and it reads again.
  struct alignas(64) Job {
    std::atomic<uint64_t> state{kClosed};
    std::atomic<const std::function<void(size_t)>*> fn{nullptr};
    std::atomic<uint32_t> n{0}, chunk{1};
    std::atomic<uint32_t> done{0};
    std::atomic<bool> threw{false};  // some item's fn left by an exception (it counts as done)
    std::atomic<bool> busy{false};
  };

  // Takes items of `job` until none is left (or one chunk only); true if it ran any.
  bool work(Job& job, bool one_chunk = false) {
    bool any = false;
    uint64_t s = job.state.load(std::memory_order_acquire);
    for (;;) {
      const uint64_t next = s & kClosed;
      if (next == kClosed) return any;
      const uint32_t n = job.n.load(std::memory_order_relaxed), c = job.chunk.load(std::memory_order_relaxed);
      const std::function<void(size_t)>* fn = job.fn.load(std::memory_order_relaxed);
      if (next >= n) return any;
      const uint64_t end = next + c < n ? next + c : n;
      if (!job.state.compare_exchange_weak(s, (s & ~kClosed) | end, std::memory_order_acq_rel,
                                           std::memory_order_acquire))
        continue;  // s holds what is there now
      // the claim went through on the state the fields were read under: they belong to this job,
      // and the job cannot end before these items are counted
      for (size_t i = (size_t)next; i < (size_t)end; ++i) {
        try {  // a worker thread has nowhere to unwind to, and the caller waits for the count
          (*fn)(i);
        } catch (...) {
          job.threw.store(true, std::memory_order_relaxed);
        }
      }
      job.done.fetch_add((uint32_t)(end - next), std::memory_order_release);
      any = true;
      if (one_chunk) return true;
      s = job.state.load(std::memory_order_acquire);
    }
  }

  template <class Pred>
  static bool spin_until(Pred&& p) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int k = 0; k < 64; ++k) {
        if (p()) return true;
        PP_POOL_PAUSE();
      }
      if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > kSpinMs)
        return p();
    }
  }

  void loop() {
    for (;;) {
      const uint64_t seen = epoch_.load(std::memory_order_seq_cst);
      bool any = false;
      for (Job& j : jobs_) any |= work(j);
      if (stop_.load(std::memory_order_acquire)) return;
      if (any) continue;  // another job may have opened meanwhile
      auto news = [&] {
        return stop_.load(std::memory_order_acquire) || epoch_.load(std::memory_order_seq_cst) != seen;
      };
      if (!spin_until(news)) {
        std::unique_lock<std::mutex> l(m_);
        sleepers_.fetch_add(1, std::memory_order_seq_cst);
        cv_.wait(l, news);
        sleepers_.fetch_sub(1, std::memory_order_seq_cst);
      }
      if (stop_.load(std::memory_order_acquire)) return;
    }
  }

  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  Job jobs_[kJobs];
  std::atomic<uint64_t> epoch_{0};      // jobs opened so far
  std::atomic<uint32_t> sleepers_{0};   // workers inside cv_.wait
  std::atomic<bool> stop_{false};
};

#endif
