// parallel.hpp -- fn(i) for i in [0, n) on up to `threads` threads of a pool that stays with the process.
//
// The host stages of a batch -- runs to ops, erosion scans, splicing the patches in, swizzle and record text: four passes over a batch's
// records, each independent per record (the reference runs one Taskflow task per record, computeAlignments.hpp:391-435) -- used to start and join
// their own std::threads: 2.4 - 4 ms per pass for the 83 threads a batch of 660 records asks for, 8 - 10 ms of a C2 batch's 55, all of it with the
// device idle (scripts/prof_concurrency.py: 5.1 + 4.2 ms without a kernel between the main alignment, the patches and the records' tails).
// Here a pass is a job in a queue: the caller works on it, up to threads - 1 pool threads join it, and the call returns when every index is done.
// Several callers (the align driver's workers, each on a batch of its own) share the pool; a job's helpers are whoever is free.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace wfmash_host {

class WorkPool {
 public:
  // (a heap singleton that is never destroyed: its threads may be parked in wait() when the process's statics go)
  static WorkPool& get() {
    static WorkPool* p = new WorkPool;
    return *p;
  }

  template <typename F>
  void parallel_for(size_t n, int threads, F&& fn) {
    if (n == 0) return;
    if (threads <= 1 || n == 1) {
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    auto job = std::make_shared<Job>();
    job->n = n;
    job->helpers_wanted = (int)std::min<size_t>((size_t)threads, n) - 1;
    std::function<void(size_t)> call = [&fn](size_t i) { fn(i); };
    job->fn = &call;
    {
      std::lock_guard<std::mutex> lk(mu_);
      grow_locked(job->helpers_wanted);
      jobs_.push_back(job);
    }
    cv_.notify_all();
    run(*job);  // the caller's share
    {
      // the last index may still be running on a helper
      std::unique_lock<std::mutex> lk(job->mu);
      job->cv.wait(lk, [&] { return job->done.load(std::memory_order_acquire) == job->n; });
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (auto it = jobs_.begin(); it != jobs_.end(); ++it)
        if (it->get() == job.get()) { jobs_.erase(it); break; }
    }
    // (helpers that still hold the job only look at its counters: nothing of the caller's frame -- fn is not called once done == n)
    if (job->error) std::rethrow_exception(job->error);
  }

 private:
  struct Job {
    size_t n = 0;
    std::atomic<size_t> next{0}, done{0};
    std::atomic<int> helpers{0};
    int helpers_wanted = 0;
    const std::function<void(size_t)>* fn = nullptr;
    std::mutex mu;
    std::condition_variable cv;
    std::exception_ptr error;
  };

  static void run(Job& j) {
    size_t mine = 0;
    for (size_t i; (i = j.next.fetch_add(1, std::memory_order_relaxed)) < j.n;) {
      try {
        (*j.fn)(i);
      } catch (...) {
        std::lock_guard<std::mutex> lk(j.mu);
        if (!j.error) j.error = std::current_exception();
      }
      ++mine;
    }
    if (mine) {
      const size_t d = j.done.fetch_add(mine, std::memory_order_acq_rel) + mine;
      if (d == j.n) {
        std::lock_guard<std::mutex> lk(j.mu);
        j.cv.notify_all();
      }
    }
  }

  void grow_locked(int want) {
    static const int cap = (int)std::max(8u, 2 * std::thread::hardware_concurrency());
    want = std::min(want, cap);
    while ((int)threads_.size() < want) threads_.emplace_back([this] { loop(); });
  }

  void loop() {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] {
          for (auto& j : jobs_)
            if (j->next.load(std::memory_order_relaxed) < j->n && j->helpers.load(std::memory_order_relaxed) < j->helpers_wanted) { job = j; return true; }
          return false;
        });
        job->helpers.fetch_add(1, std::memory_order_relaxed);
      }
      run(*job);
    }
  }

  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::shared_ptr<Job>> jobs_;
  std::vector<std::thread> threads_;  // never joined: the pool lives as long as the process
};

template <typename F>
inline void parallel_for(size_t n, int threads, F&& fn) {
  WorkPool::get().parallel_for(n, threads, std::forward<F>(fn));
}

}  // namespace wfmash_host
