// thread_pool.hpp — a small persistent pool for the host-side per-stream work (control plane, work lists).
#pragma once

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace dabhip {

class ThreadPool {
 public:
  explicit ThreadPool(int nthreads)
  {
    for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~ThreadPool()
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  ThreadPool(const ThreadPool&) = delete;
  ThreadPool& operator=(const ThreadPool&) = delete;

  // fn(i) for every i in [0, n); the calling thread takes part; returns when all are done
  void parallel_for(int n, const std::function<void(int)>& fn)
  {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      n_ = n;
      next_.store(0);
      pending_ = n;
      ++epoch_;
    }
    cv_.notify_all();
    run();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void run()
  {
    int did = 0;
    for (int i = next_.fetch_add(1); i < n_; i = next_.fetch_add(1)) { (*fn_)(i); ++did; }
    if (did) {
      std::lock_guard<std::mutex> lk(mu_);
      pending_ -= did;
      if (pending_ == 0) done_.notify_all();
    }
  }
  void loop()
  {
    unsigned seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
      }
      run();
    }
  }

  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int n_ = 0, pending_ = 0;
  unsigned epoch_ = 0;
  bool stop_ = false;
};

}  // namespace dabhip
