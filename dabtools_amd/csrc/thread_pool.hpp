// thread_pool.hpp — a small persistent pool for the host-side per-stream work (control plane, work lists).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "placement.hpp"

namespace dabhip {

class ThreadPool {
 public:
  // cpus: the CPUs the workers are bound to before they do (or allocate) anything; empty = wherever the scheduler puts them (placement.hpp)
  explicit ThreadPool(int nthreads, std::vector<int> cpus = {}) : cpus_(std::move(cpus))
  {
    for (int t = 0; t < nthreads; ++t) workers_.emplace_back([this] { (void)bind_this_thread(cpus_); loop(); });
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

  // fn(i) for every i in [0, n); the calling thread takes part; returns when all are done.  One parallel_for at a time: the pool has ONE job slot
  // (the engine calls it from its decode thread -- upload staging -- and from its host lane -- control plane, work lists -- which never overlap by
  // construction; a second caller that did arrive simply waits its turn here)
  void parallel_for(int n, const std::function<void(int)>& fn)
  {
    if (n <= 0) return;
    std::lock_guard<std::mutex> one_caller(call_mu_);
    Job job;
    {
      std::lock_guard<std::mutex> lk(mu_);
      job = Job{&fn, n, ++epoch_};
      job_ = job;
      pending_ = n;
      ticket_.store(static_cast<uint64_t>(job.epoch) << 32);
    }
    cv_.notify_all();
    run(job);
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
  }

 private:
  struct Job {
    const std::function<void(int)>* fn = nullptr;
    int n = 0;
    uint32_t epoch = 0;
  };
  // Tickets carry the epoch they belong to: a worker that woke late for an earlier parallel_for finds the epoch of its
  // snapshot gone and takes nothing, so an item can neither run twice nor after its parallel_for has returned.
  void run(const Job& job)
  {
    int did = 0;
    uint64_t cur = ticket_.load();
    for (;;) {
      if (static_cast<uint32_t>(cur >> 32) != job.epoch) break;
      const int i = static_cast<int>(cur & 0xffffffffu);
      if (i >= job.n) break;
      if (!ticket_.compare_exchange_weak(cur, cur + 1)) continue;   // cur reloaded
      (*job.fn)(i);
      ++did;
      cur = ticket_.load();
    }
    if (did) {
      std::lock_guard<std::mutex> lk(mu_);
      pending_ -= did;            // same epoch: a ticket of epoch e is only handed out while parallel_for e waits
      if (pending_ == 0) done_.notify_all();
    }
  }
  void loop()
  {
    uint32_t seen = 0;
    for (;;) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        job = job_;               // function, count and epoch taken together under the lock
      }
      run(job);
    }
  }

  std::vector<int> cpus_;             // (before workers_: the threads read it)
  std::vector<std::thread> workers_;
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  Job job_;
  std::atomic<uint64_t> ticket_{0};   // epoch << 32 | next index
  int pending_ = 0;
  uint32_t epoch_ = 0;
  bool stop_ = false;
};

// One persistent host thread that runs posted closures in order: the engine's control-plane pass runs here beside the GPU
// work of the same decode (it used to be a std::thread created and joined per decode).
class AsyncLane {
 public:
  explicit AsyncLane(std::vector<int> cpus = {}) : cpus_(std::move(cpus)), worker_([this] { (void)bind_this_thread(cpus_); loop(); }) {}
  ~AsyncLane()
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    worker_.join();
  }
  AsyncLane(const AsyncLane&) = delete;
  AsyncLane& operator=(const AsyncLane&) = delete;

  // fn runs on the lane's thread; everything it references must stay alive until wait() has returned
  void post(std::function<void()> fn)
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      queue_.push_back(std::move(fn));
      ++posted_;
    }
    cv_.notify_all();
  }
  // returns when everything posted so far has run
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu_);
    const uint64_t target = posted_;
    idle_.wait(lk, [&] { return finished_ >= target; });
  }

 private:
  void loop()
  {
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;          // stop requested and nothing left to run
        fn = std::move(queue_.front());
        queue_.erase(queue_.begin());
      }
      fn();
      {
        std::lock_guard<std::mutex> lk(mu_);
        ++finished_;
      }
      idle_.notify_all();
    }
  }

  std::mutex mu_;
  std::condition_variable cv_, idle_;
  std::vector<std::function<void()>> queue_;
  uint64_t posted_ = 0, finished_ = 0;
  bool stop_ = false;
  std::vector<int> cpus_;
  std::thread worker_;                       // last member: starts after the state it uses exists
};

}  // namespace dabhip
