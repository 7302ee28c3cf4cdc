// thread_pool.hpp — a small persistent pool for the host-side per-stream work (control plane, work lists).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
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

  std::vector<std::thread> workers_;
  std::mutex mu_;
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
  AsyncLane() : worker_([this] { loop(); }) {}
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
  std::thread worker_;                       // last member: starts after the state it uses exists
};

}  // namespace dabhip
