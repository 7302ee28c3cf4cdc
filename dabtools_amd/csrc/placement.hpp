// placement.hpp — where the host side of a slice runs.  Host only, no GPU call.
//
// dab2eti.c:237 has ONE demod thread for ONE device and leaves it wherever the scheduler puts it.  With one engine per GPU of a two-socket
// node the host side of a slice -- its decode thread, the control-plane pool, the staging copies into page-locked memory and that memory
// itself -- should sit on the socket the GPU hangs off: the IQ crosses the socket interconnect once otherwise, and eight slices' pools
// migrate over all cores.  So: each device's NUMA node from sysfs (/sys/bus/pci/devices/<bdf>/numa_node), the node's CPUs from
// /sys/devices/system/node/node<N>/cpulist, the CPUs of a node dealt to the slices on it in contiguous, disjoint chunks, and every host
// thread of a slice bound to its chunk (sched_setaffinity) BEFORE it allocates anything: page-locked buffers are then first touched -- and
// pinned -- on that node.  A device without a known node (-1: single-socket machines, containers without sysfs) gets no binding at all.
// DABHIP_NUMA=0 switches it off.  Unmeasured on hardware: the boxes of this pool have one GPU.  Round 5: every list is cut down to the CPUs the
// process is allowed on, and pool sizes come from usable_cpus() below, not from the machine's thread count.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace dabhip {

// "0-3,8,10-11" -> {0, 1, 2, 3, 8, 10, 11}; anything unparsable ends the list
inline std::vector<int> parse_cpulist(const std::string& s)
{
  std::vector<int> out;
  const char* p = s.c_str();
  while (*p) {
    char* end = nullptr;
    const long a = std::strtol(p, &end, 10);
    if (end == p || a < 0) break;
    long b = a;
    p = end;
    if (*p == '-') {
      b = std::strtol(p + 1, &end, 10);
      if (end == p + 1 || b < a) break;
      p = end;
    }
    for (long c = a; c <= b && out.size() < 4096; ++c) out.push_back(static_cast<int>(c));
    if (*p == ',') ++p;
    else break;
  }
  return out;
}

// The CPUs of every slice: slice i sits on node slice_node[i]; node_cpus[n] lists node n's CPUs.  The k-th of the m slices on a node takes the
// k-th of m contiguous chunks of that node's list (sizes differ by at most one; a chunk is never empty while the node has at least m CPUs).
// Slices on an unknown node (< 0 or beyond the table) or on a node without CPUs get an empty list = no binding.
inline std::vector<std::vector<int>> plan_placement(const std::vector<int>& slice_node, const std::vector<std::vector<int>>& node_cpus)
{
  const int n = static_cast<int>(slice_node.size());
  std::vector<std::vector<int>> out(static_cast<size_t>(n));
  for (int node = 0; node < static_cast<int>(node_cpus.size()); ++node) {
    std::vector<int> members;
    for (int i = 0; i < n; ++i)
      if (slice_node[i] == node) members.push_back(i);
    const std::vector<int>& cpus = node_cpus[static_cast<size_t>(node)];
    const int m = static_cast<int>(members.size()), c = static_cast<int>(cpus.size());
    if (m == 0 || c == 0) continue;
    for (int k = 0; k < m; ++k) {
      // chunk k = [k c / m, (k + 1) c / m); with fewer CPUs than slices several slices share one CPU rather than going unbound
      int a = static_cast<int>(static_cast<long long>(k) * c / m), b = static_cast<int>(static_cast<long long>(k + 1) * c / m);
      if (b <= a) b = a + 1;
      out[static_cast<size_t>(members[static_cast<size_t>(k)])].assign(cpus.begin() + a, cpus.begin() + b);
    }
  }
  return out;
}

inline bool numa_enabled()
{
  const char* e = std::getenv("DABHIP_NUMA");
  return !(e && std::atoi(e) == 0);
}

inline std::string read_small_file(const std::string& path)
{
  std::string s;
  if (FILE* f = std::fopen(path.c_str(), "r")) {
    char buf[4096];
    const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
    buf[n] = 0;
    s = buf;
    std::fclose(f);
  }
  while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
  return s;
}

// NUMA node of a PCI device ("0000:c1:00.0"), -1 when sysfs does not say
inline int numa_node_of_pci(const std::string& bdf)
{
  std::string lower = bdf;
  for (char& ch : lower)
    if (ch >= 'A' && ch <= 'F') ch = static_cast<char>(ch - 'A' + 'a');
  const std::string s = read_small_file("/sys/bus/pci/devices/" + lower + "/numa_node");
  if (s.empty()) return -1;
  return std::atoi(s.c_str());
}

// (the machine's lists as sysfs gives them; allowed_node_cpus() below is what placement works with)
inline std::vector<std::vector<int>> system_node_cpus()
{
  std::vector<std::vector<int>> nodes;
  for (int n = 0; n < 64; ++n) {
    const std::string s = read_small_file("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist");
    if (s.empty()) {
      if (n > 0) break;
      nodes.emplace_back();
      continue;
    }
    nodes.push_back(parse_cpulist(s));
  }
  return nodes;
}

// ---- how many CPUs this process may really use (round 5) ----------------------------------------------------------------
// std::thread::hardware_concurrency() counts the machine's hardware threads -- 256 on the GPU boxes of this pool, whose containers are granted
// 16 CPUs' worth of time (CFS quota) -- and a launcher's taskset / numactl shrinks the affinity mask further.  Host pools sized from the machine
// oversubscribe the grant: eight ranks x 24 pool threads burn the quota of a 100 ms period early and the cgroup freezes EVERY thread until the
// next one.  So pools are sized from min(|affinity mask|, quota), and CPU lists are intersected with the mask before anything is bound to them.

// The CPUs this PROCESS may run on (what a launcher's taskset / numactl / cpuset left), ascending: /proc/self/status "Cpus_allowed_list" -- the mask of the
// thread-group leader, which is what a launcher set -- and not sched_getaffinity(0), which answers for the CALLING thread (a pool thread that has been
// bound already would report its own chunk) and fails with EINVAL on machines with more than 1024 CPU ids when handed a fixed cpu_set_t.  Falls back to
// the calling thread's mask read with a set sized for the machine; an unreadable mask is NO restriction (every CPU the machine reports), never "one CPU".
inline std::vector<int> allowed_cpus()
{
  {
    const std::string st = read_small_file("/proc/self/status");
    const size_t at = st.find("Cpus_allowed_list:");
    if (at != std::string::npos) {
      const size_t from = at + 18, eol = st.find('\n', from);
      const std::vector<int> list = parse_cpulist(st.substr(from, eol == std::string::npos ? std::string::npos : eol - from));
      if (!list.empty()) return list;
    }
  }
  std::vector<int> out;
  for (size_t ncpu = 1024; ncpu <= (size_t(1) << 20); ncpu *= 4) {       // grow the set until the kernel's mask fits (EINVAL: too small)
    cpu_set_t* set = CPU_ALLOC(ncpu);
    if (!set) break;
    const size_t bytes = CPU_ALLOC_SIZE(ncpu);
    CPU_ZERO_S(bytes, set);
    const int rc = sched_getaffinity(0, bytes, set);
    if (rc == 0)
      for (size_t c = 0; c < ncpu; ++c)
        if (CPU_ISSET_S(c, bytes, set)) out.push_back(static_cast<int>(c));
    CPU_FREE(set);
    if (rc == 0 || errno != EINVAL) break;
  }
  if (out.empty()) {                                                      // unreadable: no restriction known
    const unsigned hw = std::thread::hardware_concurrency();
    for (unsigned c = 0; c < (hw ? hw : 1u); ++c) out.push_back(static_cast<int>(c));
  }
  return out;
}

// "max 100000" / "1600000 100000" (cgroup v2 cpu.max) -> CPUs' worth of time, rounded up; 0 = no limit or not readable
inline int parse_cpu_max(const std::string& text)
{
  if (text.empty() || text.compare(0, 3, "max") == 0) return 0;
  char* end = nullptr;
  const long long quota = std::strtoll(text.c_str(), &end, 10);
  const long long period = end ? std::strtoll(end, nullptr, 10) : 0;
  if (quota <= 0 || period <= 0) return 0;
  return static_cast<int>((quota + period - 1) / period);
}

// the tightest CFS quota on the way from this process's cgroup to the root, in CPUs (0 = none): cgroup v2 cpu.max, else v1 cfs_quota_us / cfs_period_us
inline int cfs_quota_cpus()
{
  int best = 0;
  auto take = [&](int q) { if (q > 0 && (best == 0 || q < best)) best = q; };
  std::string rel;                                         // "0::/some/path" in /proc/self/cgroup (v2)
  {
    const std::string cg = read_small_file("/proc/self/cgroup");
    size_t pos = 0;
    while (pos < cg.size()) {
      const size_t eol = cg.find('\n', pos);
      const std::string line = cg.substr(pos, eol == std::string::npos ? std::string::npos : eol - pos);
      if (line.compare(0, 3, "0::") == 0) rel = line.substr(3);
      if (eol == std::string::npos) break;
      pos = eol + 1;
    }
  }
  while (!rel.empty() && rel.back() == '/') rel.pop_back();
  std::string path = "/sys/fs/cgroup" + rel;
  for (int depth = 0; depth < 32; ++depth) {               // the process's own group, then every ancestor (a container usually sees its own as the root)
    take(parse_cpu_max(read_small_file(path + "/cpu.max")));
    if (path.size() <= std::string("/sys/fs/cgroup").size()) break;
    const size_t slash = path.find_last_of('/');
    if (slash == std::string::npos) break;
    path.resize(slash);
  }
  if (best == 0) {                                         // cgroup v1
    const long long q = std::atoll(read_small_file("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").c_str());
    const long long per = std::atoll(read_small_file("/sys/fs/cgroup/cpu/cpu.cfs_period_us").c_str());
    if (q > 0 && per > 0) take(static_cast<int>((q + per - 1) / per));
  }
  return best;
}

// what host pools are sized from: min(CPUs in the affinity mask, CFS quota), at least 1.  DABHIP_CPUS=n overrides it (tests, odd containers).
inline int usable_cpus()
{
  if (const char* e = std::getenv("DABHIP_CPUS")) {
    const int v = std::atoi(e);
    if (v > 0) return v;
  }
  int n = static_cast<int>(allowed_cpus().size());
  if (n <= 0) n = 1;
  const int q = cfs_quota_cpus();
  return q > 0 && q < n ? q : n;
}

// cpus without those outside `allowed` (order kept)
inline std::vector<int> intersect_cpus(const std::vector<int>& cpus, const std::vector<int>& allowed)
{
  std::vector<int> out;
  for (int c : cpus)
    for (int a : allowed)
      if (a == c) { out.push_back(c); break; }
  return out;
}

// every node's CPUs this process is allowed on (a launcher that pinned the process to one socket leaves the other node's list empty)
inline std::vector<std::vector<int>> allowed_node_cpus()
{
  std::vector<std::vector<int>> nodes = system_node_cpus();
  const std::vector<int> allowed = allowed_cpus();
  if (!allowed.empty())
    for (auto& n : nodes) n = intersect_cpus(n, allowed);
  return nodes;
}

// bind the CALLING thread to the part of `cpus` the process is allowed on; an empty list -- or an empty intersection: the launcher put this process
// somewhere else, and that stands -- leaves it alone (returns true: nothing was asked of the kernel).  Returns false when the kernel refuses.
inline bool bind_this_thread(const std::vector<int>& cpus)
{
  if (cpus.empty()) return true;
  const std::vector<int> use = intersect_cpus(cpus, allowed_cpus());
  if (use.empty()) return true;
  const size_t ncpu = static_cast<size_t>(*std::max_element(use.begin(), use.end())) + 1;
  cpu_set_t* set = CPU_ALLOC(ncpu);
  if (!set) return false;
  const size_t bytes = CPU_ALLOC_SIZE(ncpu);
  CPU_ZERO_S(bytes, set);
  for (int c : use)
    if (c >= 0) CPU_SET_S(static_cast<size_t>(c), bytes, set);
  const bool ok = pthread_setaffinity_np(pthread_self(), bytes, set) == 0;
  CPU_FREE(set);
  return ok;
}

}  // namespace dabhip
