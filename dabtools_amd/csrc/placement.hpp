// placement.hpp — where the host side of a slice runs.  Host only, no GPU call.
//
// dab2eti.c:237 has ONE demod thread for ONE device and leaves it wherever the scheduler puts it.  With one engine per GPU of a two-socket
// node the host side of a slice -- its decode thread, the control-plane pool, the staging copies into page-locked memory and that memory
// itself -- should sit on the socket the GPU hangs off: the IQ crosses the socket interconnect once otherwise, and eight slices' pools
// migrate over all cores.  So: each device's NUMA node from sysfs (/sys/bus/pci/devices/<bdf>/numa_node), the node's CPUs from
// /sys/devices/system/node/node<N>/cpulist, the CPUs of a node dealt to the slices on it in contiguous, disjoint chunks, and every host
// thread of a slice bound to its chunk (sched_setaffinity) BEFORE it allocates anything: page-locked buffers are then first touched -- and
// pinned -- on that node.  A device without a known node (-1: single-socket machines, containers without sysfs) gets no binding at all.
// DABHIP_NUMA=0 switches it off.  Unmeasured on hardware: the boxes of this pool have one GPU.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <string>
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

// bind the CALLING thread; an empty list leaves it alone.  Returns false when the kernel refuses (a cpuset that excludes the list, ...)
inline bool bind_this_thread(const std::vector<int>& cpus)
{
  if (cpus.empty()) return true;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus)
    if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
  return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
}

}  // namespace dabhip
