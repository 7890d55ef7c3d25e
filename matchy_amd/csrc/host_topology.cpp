// Host topology for multi-GPU scatter / gather: which CPUs sit on the NUMA node of a GPU.
//
// The path shards by line block: every GPU gets its batches from host memory over its own PCIe link, and on a two-socket
// 8-GPU node the H2D rate of a batch depends on whether the thread that faults the batch's pages in, pins them and queues the copy
// runs on the socket the GPU hangs off. The workers of the multi-device scanner (capi.cpp) and the ranks of bench.py therefore bind
// themselves — sched_setaffinity in-process, never an exec — to the CPUs of their GPU's node:
//   hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bus id>/numa_node -> /sys/devices/system/node/node<N>/cpulist.
// Everything below the HIP call takes the sysfs root as a parameter so that the mapping is testable without a GPU.
#include "host_topology.h"

#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace mxy {

static std::string read_first_line(const std::string& path) {
    std::ifstream f(path);
    std::string s;
    if (f) std::getline(f, s);
    while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
    return s;
}

// "0-3,8,10-11" -> {0,1,2,3,8,10,11}; anything malformed ends the list there
std::vector<int> parse_cpulist(const std::string& text) {
    std::vector<int> cpus;
    size_t i = 0;
    const size_t n = text.size();
    auto number = [&](long& v) {
        if (i >= n || !isdigit((unsigned char)text[i])) return false;
        v = 0;
        while (i < n && isdigit((unsigned char)text[i])) { v = v * 10 + (text[i] - '0'); if (v > 1 << 20) return false; ++i; }
        return true;
    };
    while (i < n) {
        while (i < n && (text[i] == ',' || isspace((unsigned char)text[i]))) ++i;
        long a, b;
        if (!number(a)) break;
        b = a;
        if (i < n && text[i] == '-') { ++i; if (!number(b) || b < a) break; }
        for (long c = a; c <= b; ++c) cpus.push_back((int)c);
    }
    std::sort(cpus.begin(), cpus.end());
    cpus.erase(std::unique(cpus.begin(), cpus.end()), cpus.end());
    return cpus;
}

// hipDeviceGetPCIBusId gives "0000:c1:00.0" (any case); sysfs names the directory in lower case
static std::string lower(std::string s) { for (char& c : s) c = (char)tolower((unsigned char)c); return s; }

int numa_node_of_pci(const std::string& sysfs_root, const std::string& pci_bus_id) {
    const std::string s = read_first_line(sysfs_root + "/bus/pci/devices/" + lower(pci_bus_id) + "/numa_node");
    if (s.empty()) return -1;
    char* end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    return (end == s.c_str() || v < 0) ? -1 : (int)v;   // -1: the platform does not say (single node)
}

std::vector<int> cpus_of_node(const std::string& sysfs_root, int node) {
    if (node < 0) return {};
    return parse_cpulist(read_first_line(sysfs_root + "/devices/system/node/node" + std::to_string(node) + "/cpulist"));
}

std::vector<int> cpus_near_pci(const std::string& sysfs_root, const std::string& pci_bus_id) {
    return cpus_of_node(sysfs_root, numa_node_of_pci(sysfs_root, pci_bus_id));
}

// The affinity of the PROCESS as it was when this library was loaded (a container's cpuset, a launcher's taskset). Bindings are taken
// relative to it, not to the calling thread's current mask: a thread created by one that had already bound itself to GPU 0's node
// inherits that node's CPUs, and an intersection with them is empty for every GPU of the other socket (round-4 advisor finding:
// half the scatter/gather workers of a two-socket node stayed on the wrong socket).
static cpu_set_t initial_affinity() {
    cpu_set_t s;
    CPU_ZERO(&s);
    if (sched_getaffinity(0, sizeof(s), &s) != 0) CPU_ZERO(&s);
    return s;
}
static const cpu_set_t g_initial = initial_affinity();   // runs at load time (dlopen / program start), before any binding of ours

// Restrict the calling thread to `cpus` ∩ the process's initial affinity (a container's cpuset stays in force); returns the number of
// CPUs the thread may run on afterwards, 0 when nothing was changed (empty list, empty intersection, or the call failed).
int bind_calling_thread(const std::vector<int>& cpus) {
    if (cpus.empty()) return 0;
    cpu_set_t want;
    CPU_ZERO(&want);
    int n = 0;
    for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &g_initial)) { CPU_SET(c, &want); ++n; }
    if (n == 0) return 0;
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return 0;
    return n;
}

// Back to the process's initial affinity (a thread that bound itself for one allocation and goes on to other work)
int unbind_calling_thread() {
    if (CPU_COUNT(&g_initial) == 0) return 0;
    if (sched_setaffinity(0, sizeof(g_initial), &g_initial) != 0) return 0;
    return CPU_COUNT(&g_initial);
}

}  // namespace mxy
