// Host topology helpers (host_topology.cpp): CPUs on the NUMA node of a PCI device, thread binding.
#pragma once
#include <string>
#include <vector>

namespace mxy {
std::vector<int> parse_cpulist(const std::string& text);
int numa_node_of_pci(const std::string& sysfs_root, const std::string& pci_bus_id);   // -1: unknown / single node
std::vector<int> cpus_of_node(const std::string& sysfs_root, int node);
std::vector<int> cpus_near_pci(const std::string& sysfs_root, const std::string& pci_bus_id);
int bind_calling_thread(const std::vector<int>& cpus);   // cpus ∩ the process's affinity at load time; CPUs the thread may run on afterwards; 0 = unchanged
int unbind_calling_thread();                              // back to the process's affinity at load time
}  // namespace mxy
