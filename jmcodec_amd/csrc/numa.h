// jmcodec_amd/csrc/numa.h -- where a GPU sits in the host: the NUMA node of its PCI function, the CPUs of that node, and the two things the host
// pipeline does with them (run its threads there, take page-locked memory from there).  No reference counterpart: the reference drives one device
// from one thread (nv_dec.cpp:202-273); on an 8 x MI355X node the job lists of a GPU should not cross the socket interconnect on their way to it.
#pragma once
#include <vector>

namespace jmamd {

// NUMA node of HIP device `dev`: /sys/bus/pci/devices/<domain:bus:device.function>/numa_node; -1 when it cannot be told (one-node hosts report -1).
// JM_AMD_DEC_FAKE_NUMA="0:0,1:1" (device:node pairs) overrides the lookup -- tests of the many-GPUs-in-one-process mode on a host without GPUs.
// `query_hip` = false: only the override is consulted (parse-only handles never touch the HIP runtime).
int numa_node_of_device(int dev, bool query_hip);
// the CPUs of the node that this process may run on (cpulist of the node, cut down to the affinity mask); empty when unknown
std::vector<int> numa_cpus_of_node(int node);
// bind the calling thread to the node's CPUs; false (and no change) when the node or its CPUs are unknown
bool numa_bind_this_thread(int node);
// while alive, page allocations of the calling thread prefer `node` (set_mempolicy MPOL_PREFERRED): hipHostMalloc pins pages where they are first placed
// the policy the thread had before (whatever the application or numactl set) is restored when the object dies
struct NumaPreferred { explicit NumaPreferred(int node); ~NumaPreferred(); bool on = false; int old_mode = 0; unsigned long old_mask[16]; };

// ---- who else is on the GPU (KFD sysfs) ----
// The KFD driver's id of HIP device `dev` (/sys/class/kfd/kfd/topology/nodes/*/gpu_id, matched by PCI location); 0 when it cannot be told.
unsigned kfd_gpu_id_of_device(int dev);
// true when two or more processes (this one included) have compute queues on that GPU (/sys/class/kfd/kfd/proc/<pid>/queues/*/gpuid): the engine then forms no
// chain launches -- their no-deadlock argument needs the whole GPU (chain.hip); two processes on one device made them time out and be decoded again
bool kfd_gpu_has_other_users(unsigned gpu_id);

}  // namespace jmamd
