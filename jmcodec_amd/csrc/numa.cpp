// jmcodec_amd/csrc/numa.cpp -- see numa.h.
#include "numa.h"
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace jmamd {

int numa_node_of_device(int dev, bool query_hip) {
    if (const char *f = getenv("JM_AMD_DEC_FAKE_NUMA")) {
        for (const char *p = f; *p;) {
            char *e; long d = strtol(p, &e, 10);
            if (e == p || *e != ':') break;
            long n = strtol(e + 1, &e, 10);
            if (d == dev) return (int)n;
            p = *e ? e + 1 : e;
        }
        return -1;
    }
    if (!query_hip) return -1;
    int bus = -1, pdev = -1, dom = -1;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) != hipSuccess || hipDeviceGetAttribute(&pdev, hipDeviceAttributePciDeviceId, dev) != hipSuccess ||
        hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    char path[128];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node", dom, bus, pdev);
    int node = -1;
    if (FILE *fp = fopen(path, "r")) { if (fscanf(fp, "%d", &node) != 1) node = -1; fclose(fp); }
    return node;
}

std::vector<int> numa_cpus_of_node(int node) {
    std::vector<int> out;
    if (node < 0) return out;
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *fp = fopen(path, "r");
    if (!fp) return out;
    const bool ok = fgets(buf, sizeof buf, fp) != nullptr;
    fclose(fp);
    if (!ok) return out;
    cpu_set_t allowed; CPU_ZERO(&allowed);
    const bool have_mask = sched_getaffinity(0, sizeof allowed, &allowed) == 0;
    for (const char *p = buf; *p && *p != '\n';) {                  // "0-63,128-191"
        char *e; long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (!have_mask || CPU_ISSET((int)c, &allowed)) out.push_back((int)c);
        p = *e == ',' ? e + 1 : e;
    }
    return out;
}

bool numa_bind_this_thread(int node) {
    const std::vector<int> cpus = numa_cpus_of_node(node);
    if (cpus.empty()) return false;
    cpu_set_t set; CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    return sched_setaffinity(0, sizeof set, &set) == 0;
}

// set_mempolicy without libnuma: mode 1 = MPOL_PREFERRED, 0 = MPOL_DEFAULT
NumaPreferred::NumaPreferred(int node) {
    if (node < 0 || node >= 1024 || numa_cpus_of_node(node).empty()) return;
    unsigned long mask[16]; memset(mask, 0, sizeof mask);
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    on = syscall(SYS_set_mempolicy, 1, mask, (unsigned long)(sizeof mask * 8)) == 0;
}
NumaPreferred::~NumaPreferred() { if (on) (void)syscall(SYS_set_mempolicy, 0, nullptr, 0ul); }

}  // namespace jmamd
