// jmcodec_amd/csrc/numa.cpp -- see numa.h.
#include "numa.h"
#include <hip/hip_runtime_api.h>
#include <dirent.h>
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
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) != hipSuccess || hipDeviceGetAttribute(&pdev, hipDeviceAttributePciDeviceId,
        dev) != hipSuccess ||
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

// set_mempolicy / get_mempolicy without libnuma: mode 1 = MPOL_PREFERRED.  The constructor runs on the APPLICATION's thread (gpu_alloc_sequence), so the
// policy that thread had -- a membind or interleave set by the application or by numactl -- is read first and put back by the destructor;
// when it cannot be read, nothing is changed at all.
NumaPreferred::NumaPreferred(int node) {
    if (node < 0 || node >= 1024 || numa_cpus_of_node(node).empty()) return;
    memset(old_mask, 0, sizeof old_mask);
    if (syscall(SYS_get_mempolicy, &old_mode, old_mask, (unsigned long)(sizeof old_mask * 8), nullptr, 0ul) != 0) return;
    unsigned long mask[16]; memset(mask, 0, sizeof mask);
    mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
    on = syscall(SYS_set_mempolicy, 1, mask, (unsigned long)(sizeof mask * 8)) == 0;
}
NumaPreferred::~NumaPreferred() {
    if (!on) return;
    bool any = false;
    for (unsigned long w : old_mask) any |= w != 0;
    if (syscall(SYS_set_mempolicy, old_mode, any ? old_mask : nullptr, any ? (unsigned long)(sizeof old_mask * 8) : 0ul) != 0)
        (void)syscall(SYS_set_mempolicy, 0, nullptr, 0ul);
}

// ------------------------------------------------------------------------------------------------------------------------------------------
static bool read_small(const char *path, char *buf, size_t n) {
    FILE *fp = fopen(path, "r");
    if (!fp) return false;
    const size_t got = fread(buf, 1, n - 1, fp);
    fclose(fp);
    buf[got] = 0;
    return got > 0;
}

unsigned kfd_gpu_id_of_device(int dev) {
    int bus = -1, pdev = -1, dom = -1;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) != hipSuccess || hipDeviceGetAttribute(&pdev, hipDeviceAttributePciDeviceId,
        dev) != hipSuccess ||
        hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const unsigned want_loc = ((unsigned)bus << 8) | ((unsigned)pdev << 3);
    unsigned only = 0; int n_readable = 0;
    for (int node = 0; node < 256; node++) {
        char path[128], buf[4096];
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", node);
        if (!read_small(path, buf, sizeof buf)) { if (node > 64) break; continue; }         // (a container sees the nodes of other tenants' GPUs as unreadable)
        const unsigned id = (unsigned)strtoul(buf, nullptr, 10);
        if (!id) continue;                                                               // CPU node
        only = id; n_readable++;
        snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", node);
        if (!read_small(path, buf, sizeof buf)) continue;
        unsigned loc = ~0u, domain = 0;
        if (const char *p = strstr(buf, "location_id ")) loc = (unsigned)strtoul(p + 12, nullptr, 10);
        if (const char *p = strstr(buf, "domain ")) domain = (unsigned)strtoul(p + 7, nullptr, 10);
        if (loc == want_loc && (int)domain == dom) return id;
    }
    return n_readable == 1 ? only : 0;                                                   // one GPU visible: it is the one
}

// The directory names are process ids of the HOST's pid namespace: inside a container getpid() names nothing there, so "a process other than this
// one" cannot be told by id.  Counted instead: processes with at least one compute queue (type != 1: SDMA queues belong to copies) on the GPU.  This
// process is one of them -- the engine creates its streams before the first check -- so two or more means company.
bool kfd_gpu_has_other_users(unsigned gpu_id) {
    if (!gpu_id) return false;
    DIR *d = opendir("/sys/class/kfd/kfd/proc");
    if (!d) return false;
    int users = 0;
    while (struct dirent *e = readdir(d)) {
        char *end; (void)strtol(e->d_name, &end, 10);
        if (end == e->d_name || *end) continue;
        char qdir[160]; snprintf(qdir, sizeof qdir, "/sys/class/kfd/kfd/proc/%s/queues", e->d_name);
        DIR *q = opendir(qdir);
        if (!q) continue;
        bool here = false;
        while (struct dirent *f = readdir(q)) {
            if (f->d_name[0] == '.') continue;
            char path[256], buf[32]; snprintf(path, sizeof path, "%s/%s/gpuid", qdir, f->d_name);
            if (!read_small(path, buf, sizeof buf) || (unsigned)strtoul(buf, nullptr, 10) != gpu_id) continue;
            snprintf(path, sizeof path, "%s/%s/type", qdir, f->d_name);
            if (read_small(path, buf, sizeof buf) && strtoul(buf, nullptr, 10) == 1) continue;       // an SDMA queue
            here = true; break;
        }
        closedir(q);
        users += here;
    }
    closedir(d);
    return users >= 2;
}

}  // namespace jmamd
