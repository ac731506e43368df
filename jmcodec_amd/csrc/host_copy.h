// jmcodec_amd/csrc/host_copy.h -- the "direct" output route: one copy-engine transfer from a frame's device staging into the CALLER's buffer.
//
// jm_nvdec_output_frame hands the frame over in a host buffer the caller owns (nv_dec.cpp:750-828 copies device -> host there, then repacks on
// the CPU).  Here the repack already happened on the device (k_packout), so what is left is 3.1 MB per 1080p frame over PCIe -- and that link is
// the hard ceiling of the host-output rate: 54.0 GB/s measured = 17.4 k frames/s (tools/sdma_probe.cpp: three SDMA engines in turn; two 52.5, one alone 47
// GB/s).
// The HIP runtime sends every device-to-host copy of a process to one engine and its waits either spin or need a poll loop; this goes to the ROCr
// layer underneath instead: the caller's buffer is page-locked for the call (hsa_amd_memory_lock, 1.4 us), the copy is put on one of the two PCIe-capable
// engines
// in turn (hsa_amd_memory_async_copy_on_engine) and the calling thread sleeps on the completion signal (interrupt wait, no CPU) -- no staging copy,
// no CPU memcpy (the pinned route costs 0.3-0.57 ms of CPU per frame for it), no helper thread.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <atomic>

namespace jmamd {

class HostCopier {
public:
    static HostCopier *get(int hip_device);             // nullptr when the ROCr layer cannot be set up for this device (the route is then not used)
    // page-lock [p, p + n) for the device; returns the address the copy engines use for it, nullptr on failure
    void *lock(void *p, size_t n);
    void  unlock(void *p);
    // device -> locked host memory, blocking (asleep) until the bytes are there; sig = handle from new_signal().
    //   kDone          the bytes are in the buffer
    //   kNotSubmitted  nothing was queued (no signal, the runtime refused): the caller may take another route into the same buffer
    //   kFailed        the transfer ended with an error (signal below 0): it is OVER, the buffer may be unlocked / written by another route
    //   kStuck         the transfer was queued and never completed: it may still write into the buffer -- the caller must neither unlock nor reuse it
    enum Result { kDone = 0, kNotSubmitted = 1, kFailed = 2, kStuck = 3 };
    Result copy(void *locked_dst, const void *dev_src, size_t n, uint64_t sig);
    uint64_t new_signal();
    uint32_t engine_mask() const { uint32_t m = 0; for (int i = 0; i < n_engines_; i++) m |= engines_[i]; return m; }
    void  free_signal(uint64_t sig);
private:
    HostCopier() {}
    uint64_t gpu_ = 0, cpu_ = 0;                        // hsa_agent_t handles
    std::atomic<long> typical_wait_ns_{0};              // how long copies have been taking lately, queueing included (sizes the waiter's first sleep)
    uint32_t engines_[4] = {0, 0, 0, 0}; int n_engines_ = 0;   // engine ids used in turn (none: the runtime chooses)
};

}  // namespace jmamd
