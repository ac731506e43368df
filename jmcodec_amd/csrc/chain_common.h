// jmcodec_amd/csrc/chain_common.h -- the device-memory protocol that couples consecutive pictures of one stream inside ONE launch (k_chain).
//
// Why: a picture's in-loop deblocking is a 254-step dependency chain at 1080p (x + 2y macroblock wavefront, clause 8.7 order) and the
// next picture's motion compensation reads its output, so one stream alone leaves the GPU idle and tops out near 1 / (chain latency).
// A chain launch holds up to kMaxChainDepth consecutive pictures of a stream; picture n+1 follows picture n at MACROBLOCK granularity:
//   * deblocking workgroups publish, per band of 16 macroblock rows and plane, how many wavefront steps are final in memory (`fin`);
//   * a reconstruction wave of picture n+1 waits until the steps that store the last sample of its reference windows are final;
//   * reconstruction waves publish each finished macroblock in a per-row bitmap (`bits`), on which picture n+1's own deblocking waits.
// Everything that crosses workgroups this way is written with write-through stores and read with loads that bypass the non-coherent
// cache levels (agent-scope relaxed atomics: the 8 XCDs of an MI355X have private L2s, and a 128-byte line fetched for a final sample
// also holds neighbours that are not final yet).  Acquire / release fences are deliberately not used: at agent scope they write back /
// invalidate a whole L2.
// THE INVARIANT (who may wait on whom, and why the waited-on workgroup is resident; round 4 wrote it down after two parked give-ups were traced to it):
//   1. There is ONE chain launch per device at a time (the ordinary lane's in-order stream), and its band workgroups -- deblocking bands, intra bands
//      -- come first in the grid and number at most half of what the device holds (Engine::form, chain_resident_workgroups): the dispatcher starts
//      workgroups in grid order per XCD and a waiting workgroup keeps its slot, so every band is resident before the first reconstruction group that
//      could wait for it, and stays for its whole wavefront.  (Two chain lanes at once -- tried in round 2 -- break exactly this: neither launch's
//      bands are guaranteed slots.)
//   2. A band waits only for (a) the band above it in the same picture (resident, 1.), (b) reconstruction bits of its own picture.
//   3. A reconstruction group waits only for `fin` of bands (resident) of EARLIER pictures of its stream.
//   4. The work list is sorted by a key such that everything a group depends on -- through 3., then 2.(b) of those bands and, through 2.(a), of the bands
//      above them -- has a SMALLER key (Engine::launch builds the keys; tools/chain_keys.py restates every wait of this file, deblock_device.h and
//      intra_device.h and checks the rule by brute force; tests/test_chain_keys.py).  Hence the unfinished group with the smallest key is resident --
//      everything before it in its XCD's order has finished or is a band -- and all it waits for is finished or resident: it runs.  No deadlock at any
//      occupancy, PROVIDED the keys are right: round 3's keys ignored that a band trails the band above it by ~7 steps (nine bands at 4K: the first
//      chain launch of a C2 run gave up in one run of three), and round 4's first one-row schedule kept one key slope per launch (launches of 4 / 8
//      streams gave up in three runs of ten).  Both were found as VIOLATIONS OF 4., not as mysteries.
// Every wait is still bounded (by the time its wave spent waiting: WaitClock below); a wait that gives up sets the error word and the abort word, records itself (record_first_giveup), and the
// engine decodes the launch's pictures again with the stage kernels (Engine::recover): a wrong key costs time, never a wrong frame.
// There is no reference counterpart (the reference hands whole pictures to the NVDEC ASIC, nv_dec.cpp:33-41).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "jobs.h"

namespace jmamd {

// per-picture block of ints in the batch's chain buffer
// (the H.264 stage kernels use the same block: one memset per batch clears every counter of every picture)
constexpr int kChainRing = 0;          // [64]  band step counters of the ring-row hand-over between deblocking bands (deblock_device.h)
constexpr int kChainFin = 64;          // [64]  steps final in memory: luma bands 0..31, chroma bands 32..63
constexpr int kChainIntraRing = 128;   // [64]  band step counters of the intra wavefront (intra_device.h)
constexpr int kChainIntraFin = 192;    // [64]  steps of the intra wavefront whose samples are in memory: luma bands 0..31, chroma bands 32..63
constexpr int kChainBits = 256;        // [mb_h][kChainRowWords] reconstruction bitmap, bit x of row y = macroblock (x, y) is in memory
constexpr int kChainRowWords = 8;      // pictures up to 256 macroblocks wide (4096 samples)
constexpr int kChainMaxRows = 512;
constexpr int kChainMaxPics = 64;      // == kMaxBatch (engine.h)
constexpr int kChainStride = kChainBits + kChainMaxRows * kChainRowWords;
// behind the pictures' blocks: the launch-wide abort word, the census, the record of the first give-up and the launch's wait limit (32 ints, layout below),
// then four time stamps per picture (diagnostic launches: ChainView::stamp)
constexpr int kChainTailHead = 32;
constexpr int kChainTail = kChainTailHead + 4 * kChainMaxPics;
// words of the tail head: [0] abort, [1..7] census (ChainView), [8] give-ups, [9..14] record of the first give-up (record_first_giveup),
// [15] wait limit of this launch in 100 MHz ticks (written by the host, Engine::launch; 0 = kWaitTicks), [16..21] second half of the record:
// what an atomic read-modify-write of the counter returned when the wait gave up (the value at the point of coherence, beside what the polls saw), the waiter's
// XCC_ID and HW_ID1 registers, its number of looks, the clock gaps its timer had skipped and the largest of them; [22..23] census of the reference windows
constexpr int kTailWaitLimit = 15, kTailRmw = 16, kTailXcc = 17, kTailHwId = 18, kTailSpins = 19, kTailGaps = 20, kTailGapMax = 21;
constexpr int kSpinLimit = 1 << 20;    // polls before a wait of the STAGE kernels gives up (about a second; a healthy wait takes microseconds)
// Waits of a chain launch are bounded by TIME THE WAVE SPENT WAITING, in ticks of the 100 MHz wall clock.  The limit comes with the launch (tail word
// kTailWaitLimit): about ten times the longest healthy wait -- a band is resident from the launch's first microsecond and may wait for the launch's last
// reconstruction group, so the longest healthy wait is the launch itself: 1.3-2.1 ms at 1080p (profiles/r04_chain_timeline.txt), more for bigger
// pictures and deeper chains (Engine::launch scales it).  Rounds 2-5 used a flat 100 ms of WALL time.  Round 5's two unexplained give-ups (2 of 41,800
// launches, profiles/r05_chain_soak.txt: a band one step short of what its neighbour needed, nothing in the code that could stall it) fit a timer that
// keeps running while its wave does not -- and round 6 SAW that happen: in each of three soaks of ~13,700 launches several launches had a wait whose
// clock jumped by 23-33 ms between two looks (profiles/r06_chain_soak.txt), typically once per run, when the load on the device drops at the end of a
// pass.  Every wave of the launch stands still that long while the clock does not; the first wave to wake up finds its time used up with the counter exactly
// where a healthy run would have it.  (Not queue creation, not the freeing of output buffers: both were tried; what is left -- clock / power state
// changes of the device, the driver's scheduler -- cannot be observed or changed by an ordinary user of this pool.)  The timers do not need to know:
// they measure waiting, not absence.  The clock is read every 64 looks, 64 looks take at most ~3 ms (wait_final's naps), and a jump of more than
// kGapTicks between two readings is time the wave was not run -- it is taken out of the wait and recorded (WaitClock::gaps, the launch's evidence words:
// Engine::complete counts them).  A give-up then means what it says: the wave looked for `limit` of its own running time and the counter did not move.
// The engine decodes the launch's pictures again with the stage kernels (Engine::recover) either way: a timeout costs time, not correctness.
constexpr uint32_t kWaitTicks = 2u * 1000u * 1000u;      // 20 ms: the default when the host wrote no limit
constexpr uint32_t kGapTicks = 500u * 1000u;             // 5 ms between two clock readings of a waiting wave: it was not run
struct WaitClock { uint32_t t0 = 0, last = 0, limit = 0, gaps = 0, gap_max = 0; };
// launch = the launch's abort word (tail word 0), nullptr outside chain launches
__device__ __forceinline__ bool wait_expired(int spins, WaitClock &c, const int *launch) {          // (32 bits of the clock: differences survive the wrap)
    if (spins & 63) return false;
    const uint32_t now = (uint32_t)wall_clock64();
    if (spins == 64) { c.t0 = c.last = now;
        const uint32_t lim = launch ? (uint32_t)__hip_atomic_load(launch + kTailWaitLimit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        c.limit = lim ? lim : kWaitTicks; return false; }
    const uint32_t d = now - c.last;
    c.last = now;
    if (d > kGapTicks) { c.t0 += d; c.gaps++; c.gap_max = d > c.gap_max ? d : c.gap_max; }
    return now - c.t0 > c.limit;
}
// a wait that saw clock gaps leaves them in the batch's evidence words (host-pinned, behind the error words: plain stores, the last writer wins --
// evidence, not accounting).  evid = err + kChainMaxPics
__device__ __forceinline__ void note_gaps(const WaitClock &c, int *evid) {
    if (c.gaps && evid) { __hip_atomic_store(evid, (int)c.gaps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(evid + 1, (int)c.gap_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}
enum : int { CHAIN_ERR_FIN_TIMEOUT = 1, CHAIN_ERR_BITS_TIMEOUT = 2, CHAIN_ERR_RING_TIMEOUT = 4, CHAIN_ERR_INTRA_TIMEOUT = 8, CHAIN_ERR_IFIN_TIMEOUT = 16,
               // host side (Engine::recover): the pictures of a timed-out chain launch could not be decoded again from intact references
               CHAIN_ERR_NOT_RECOVERED = 32,
               // a kernel instantiation met a motion record it was compiled without (k_recon_inter<HAS_BI = false> and a two-list record: ADVICE r5)
               CHAIN_ERR_BAD_RECORD = 64 };

// Steps between a macroblock row and the row below it (round 4).  Clause 8.7 orders the macroblocks in raster order, and within one the vertical edges (V)
// before the horizontal edges (H). Which filters touch the same samples: V(x, y) the macroblock and columns 12..15 of (x - 1, y); H(x, y) the macroblock and
// rows
// 12..15 of (x, y - 1). So V(x, y) needs H(x - 1, y) and nothing of the row above, and H(x, y) needs V(x, y) and -- for columns 13..15 of those four rows,
// which
// the left edge of (x + 1, y - 1) changes -- V(x + 1, y - 1), but NOT H(x + 1, y - 1).  Rounds 1-3 ran whole macroblocks per step, which needs two steps
// between rows (s = x + 2y: 254 steps at 1080p, 508 at 4K).  With the step cut into a V phase and an H phase by a second barrier, (x + 1, y - 1) and (x, y)
// share a step: s = x + y, 188 steps at 1080p (-26 %), 375 at 4K, every filter still sees exactly the samples raster order would give it (no two filters
// that touch a common sample change their order).  As built (deblock_device.h `step`): step s of row y = H + store of macroblock s - 1 - y, then V of
// macroblock s - y, ONE barrier per step; the final samples of macroblock (X, Y) are therefore stored in step X + Y + 1.
constexpr int kRowLag = 1;
static_assert(kRowLag == 1 || kRowLag == 2, "row lag of the deblocking wavefront");

typedef __attribute__((address_space(1))) int gint;

// loads / stores of data that another workgroup of the same launch produces or consumes
__device__ __forceinline__ int ld_coh(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_coh(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coh(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Wide write-through stores (16 / 8 bytes per lane, `sc1` = agent scope on gfx950).  The atomic builtins stop at 8 bytes and split a row into dword
// stores, each of which the memory side handles as its own partial-sector write: WRITE_SIZE showed 68 MB per 1080p picture for k_chain against
// 6.3 MB of samples.  Inline assembly is safe here: the compiler does not count these stores in its vmcnt bookkeeping, so the waits it inserts for
// later loads only become more conservative; the waits that matter for the stores themselves are explicit (s_waitcnt vmcnt) in the callers.
typedef uint32_t jm_v4u __attribute__((ext_vector_type(4)));
typedef uint32_t jm_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_wt16(void *p, uint4 v) {
    const jm_v4u t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"((__attribute__((address_space(1))) void *)p), "v"(t) : "memory");
}
__device__ __forceinline__ void st_wt8(void *p, uint2 v) {
    const jm_v2u t = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"((__attribute__((address_space(1))) void *)p), "v"(t) : "memory");
}
// reference samples: COH = the reference picture may have been written by this launch -- `sc1` loads, which every level that is not coherent across the
// XCDs passes through.  Round 4: NOT the atomic builtins any more.  An atomic load can neither be merged nor moved, and with 36 of them per sample in the
// literal path and a dozen per lane in the window paths the reconstruction role of the chain kernels needed 146 registers against the 96 of the same
// code with plain loads -- which made k_chain a 3-waves-per-SIMD kernel, and its reconstruction waves (a latency-bound role: they wait for other
// workgroups and for uncached loads) are what bounds 2..16 streams.  Now they are ordinary buffer loads with the sc1 policy bit (aux 16 on gfx940+):
// same instruction on the memory side (buffer_load_* ... sc1), freely scheduled by the compiler, and addressed by a 32-bit offset into ONE descriptor
// over the handle's surface block (PicParams.surf_base; jobs.h) instead of a 64-bit pointer per load: 101 / 108 registers.  What ordered the atomic loads
// behind the polls of wait_final -- the data dependency on the counter -- is kept by a compiler barrier there (the hardware issues in program order).
struct RefBuf {
    __amdgpu_buffer_rsrc_t rsrc; const uint8_t *base;
    __device__ __forceinline__ explicit RefBuf(const uint8_t *b) : rsrc(__builtin_amdgcn_make_buffer_rsrc((void *)b, 0, -1, 0x00020000)), base(b) {}
    __device__ __forceinline__ int off(const uint8_t *p) const { return (int)((uint32_t)(uintptr_t)p - (uint32_t)(uintptr_t)base); }
};
constexpr int kAuxSc1 = 16;
// a single coherent byte through a plain pointer (hevc_kernels.hip: a few edge samples per coding tree block)
__device__ __forceinline__ int ld_coh8(const uint8_t *p) { return (int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (the plain loads go through address-space-1 pointers: a generic pointer makes them FLAT instructions, which count on lgkmcnt as well as vmcnt, so every
// wait for an LDS result would also wait for the reference loads in flight)
#define JM_GLOBAL_AS __attribute__((address_space(1)))
template <bool COH> __device__ __forceinline__ int ld_ref8(const RefBuf &rb, const uint8_t *p) {
    if (COH) return (int)__builtin_amdgcn_raw_buffer_load_b8(rb.rsrc, rb.off(p), 0, kAuxSc1);
    return (int)*(const JM_GLOBAL_AS uint8_t *)p;
}
template <bool COH> __device__ __forceinline__ uint32_t ld_ref16(const RefBuf &rb, const uint8_t *p) {        // p even
    if (COH) return (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rb.rsrc, rb.off(p), 0, kAuxSc1);
    return *(const JM_GLOBAL_AS uint16_t *)p;
}
template <bool COH> __device__ __forceinline__ uint32_t ld_ref32(const RefBuf &rb, const uint8_t *p) {
    if (COH) return __builtin_amdgcn_raw_buffer_load_b32(rb.rsrc, rb.off(p), 0, kAuxSc1);
    return *(const JM_GLOBAL_AS uint32_t *)p;
}
// the same with a wave-uniform plane address and a 32-bit byte offset per lane (scalar base + vector offset: no 64-bit address arithmetic per lane)
template <bool COH> __device__ __forceinline__ uint32_t ld_ref32(const RefBuf &rb, const uint8_t *plane, uint32_t off) {
    if (COH) return __builtin_amdgcn_raw_buffer_load_b32(rb.rsrc, rb.off(plane) + (int)off, 0, kAuxSc1);
    return *(const JM_GLOBAL_AS uint32_t *)((const JM_GLOBAL_AS uint8_t *)plane + off);
}

// A bounded wait gave up: record it in the batch's error array (pinned HOST memory mapped into the device, one word per picture; the engine
// reads it when the batch retires and reports a decode error for the handle -- a damaged hand-over is never silent).
__device__ __forceinline__ void report_wait_timeout(int *err_word, int code) {
    if (err_word) __hip_atomic_store(err_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The FIRST wait of a launch that runs out of time (not the ones that follow the abort word) leaves a record behind the abort word: which wait, of which
// picture, where, what it needed and what it saw -- when the launch retires every counter is complete again (after the abort the waits let go), so this
// is the only trace of where a launch was stuck (Engine::dump_chain_state prints it).  launch = the abort word's address.
__device__ __forceinline__ void record_first_giveup(int *launch, int code, int pic, int where, int need, int seen0, int seen1) {
    if (__hip_atomic_fetch_add(launch + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
    launch[9] = code; launch[10] = pic; launch[11] = where; launch[12] = need; launch[13] = seen0; launch[14] = seen1;
}
// ... and the second half: `ctr` = the counter (or bitmap word) the wait polled, read once more by an atomic read-modify-write -- performed where the
// device's memory is coherent, so it cannot be served by a stale line: a value that differs from what the polls saw means "published, not seen";
// the same value means the producer really stood there.  Plus where the waiter ran and what its timer went through.
__device__ __forceinline__ void record_giveup_evidence(int *launch, const int *ctr, int spins, const WaitClock &c) {
    if (__hip_atomic_load(launch + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1 || launch[kTailSpins]) return;      // the first give-up only
    launch[kTailRmw] = ctr ? __hip_atomic_fetch_or(const_cast<int *>(ctr), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
    launch[kTailXcc] = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);       // XCC_ID[3:0] (hardware register 20 on gfx940+)
    launch[kTailHwId] = (int)__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 23);     // HW_ID1: wave, SIMD, CU, shader array, shader engine
    launch[kTailSpins] = spins; launch[kTailGaps] = (int)c.gaps; launch[kTailGapMax] = (int)c.gap_max;
}

// Bits of one macroblock row of the reconstruction bitmap as a band workgroup sees them: `known` macroblocks from the left are known to be in memory
// (the run of set bits found by the last poll), so a band that trails the reconstruction polls once per row.  Uniform per 16-lane group.
// want = this group needs macroblock x now; returns false when the wait gave up (reported by the caller).
// tag = picture << 16 | row (for the record of the first give-up)
__device__ __forceinline__ bool wait_row_bit(const uint32_t *bits_row, int &known, bool want, int x, int *abort_word, int tag = 0, int *evid = nullptr) {
    bool pending = want && x >= known;
    int spins = 0; WaitClock t0;
    for (;;) {
        if (pending) {
            const uint32_t m = ld_coh(bits_row + (x >> 5)) >> (x & 31);
            if (m & 1) { known = x + (m == 0xffffffffu ? 32 : __builtin_ctz(~m)); pending = false; }   // the run of set bits that starts at x
        }
        if (!__builtin_amdgcn_ballot_w64(pending)) { note_gaps(t0, evid); return true; }
        const bool expired = wait_expired(++spins, t0, abort_word);
        if (expired && pending) { record_first_giveup(abort_word, CHAIN_ERR_BITS_TIMEOUT, tag >> 16, (tag & 0xffff) << 16 | (x & 0xffff), x, known,
            (int)ld_coh(bits_row + (x >> 5)));
            record_giveup_evidence(abort_word, (const int *)(bits_row + (x >> 5)), spins, t0); }
        if (expired || ((spins & 255) == 0 && ld_coh(abort_word))) { note_gaps(t0, evid); known = 0x7fffffff; return false; }   // gave up: do not wait again
        // a band may be resident long before its rows are reconstructed
        if (spins < 64) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
    }
}
// the same for a step counter (`fin` of the intra wavefront): wait until *ctr >= need
__device__ __forceinline__ bool wait_counter(const int *ctr, int &known, bool want, int need, int *abort_word, int tag = 0, int *evid = nullptr) {
    bool pending = want && known < need;
    int spins = 0; WaitClock t0;
    for (;;) {
        if (pending) { known = ld_coh(ctr); pending = known < need; }
        if (!__builtin_amdgcn_ballot_w64(pending)) { note_gaps(t0, evid); return true; }
        const bool expired = wait_expired(++spins, t0, abort_word);
        if (expired && pending) { record_first_giveup(abort_word, CHAIN_ERR_IFIN_TIMEOUT, tag >> 16, tag & 0xffff, need, known, 0);
            record_giveup_evidence(abort_word, ctr, spins, t0); }
        if (expired || ((spins & 255) == 0 && ld_coh(abort_word))) { note_gaps(t0, evid); known = 0x7fffffff; return false; }
        if (spins < 64) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(16);
    }
}

struct ChainView {
    int *base;                          // the batch's control buffer (device): kChainStride ints per picture, then the launch-wide abort word
    int *err;                           // the batch's error words (host-pinned, device-visible), one per picture
    __device__ __forceinline__ int *pic(int idx) const { return base + (size_t)idx * kChainStride; }
    // Once one wait of a launch has given up, the launch is damaged anyway: every other wait then gives up as soon as it looks (every 256 polls),
    // so a broken hand-over costs about one timeout, not one per waiting wave.
    __device__ __forceinline__ int *abort_word() const { return base + (size_t)kChainMaxPics * kChainStride; }
    // Census of the launch (round 4, what a give-up is diagnosed with: Engine::dump_chain_state): behind the abort word, workgroups started / finished
    // per role and the highest work-list index started.  Only in diagnostic launches (JM_AMD_DEC_CENSUS; `on` comes with the kernel's `pub` argument):
    // three atomics per workgroup on ONE cache line are 180 k of them in a 2 ms launch of eight streams -- the rate at which a single address saturates
    // -- and a wave's loads retire behind its own older atomics: switched on for every launch they cost 8 / 16 streams 11 % / 9 % of their rate.
    enum : int { CENSUS_RECON_STARTED = 1, CENSUS_RECON_DONE = 2, CENSUS_BAND_STARTED = 3, CENSUS_BAND_DONE = 4, CENSUS_MAX_GROUP = 5,
                 CENSUS_WAIT_TICKS = 6, CENSUS_RECON_TICKS = 7,
                 // reconstruction workgroups that fetched ONE shared reference window for their four macroblocks / whose macroblocks fetched their own (recon_device.h)
                 CENSUS_QUAD = 22, CENSUS_PRIVATE = 23 };    // 100 MHz ticks wave 0 of the reconstruction workgroups spent in wait_final / in all
    bool census_on = false;
    __device__ __forceinline__ void census(int what, int value = 1) const {
        if (!census_on || threadIdx.x != 0) return;
        if (what == CENSUS_MAX_GROUP) (void)__hip_atomic_fetch_max(abort_word() + what, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else (void)__hip_atomic_fetch_add(abort_word() + what, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // Time line of a diagnostic launch (JM_AMD_DEC_CENSUS; printed with JM_AMD_DEC_CHAIN_TIMELINE, Engine::dump_chain_state): per picture the first start
    // and the last end of its reconstruction workgroups and of its bands, in ticks of the 100 MHz clock.  "First" is kept as the maximum of 2^30 - t.
    enum : int { STAMP_RECON_FIRST = 0, STAMP_RECON_LAST = 1, STAMP_BAND_FIRST = 2, STAMP_BAND_LAST = 3 };
    __device__ __forceinline__ void stamp(int pic_idx, int which) const {
        if (!census_on || threadIdx.x != 0) return;
        const int t = (int)((uint32_t)wall_clock64() & 0x3fffffffu);
        (void)__hip_atomic_fetch_max(abort_word() + kChainTailHead + 4 * pic_idx + which, (which & 1) ? t : 0x40000000 - t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // Wait until every sample of the rectangle [.., xmax] x [ymin, ymax] (luma coordinates, already clamped to the picture) of the picture
    // with chain index `dep` is final.  The final value of sample (px, py) is stored by the deblocking step of macroblock
    // ((px + 4) >> 4, (py + 4) >> 4) (clamped; deblock_device.h stores (-4,-4)-shifted blocks, chroma the same in its own units), i.e. in
    // wavefront step X + 2Y of the band that holds row Y.  A rectangle touches at most two bands.  Per lane; dep < 0 = nothing to wait for.
    // Returns false when the wait gave up.
    // row_lag = steps between a macroblock row and the next in the deblocking wavefront (deblock_device.h kRowLag): macroblock (X, Y) is step X + row_lag * Y
    __device__ __forceinline__ bool wait_final(int dep, int xmax, int ymin, int ymax, int mb_w, int mb_h, int row_lag) const {
        bool pending = dep >= 0;
        const int *fin = pic(pending ? dep : 0) + kChainFin;
        const int xs = min((xmax + 4) >> 4, mb_w - 1), yhi = min((ymax + 4) >> 4, mb_h - 1), ylo = min(max((ymin + 4) >> 4, 0), yhi);
        const int bhi = yhi >> 4, blo = ylo >> 4;
        const int store_lag = row_lag == 1 ? 1 : 0;                // one row per step: macroblock (X, Y) is stored in step X + Y + 1 (deblock_device.h)
        const int need_hi = xs + row_lag * yhi + 1 + store_lag, need_lo = xs + row_lag * (blo * 16 + 15) + 1 + store_lag;
        // Polls cost: a waiting wave that looks every 0.2 us sends two or three L2 requests per microsecond, and a chain launch keeps thousands of waves
        // waiting (one stream: the whole device waits for ONE picture's bands) -- all onto a handful of counter lines, in front of the stores that move
        // those counters.  The counters are step numbers, so a wave knows how far away its samples are: it naps 0.75 us for every step still missing (a
        // deblocking step takes >= 2.3 us, a band that has not started begins at step row_lag * 16 * band: it never oversleeps), at most 63 naps in a
        // row, and looks again.  Waits that are nearly over poll as before.  (4 / 8 streams +4 %, profiles/r04_ab10_wait_naps.json; the bytes fetched
        // from MEMORY did not change: the polls hit in the L2.)
        constexpr int kNapSlack = 3, kTopNap = 32;                  // naps in a row: at most 2 * kTopNap - 1 (47 us)
        const int first_hi = row_lag * 16 * bhi, first_lo = row_lag * 16 * blo;
        int spins = 0; WaitClock t0;
        for (;;) {
            int missing = 0;                                        // steps until the rectangle is final, as far as the last look could tell
            if (pending) {
                // (the luma and the chroma counter in one round trip: a reconstruction workgroup lives ~25 us, and a look that passes is the common case)
                const int fy = ld_coh(fin + bhi), fc = ld_coh(fin + 32 + bhi);
                missing = need_hi - max(min(fy, fc), first_hi);
                if (missing <= 0 && blo != bhi) { const int gy = ld_coh(fin + blo), gc = ld_coh(fin + 32 + blo);
                    missing = need_lo - max(min(gy, gc), first_lo); }
                pending = missing > 0;
            }
            // (nothing that reads the picture moves above the polls)
            if (!__builtin_amdgcn_ballot_w64(pending)) { note_gaps(t0, err ? err + kChainMaxPics : nullptr); asm volatile("" ::: "memory"); return true; }
            const bool expired = wait_expired(++spins, t0, abort_word());
            if (expired && pending) { record_first_giveup(abort_word(), CHAIN_ERR_FIN_TIMEOUT, dep, bhi << 16 | (xs & 0xffff), need_hi, ld_coh(fin + bhi),
                ld_coh(fin + 32 + bhi));
                record_giveup_evidence(abort_word(), fin + bhi, spins, t0); }
            if (expired || ((spins & 255) == 0 && ld_coh(abort_word()))) { note_gaps(t0, err ? err + kChainMaxPics : nullptr); st_coh(abort_word(), 1);
                return false; }
            // the wave goes on when its LAST lane is served: nap by the largest number of missing steps (bit by bit: seven ballots)
            int naps = 0;
            for (int t = kTopNap; t; t >>= 1) if (__builtin_amdgcn_ballot_w64(pending && missing - kNapSlack >= naps + t)) naps += t;
            for (int i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(28);       // 28 x 64 cycles: 0.75 us at 2.4 GHz
            __builtin_amdgcn_s_sleep(4);
        }
    }
};

}  // namespace jmamd
