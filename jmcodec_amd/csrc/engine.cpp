// jmcodec_amd/csrc/engine.cpp -- see engine.h.
#include "engine.h"
#include <sys/prctl.h>
#include "chain_order.h"
#include "decoder.h"
#include "kernels.h"
#include "hevc_kernels.h"
#include "numa.h"
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace jmamd {

// developer aid (JM_AMD_DEC_LANE_TRACE=<max lines>): the lanes' time line on stderr -- batches launched / retired, pictures launched ahead of their turn,
// decoders left out of an ordinary batch and what held them (profiles/r06_lane_trace.txt)
static long g_lane_trace = getenv("JM_AMD_DEC_LANE_TRACE") ? atol(getenv("JM_AMD_DEC_LANE_TRACE")) : 0;
static double trace_ms() { static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
#define LANE_TRACE(...) do { if (g_lane_trace > 0) { g_lane_trace--; fprintf(stderr, "lane-trace %9.3f " , trace_ms()); fprintf(stderr, __VA_ARGS__); } } while (0)

static std::mutex g_engines_m;
static Engine *g_engines[64] = {nullptr};

Engine *Engine::get(int device) {
    std::lock_guard<std::mutex> lk(g_engines_m);
    if (device < 0 || device >= 64) return nullptr;
    if (!g_engines[device]) {
        Engine *e = new Engine(device);               // intentionally leaked: lives as long as the process
        if (!e->ok_) return nullptr;
        g_engines[device] = e;
    }
    return g_engines[device];
}

// developer aid (JM_AMD_DEC_MEMTRACE=1): resident memory of the process at the steps of engine / handle set-up
void mem_trace(const char *tag) {
    static const bool on = getenv("JM_AMD_DEC_MEMTRACE") != nullptr;
    if (!on) return;
    long pages = 0, res = 0;
    if (FILE *f = fopen("/proc/self/statm", "r")) { if (fscanf(f, "%ld %ld", &pages, &res) != 2) res = 0; fclose(f); }
    fprintf(stderr, "jm_amd_dec: memtrace %-44s %8.1f MB resident\n", tag, res * 4096.0 / 1e6);
}

Engine::Engine(int device) : device_(device) {
    (void)trace_ms();                                 // (the diagnostic clock starts with the first engine)
    mem_trace("engine: start");
    if (const char *e = getenv("JM_AMD_DEC_CHAIN_DEPTH")) { chain_depth_ = std::max(1, std::min(atoi(e), 16)); chain_depth_few_ = chain_depth_.load(); }
    if (const char *e = getenv("JM_AMD_DEC_CHAIN_STREAMS")) chain_max_streams_ = std::max(0, atoi(e));
    if (const char *e = getenv("JM_AMD_DEC_CHAIN_LAG")) chain_lag_steps_ = std::max(kMinChainLag, std::min(atoi(e), 1024));
    if (const char *e = getenv("JM_AMD_DEC_CHAIN_LINGER")) chain_linger_streams_ = std::max(0, atoi(e));
    if (const char *e = getenv("JM_AMD_DEC_CROSS_LANE")) cross_lane_ = atoi(e) != 0;
    if (const char *e = getenv("JM_AMD_DEC_FILL_LINGER_US")) fill_linger_ns_ = std::max(0, atoi(e)) * 1000ll;
    if (const char *e = getenv("JM_AMD_DEC_EARLY_INTRA")) early_intra_ = atoi(e) != 0;
    if (hipSetDevice(device_) != hipSuccess) return;
    // The copy streams (job-list uploads) get a stream priority of their own: the runtime keeps a separate pool of hardware queues per priority, so an upload
    // never shares a queue with a lane's kernels -- which streams share one is otherwise decided by the order in which the runtime first sees them, and this round
    // met both outcomes by accident (a second copy stream created beside the first: C3 / C2 -8 %; created last: 8 streams -19 %; profiles/r06_copy_streams.txt).
    if (const char *e = getenv("JM_AMD_DEC_COPY_STREAMS")) n_copy_ = std::max(1, std::min(atoi(e), 4));
    int prio_lo = 0, prio_hi = 0, copy_prio = 0;
    hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);          // (numerically lower = higher priority)
    { const char *e = getenv("JM_AMD_DEC_COPY_PRIORITY"); const int want = e ? atoi(e) : 1;      // 1 = high (default), 0 = the lanes' own, -1 = low
      copy_prio = want > 0 ? prio_hi : (want < 0 ? prio_lo : 0); }
    for (int k = 0; k < n_copy_; k++) if (hipStreamCreateWithPriority(&copy_streams_[k], hipStreamNonBlocking, copy_prio) != hipSuccess) {
        (void)hipGetLastError();
        if (hipStreamCreateWithFlags(&copy_streams_[k], hipStreamNonBlocking) != hipSuccess) return; }
    copy_stream_ = copy_streams_[0];
    mem_trace("engine: copy stream");
    for (auto &ln : lanes_) {
        hipStream_t s, p, q;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&p, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) return;
        ln.stream = s; ln.pack_stream = p; ln.pre_stream = q;
        mem_trace("engine: lane streams");
        for (auto &b : ln.ring) {
            if (hipHostMalloc((void **)&b.h_pics, sizeof(PicParams) * kMaxBatch, hipHostMallocDefault) != hipSuccess) return;
            if (hipMalloc((void **)&b.d_pics, sizeof(PicParams) * kMaxBatch) != hipSuccess) return;
            if (hipHostMalloc((void **)&b.h_hpics, sizeof(HevcPicParams) * kMaxBatch, hipHostMallocDefault) != hipSuccess) return;
            if (hipMalloc((void **)&b.d_hpics, sizeof(HevcPicParams) * kMaxBatch) != hipSuccess) return;
            if (hipMalloc((void **)&b.d_progress, sizeof(int) * kMaxBatch * kHevcProgressStride) != hipSuccess) return;
            // + the launch-wide tail (chain_common.h)
            if (hipMalloc((void **)&b.d_ctl, sizeof(int) * ((size_t)kMaxBatch * chain_ctl_ints() + chain_tail_ints())) != hipSuccess) return;
            // error words, one per picture, then two evidence words of the batch (clock gaps its waits saw: chain_common.h note_gaps)
            if (hipHostMalloc((void **)&b.h_err, sizeof(int) * (kMaxBatch + 4), hipHostMallocMapped) != hipSuccess) return;
            if (hipHostGetDevicePointer((void **)&b.d_err, b.h_err, 0) != hipSuccess) return;
            memset(b.h_err, 0, sizeof(int) * (kMaxBatch + 4));
            if (hipHostMalloc((void **)&b.h_groups, sizeof(uint32_t) * kMaxChainGroups, hipHostMallocDefault) != hipSuccess) return;
            if (hipMalloc((void **)&b.d_groups, sizeof(uint32_t) * kMaxChainGroups) != hipSuccess) return;
            if (hipHostMalloc((void **)&b.h_jobs, sizeof(PackJob) * 4 * kMaxBatch, hipHostMallocDefault) != hipSuccess) return;
            if (hipMalloc((void **)&b.d_jobs, sizeof(PackJob) * 4 * kMaxBatch) != hipSuccess) return;
            if (hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) return;
            if (hipEventCreateWithFlags(&b.kdone, hipEventDisableTiming) != hipSuccess) return;
            if (hipEventCreateWithFlags(&b.packed, hipEventDisableTiming) != hipSuccess) return;
            if (hipEventCreateWithFlags(&b.pre_done, hipEventDisableTiming) != hipSuccess) return;
            for (auto &e : b.pev) if (hipEventCreate(&e) != hipSuccess) return;
        }
    }
    mem_trace("engine: lane buffers");
    // The no-deadlock argument of a chain launch (chain.hip) needs its band workgroups -- resident for their whole wavefront -- to leave room for the
    // reconstruction groups they wait for: at most half of what the device holds.  Taken from the device in use (a compute partition, a smaller part
    // or a build with other register counts holds fewer than the constants assume); launches whose first pictures do not fit run unchained.
    { const int r = chain_resident_workgroups(false), ri = chain_resident_workgroups(true);
      if (r > 0) chain_bands_max_ = std::min(kMaxChainBands, r / 2);
      if (ri > 0) chain_bands_max_intra_ = std::min(kMaxChainBandsIntra, ri / 2);
      if (getenv("JM_AMD_DEC_VERBOSE")) fprintf(stderr,
          "jm_amd_dec: device %d holds %d / %d chain workgroups (plain / with the intra role): band budget %d / %d\n", device_, r, ri, chain_bands_max_,
          chain_bands_max_intra_); }
    mem_trace("engine: occupancy queries");
    ok_ = true;
    numa_node_ = numa_node_of_device(device_, true);
    kfd_gpu_id_ = getenv("JM_AMD_DEC_IGNORE_SHARED_GPU") ? 0 : kfd_gpu_id_of_device(device_);
    th_ = std::thread([this] { pthread_setname_np(pthread_self(), "jm-engine"); numa_bind_this_thread(numa_node_); run(); });
    th_.detach();
}

unsigned long long Engine::upload(uint8_t *dev, const uint8_t *host, size_t n, ihipEvent_t *ev, bool spread) {
    std::lock_guard<std::mutex> lk(um_);              // keeps (copy, event, sequence number) consistent with stream order
    hipSetDevice(device_);
    unsigned long long seq = ++upload_seq_;
    if (!spread) while (seq % (unsigned)n_copy_) seq = ++upload_seq_;      // (the stream is seq mod n_copy_: Engine::launch finds it again from the number)
    hipStream_t cs = copy_streams_[seq % (unsigned)n_copy_];
    hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, cs);
    hipEventRecord(ev, cs);
    return seq;
}

void Engine::submit(EnginePic &&p) {
    {
        std::lock_guard<std::mutex> lk(m_);
        if (p.codec == 0) {
            const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
            auto it = std::find_if(recent_.begin(), recent_.end(), [&](const Recent &r) { return r.dec == p.dec; });
            if (it == recent_.end()) { recent_.push_back(Recent{p.dec, now, 0}); it = recent_.end() - 1; }
            it->t = now;
            if (p.has_picture) it->mbs = p.mb_w * p.mb_h;
        }
        p.seq = p.dec->engine_state().next_seq++;
        if (p.dec->engine_state().n_pending++ == 0) decoders_pending_++;
        pending_.push_back(std::move(p));
        pending_gen_++;
    }
    cv_.notify_one();
}

bool Engine::set_knob(const std::string &key, long long v) {
    // an explicit setting ends the pause that follows a recovered chain launch, and has the next batch look again whether the GPU is shared
    if (key.rfind("chain_", 0) == 0) { chain_block_until_ns_ = 0; shared_checked_ns_ = 0; }
    if (key == "chain_depth") { if (v <= 0) { chain_depth_ = 8; chain_depth_few_ = 16; }      // 0: the defaults (8; 16 with one or two active streams)
        else { chain_depth_ = (int)std::min(v, 16ll); chain_depth_few_ = chain_depth_.load(); } }
    else if (key == "chain_lag") chain_lag_steps_ = (int)std::max((long long)kMinChainLag, std::min(v, 1024ll));
    else if (key == "chain_streams") chain_max_streams_ = (int)std::max(0ll, v);
    else if (key == "debug_stall") debug_stall_ = (int)v;
    else if (key == "debug_no_bi") debug_no_bi_ = (int)v;          // test aid: launch k_recon_inter WITHOUT its two-list code whatever the batch holds (the kernel must say so)
    else if (key == "early_intra_ahead") early_intra_ahead_ = (int)std::max(1ll, std::min(v, 64ll));      // (tests: 1 = run ahead whenever the hazards allow)
    else return false;
    return true;
}

EngineStats Engine::stats() { std::lock_guard<std::mutex> lk(sm_); return st_; }

// m_ held.  Lanes other than `lane_idx`: is the device done (decode kernels AND the pack-out that reads the surfaces: event `packed`) with every picture of d
// that comes before `seq` in decode order?  Pictures of the lane itself are ordered by its stream.
bool Engine::others_done(Decoder *d, int lane_idx, unsigned long long seq) {
    for (int li = 0; li < kLanes; li++) {
        if (li == lane_idx) continue;
        Lane &o = lanes_[li];
        for (int k = 0; k < o.inflight; k++) {
            Batch &ob = o.ring[(o.tail + k) % kBatchRing];
            bool mine = false;
            for (auto &p : ob.pics) if (p.dec == d && p.seq < seq) { mine = true; break; }
            if (mine && hipEventQuery(ob.packed) != hipSuccess) { (void)hipGetLastError(); return false; }
        }
    }
    return true;
}

// m_ held: every surface the pictures of d that are in flight (any lane) decode into, reference or display
void Engine::inflight_masks(Decoder *d, uint32_t &touched) {
    touched = 0;
    for (auto &o : lanes_) for (int k = 0; k < o.inflight; k++) for (auto &p : o.ring[(o.tail + k) % kBatchRing].pics) if (p.dec == d) {
        touched |= p.ref_mask | p.out_mask;
        if (p.has_picture) touched |= 1u << (p.codec == 0 ? p.pp.cur : p.hp.cur);
    }
}

// Take the first pending picture of every decoder (arrival order) that belongs to this lane and may run now; then, on the lanes of ordinary
// pictures, extend every decoder's share of the batch by its NEXT pictures as long as they can run inside the chain kernel (chain.hip):
// consecutive P / B pictures of a stream then share one launch and follow each other at macroblock granularity instead of one per batch.
bool Engine::form(Lane &ln, int lane_idx, Batch &b) {
    b.pics.clear(); b.any_chain = false; b.chain_with_intra = false; b.max_depth = 1;
    // chain launches are formed while few streams have pictures ready (a wide batch fills the GPU anyway); then the I picture of an IDR period stays on
    // its stream's ordinary lane, where it becomes the first picture of a chain (k_chain_i), instead of going to the intra lane
    // The regime follows the number of ACTIVE streams (handles that submitted a picture in the last 50 ms), not the number that happen to have a picture
    // pending right now: with 20 or 32 streams the pending set dips below the threshold now and then, and a stream may only change lane once its pictures
    // in flight have retired.  Few streams: chain launches.  Many: stage kernels, one batch across all streams.
    bool chaining = false;
    int depth_now = chain_depth_;
    {
        const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        recent_.erase(std::remove_if(recent_.begin(), recent_.end(), [&](const Recent &r) { return now - r.t > 50ll * 1000 * 1000; }), recent_.end());
        const int n_active = (int)recent_.size();
        // A chain launch needs the whole GPU (its waits assume its bands stay resident, chain.hip).  When another process has compute queues on this
        // device -- a second rank of the same job, another tenant -- no chain launches are formed at all, instead of letting them time out against the
        // other process's kernels and decoding their pictures again (Engine::recover: correct, but every such launch costs 100 ms).  The look itself
        // (a walk through KFD's sysfs) happens on the engine thread OUTSIDE m_ (Engine::look_for_other_users): submit() never waits for it.
        // "few streams" is meant in 1080p streams: a 4K stream fills the stage kernels like four of them.  (C2, 16 streams of 4K: the stage kernels run at the
        // chains' rate, 2.4-2.5 k against 2.3-2.6 k frames/s -- and the first chain launch of such a run sometimes gave up)
        double load = 0;
        for (const Recent &r : recent_) load += std::max(1.0, r.mbs / 8160.0);
        chaining = chain_depth_ > 1 && n_active > 0 && load <= (double)chain_max_streams_ && now >= chain_block_until_ns_ &&
            !gpu_shared_.load(std::memory_order_relaxed);
        // One or two streams: a chain launch is as long as what its stream(s) fed while the previous launch ran.  Queueing a SECOND launch behind a running one
        // as soon as a picture or two are there splits that supply into a short launch and a long one, and a short chain costs nearly what a long one costs
        // (the first picture's wavefront, ~0.7 ms at 1080p).  So while a launch is in flight the next one is only formed when it would be a full chain;
        // otherwise it waits until the lane is idle and takes everything that has arrived by then (two streams 6.4 k -> 7.2 k frames/s; one stream within
        // the noise: its caller's loop, not the chain length, is what bounds it.  Forming the next launch when the running one is four fifths through, to
        // hide the host's turnaround, was tried and lost the gain again: measured, not kept).
        // one or two streams: chains of up to chain_depth_few_ pictures (a launch costs ~0.85 ms + ~0.1 ms per picture: engine.h)
        depth_now = n_active <= chain_linger_streams_ ? std::max(chain_depth_.load(), chain_depth_few_.load()) : chain_depth_.load();
        if (lane_idx == kOrdinaryLane && chaining && n_active <= chain_linger_streams_ && ln.inflight > 0) {
            int n = 0;
            for (const EnginePic &p : pending_) if (p.codec == 0 && p.has_picture && p.lane(true) == lane_idx) n++;
            if (n < depth_now * n_active) return false;
        }
    }
    // Stage batches: `k_recon_inter` + `k_deblock` hold the lane for ~0.93 ms whatever the batch holds, so a batch should hold a picture of EVERY stream.  With deep
    // queues (the engine is the bottleneck) everyone is there at once.  With shallow ones a batch formed the moment something is pending holds what arrived since
    // the last one -- 16 pictures of 32 streams at 16.5 k frames/s, the lane 88 % busy with half-empty launches (profiles/r06_copy_streams.txt).  So a batch
    // waits for the streams that have nothing pending yet -- but only while nobody is in a hurry: a picture is URGENT when its handle has no decoded frame left
    // that its caller has not fetched (a caller that waits for the PCIe link has a frame or two in hand: its next picture can wait a millisecond; a caller that
    // waits for its parser or for the engine has none: its picture goes at once).  At most fill_linger_ns_ after the first picture could have gone.
    if (lane_idx == kOrdinaryLane && !chaining && fill_linger_ns_ > 0 && (int)recent_.size() > decoders_pending_) {
        const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (!ln.wait_since_ns) ln.wait_since_ns = now;
        if (now - ln.wait_since_ns < fill_linger_ns_) {
            bool urgent = false; size_t n_seen = 0; const size_t n_dec = (size_t)decoders_pending_;
            scan_tag_form_++;
            for (auto it = pending_.begin(); it != pending_.end() && n_seen < n_dec && !urgent; ++it) {
                EngineDecoderState &es = it->dec->engine_state();
                if (es.form_tag == scan_tag_form_) continue;            // only a decoder's oldest pending picture counts
                es.form_tag = scan_tag_form_; n_seen++;
                if (it->lane(false) != lane_idx || es.inflight > es.lane_inflight[lane_idx]) continue;      // (not for this batch anyway)
                urgent = it->dec->frames_done_unfetched() < 1;
            }
            if (!urgent) { lingered_ = true; return false; }
        }
    }
    if (lane_idx == kOrdinaryLane) ln.wait_since_ns = 0;
    // The intra lane alike: `k_recon_inter` + `k_intra_band` + `k_deblock` hold it for ~1.8 ms whether the batch holds one I picture or ten, and with shallow queues
    // I pictures come one at a time (32 streams, one in 30 pictures: one every 1.8 ms).  While nobody is in a hurry a batch waits for the fourth.
    // (Deep queues: the early launches above all; chain launches keep their I pictures on the ordinary lane.)
    if (lane_idx == kIntraLane && !chaining && !deep_queues_ && fill_linger_ns_ > 0) {
        const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
        if (!ln.wait_since_ns || now - ln.wait_since_ns < fill_linger_ns_) {
            bool urgent = false; int n_elig = 0; size_t n_seen = 0; const size_t n_dec = (size_t)decoders_pending_;
            scan_tag_form_++;
            for (auto it = pending_.begin(); it != pending_.end() && n_seen < n_dec && !urgent && n_elig < 4; ++it) {
                EngineDecoderState &es = it->dec->engine_state();
                if (es.form_tag == scan_tag_form_) continue;
                es.form_tag = scan_tag_form_; n_seen++;
                if (it->codec != 0 || it->lane(false) != lane_idx) continue;
                if (es.inflight > es.lane_inflight[lane_idx] && !(cross_lane_ && others_done(it->dec, lane_idx, it->seq))) continue;
                n_elig++;
                urgent = it->dec->frames_done_unfetched() < 1;
            }
            if (n_elig == 0) return false;
            if (!ln.wait_since_ns) ln.wait_since_ns = now;
            if (n_elig < 4 && !urgent && now - ln.wait_since_ns < fill_linger_ns_) { lingered_ = true; return false; }
        }
        ln.wait_since_ns = 0;
    }
    std::vector<Decoder *> seen, members;
    const bool relaxed = cross_lane_ && !chaining;
    const long long now_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    size_t n_pre = 0, n_post = 0;                       // display frames the batch packs out before / after its decode kernels
    auto account = [&](EnginePic &p, EngineDecoderState &es) {
        if (p.has_picture) es.batch_written |= 1u << (p.codec == 0 ? p.pp.cur : p.hp.cur);
        es.batch_read |= p.ref_mask | p.out_mask;
        n_pre += p.out_before.size(); n_post += p.out_after.size();
    };
    // (the walk ends once every decoder that has pictures pending has been seen: with the engine as the bottleneck some 1,100 pictures are pending at 32 streams,
    //  the decoders' oldest ones among the first hundred or two -- and this runs under m_, which every submitting thread needs, on every turn of the engine loop)
    const size_t n_dec_pending = (size_t)decoders_pending_;
    auto took = [&](EngineDecoderState &es) { if (--es.n_pending == 0) decoders_pending_--; };
    for (auto it = pending_.begin(); it != pending_.end() && (int)b.pics.size() < kMaxBatch && seen.size() < n_dec_pending;) {
        Decoder *d = it->dec;
        if (std::find(seen.begin(), seen.end(), d) != seen.end()) { ++it; continue; }
        seen.push_back(d);                              // only a decoder's OLDEST pending picture is a candidate
        EngineDecoderState &es = d->engine_state();
        // Decode order per stream: pictures of one lane follow each other on the lane's stream.  A picture whose decoder still has EARLIER pictures in
        // flight on another lane waits for them: until round 5 for the host to have retired them (the batch's last copy, the engine thread's turn, every
        // handle's completion callback); now until the device is done with them (one event query per such batch) -- a stream's I picture left it out of 30 %
        // of the ordinary batches (profiles/r06_lane_fill.txt).  Chain launches keep the strict rule: Engine::recover reasons about one lane at a time.
        int why = 0;
        if (it->lane(chaining) != lane_idx) why = 1;
        else if (es.inflight > es.lane_inflight[lane_idx] && !(relaxed && others_done(d, lane_idx, it->seq))) why = 2;
        // the pack-job tables hold 2 * kMaxBatch entries each: a picture whose display frames no longer fit waits for the next batch
        // (a flush or an IDR picture can release a whole DPB at once: up to 16 frames from one handle)
        else if (n_pre + it->out_before.size() > (size_t)2 * kMaxBatch || n_post + it->out_after.size() > (size_t)2 * kMaxBatch) why = 3;
        if (why) {
            if (lane_idx == kOrdinaryLane && it->codec == 0) { std::lock_guard<std::mutex> lk(sm_);
                (why == 1 ? st_.rej_other_lane : why == 2 ? st_.rej_cross_lane : st_.rej_tables)++;
                if (!es.blocked_since) { es.blocked_since = now_ns;
                    if (g_lane_trace > 0) {
                        LANE_TRACE("left-out dec %p seq %llu why %d inflight %d (lane0 %d lane1 %d)\n", (void *)d, it->seq, why, es.inflight, es.lane_inflight[0],
                            es.lane_inflight[1]);
                        for (int li = 0; li < kLanes; li++) for (int k = 0; k < lanes_[li].inflight; k++) { Batch &ob = lanes_[li].ring[(lanes_[li].tail + k) % kBatchRing];
                            for (auto &p : ob.pics) if (p.dec == d) LANE_TRACE("    in flight: lane %d batch %llu seq %llu kdone %d packed %d done %d\n", li, ob.serial, p.seq,
                                (int)hipEventQuery(ob.kdone), (int)hipEventQuery(ob.packed), (int)hipEventQuery(ob.done)); }
                        (void)hipGetLastError();
                    } } }
            ++it; continue;
        }
        if (lane_idx == kOrdinaryLane && es.blocked_since) { std::lock_guard<std::mutex> lk(sm_); st_.blocked_ns += now_ns - es.blocked_since; st_.blocked_n++;
            LANE_TRACE("rejoined dec %p seq %llu after %.3f ms\n", (void *)d, it->seq, (now_ns - es.blocked_since) * 1e-6);
            es.blocked_since = 0; }
        es.lane = lane_idx; es.inflight++; es.lane_inflight[lane_idx]++;
        es.in_batch = 1; es.batch_written = es.batch_read = 0; es.batch_stop = false;
        es.batch_resid = it->has_picture && it->codec == 0 && (it->pp.stages & (PS_INTRA_LDS | PS_INTRA_V1)) != 0;
        account(*it, es);
        members.push_back(d);
        took(es);
        b.pics.push_back(std::move(*it));
        it = pending_.erase(it);
    }
    // Intra-only pictures ahead of their turn (round 6).  An IDR / I picture references nothing: all that ties it to the pictures before it in its stream is
    // the surface it decodes into.  With many streams the engine is what the streams wait for -- a decoder has some twenty parsed pictures pending -- so the
    // I picture of the NEXT IDR period is here long before its turn, and its turn used to cost its stream three to four batch times (wait for its P
    // pictures to retire, run alone on the intra lane, retire, rejoin).  It is launched as soon as it is seen when no earlier picture of its decoder,
    // pending or in flight, decodes into, references or displays its surface, it packs nothing before its kernels and displays at most itself.  Frames
    // still leave in display order (the handle's queue was filled at submit time); the P picture behind it waits for it like for any picture on another lane.
    // The look costs a walk over everything pending (some twenty pictures per decoder, under m_, which every submitting thread needs): at most every
    // 250 us -- an intra batch holds the lane for ~1.8 ms -- and only when something changed.  (The first version looked on every turn of the engine loop
    // with a quadratic walk: the engine thread sat in it, the feeders queued for m_, 27 k -> 16-20 k frames/s: profiles/r06_lane_fill.txt.)
    if (lane_idx == kIntraLane && early_intra_ && deep_queues_ && !chaining && early_scanned_gen_ != pending_gen_ && now_ns - early_scan_ns_ > 250 * 1000) {
        early_scan_ns_ = now_ns;
        const unsigned long long tag = ++early_scan_tag_;
        std::vector<std::deque<EnginePic>::iterator> cand;
        for (auto it = pending_.begin(); it != pending_.end(); ++it) {
            Decoder *d = it->dec;
            EngineDecoderState &es = d->engine_state();
            if (es.scan_tag != tag) { es.scan_tag = tag; es.scan_touched = 0; es.scan_ahead = 0;
                es.scan_closed = std::find(members.begin(), members.end(), d) != members.end(); }
            const uint32_t own = it->has_picture ? 1u << (it->codec == 0 ? it->pp.cur : it->hp.cur) : 0u;
            if (!es.scan_closed && it->codec == 0 && it->has_picture && it->lane(false) == kIntraLane && it->ref_mask == 0 && it->out_before.empty() &&
                !it->wait_prev_pack && !(it->out_mask & ~own)) {
                es.scan_closed = true;                      // only a decoder's first intra picture is looked at
                uint32_t infl; inflight_masks(d, infl);
                // (its job list must have LANDED: the copy stream carries every handle's uploads -- 18 GB/s at 27 k frames/s, the I pictures of all streams
                //  within a few milliseconds of each other -- and runs some milliseconds behind at times; a picture launched the moment it is parsed makes
                //  the lane's stream wait for its upload, and with it every other picture of the batch.  In-order pictures were uploaded twenty pictures ago)
                const bool landed = !it->uploaded || hipEventQuery(it->uploaded) == hipSuccess;
                if (!landed) (void)hipGetLastError();
                // ... and only a picture that is still far from its turn: that is the case when the ENGINE is what its stream waits for (some twenty
                // pictures pending per decoder), and then an early launch saves the stream its stall.  With shallow queues (host- or PCIe-bound
                // callers) the picture would go a batch or two early at best, in more and smaller intra batches than its turn would have formed:
                // measured -2 % on the host-output headline (profiles/r06_lane_fill.txt), so those take their turn as before
                if (landed && es.scan_ahead >= early_intra_ahead_.load() && !(own & (es.scan_touched | infl))) cand.push_back(it);
            }
            es.scan_touched |= it->ref_mask | it->out_mask | own;
            es.scan_ahead++;
        }
        // (an intra batch costs the lane the same 1.8 ms whatever it holds: the lane takes one batch at a time while this is on, Engine::run, so whatever
        //  became ready while the previous one ran goes together)
        if (!cand.empty()) {
            // (taken back to front, by position: erasing from a deque invalidates iterators)
            std::vector<size_t> pos;
            for (auto &c : cand) pos.push_back((size_t)(c - pending_.begin()));
            for (size_t k = pos.size(); k-- > 0 && (int)b.pics.size() < kMaxBatch;) {
                auto it = pending_.begin() + (std::ptrdiff_t)pos[k];
                if (n_post + it->out_after.size() > (size_t)2 * kMaxBatch) continue;
                EngineDecoderState &es = it->dec->engine_state();
                es.inflight++; es.lane_inflight[lane_idx]++;
                LANE_TRACE("early dec %p seq %llu\n", (void *)it->dec, it->seq);
                n_post += it->out_after.size();
                took(es);
                b.pics.push_back(std::move(*it));
                pending_.erase(it);
                { std::lock_guard<std::mutex> lk(sm_); st_.early_intra++; }
            }
        }
        early_scanned_gen_ = pending_gen_;                  // (taking a picture bumps the generation below: the next look then scans again)
    }
    if (b.pics.empty()) return false;
    pending_gen_++;
    if (lane_idx == kOrdinaryLane) {
        { std::lock_guard<std::mutex> lk(sm_); st_.forms++; st_.form_decoders += (long long)seen.size(); st_.form_pending += (long long)(pending_.size() + b.pics.size()); }
        // Is the ENGINE what the streams wait for?  Then a decoder has many parsed pictures pending (its job slots: up to 40); callers bound by the host's entropy
        // decode or by the PCIe link keep about ten.  With hysteresis: the intra pictures' early launches and the one-batch-at-a-time intra lane are for deep
        // queues only -- with shallow ones they made more, smaller batches on both lanes (18.5 instead of 25 pictures per ordinary batch, 2.1 instead of 3.2 per
        // intra batch in the PCIe-bound headline: profiles/r06_lane_fill.txt section 5), which costs no frames there but is waste all the same.
        const size_t depth = (pending_.size() + b.pics.size()) / std::max<size_t>(1, seen.size());
        if (depth >= 16) deep_queues_ = true; else if (depth <= 8) deep_queues_ = false;
    }
    // bounds of a chain launch: its deblocking bands (2 workgroups each, resident for their whole wavefront) must stay well below the number of
    // workgroups the GPU holds (chain.hip), and its work list must fit the table
    auto chain_cost = [&](const EnginePic &p, int &bands, int &groups) { bands = (p.chain_intra ? 4 : 2) * ((p.mb_h + 15) / 16);
        groups = p.mb_h * ((p.mb_w + 7) / 8) + 2 * ((p.mb_h + 15) / 16); };
    int tot_bands = 0, tot_groups = 0; bool any_intra = false;
    for (auto &p : b.pics) if (p.has_picture && (p.chain_ok || p.chain_intra)) { int nb, ng; chain_cost(p, nb, ng); tot_bands += nb; tot_groups += ng;
        any_intra |= p.chain_intra; }
    auto band_limit = [&](bool intra) { return intra ? chain_bands_max_intra_ : chain_bands_max_; };
    if (lane_idx == kOrdinaryLane && chaining && (int)members.size() <= chain_max_streams_ && tot_bands <= band_limit(any_intra) &&
        tot_groups <= kMaxChainGroups) {
        // Depth: few streams -> long chains (a lone stream is bound by the latency of the deblocking wavefront, which chains overlap);
        // many streams -> the batch is already wide, and kMaxBatch bounds it.
        const int depth_cap = std::min(depth_now, std::max(1, kMaxBatch / (int)members.size()));
        for (int depth = 1; depth < depth_cap; depth++) {
            bool added = false;
            for (Decoder *d : members) {
                EngineDecoderState &es = d->engine_state();
                if (es.batch_stop || (int)b.pics.size() >= kMaxBatch) continue;
                auto it = std::find_if(pending_.begin(), pending_.end(), [&](const EnginePic &p) { return p.dec == d; });
                // the next picture joins only if it runs inside k_chain, packs nothing BEFORE the kernels (such frames may not be decoded yet),
                // and decodes into a surface that no earlier picture of this decoder in the batch writes, references or displays
                const bool ok = it != pending_.end() && it->lane(true) == lane_idx && it->has_picture && (it->chain_ok || it->chain_intra) &&
                                it->out_before.empty() && !it->wait_prev_pack &&
                                !((1u << it->pp.cur) & (es.batch_written | es.batch_read)) && n_post + it->out_after.size() <= (size_t)2 * kMaxBatch;
                int nb = 0, ng = 0;
                if (ok) chain_cost(*it, nb, ng);
                if (!ok || tot_bands + nb > band_limit(any_intra || it->chain_intra) || tot_groups + ng > kMaxChainGroups) { es.batch_stop = true; continue; }
                tot_bands += nb; tot_groups += ng; any_intra |= it->chain_intra; es.batch_resid |= it->chain_intra;
                es.inflight++; es.lane_inflight[lane_idx]++; es.in_batch++;
                account(*it, es);
                took(es);
                b.pics.push_back(std::move(*it));
                pending_.erase(it);
                b.any_chain = true; added = true;
            }
            if (!added) break;
            b.max_depth = depth + 1;
        }
    }
    // HEVC (both HEVC lanes): further pictures of a member decoder join the batch as long as each is INDEPENDENT of the decoder's pictures already in
    // it -- it references none of the surfaces they decode into, decodes into none they read or display, and packs nothing before the kernels.  That is
    // the shape of a random-access pyramid: after the anchor and the middle B picture, the B pictures of a level do not depend on each other.  A
    // single stream then runs two to four pictures per launch instead of one (its rate was one kernel sequence per picture: 1.25 k frames/s at
    // 1080p).  At most kHevcWorkSets pictures per decoder: each needs its own pre-SAO work surface and residual scratch (hevc_decoder.cpp).
    if (lane_idx == kHevcLane || lane_idx == kHevcIntraLane) {
        for (Decoder *d : members) {
            EngineDecoderState &es = d->engine_state();
            while (es.in_batch < kHevcWorkSets && (int)b.pics.size() < kMaxBatch) {
                auto it = std::find_if(pending_.begin(), pending_.end(), [&](const EnginePic &p) { return p.dec == d; });
                if (it == pending_.end() || !it->has_picture || it->codec != 1 || it->lane(false) != lane_idx || !it->out_before.empty() ||
                    it->wait_prev_pack) break;
                if ((it->ref_mask & es.batch_written) || ((1u << it->hp.cur) & (es.batch_written | es.batch_read)) ||
                    n_post + it->out_after.size() > (size_t)2 * kMaxBatch) break;
                es.inflight++; es.lane_inflight[lane_idx]++; es.in_batch++;
                es.batch_written |= 1u << it->hp.cur;
                account(*it, es);
                took(es);
                b.pics.push_back(std::move(*it));
                pending_.erase(it);
            }
        }
    }
    (void)ln;
    return true;
}

// one batched launch per stage on the lane's in-order stream; pack-out on the lane's second stream
void Engine::launch(Lane &ln, Batch &b) {
    const int n = (int)b.pics.size(), li = (int)(&ln - lanes_);
    b.serial = ++ln.launched; b.last_ev = -1;
    b.launched_dry = false;
    if (profile_) {       // diagnostic: did the lane run dry -- had its previous batch's kernels already ended when this one is launched?
        const Batch &pb = ln.ring[(ln.head + kBatchRing - 1) % kBatchRing];
        if (pb.serial + 1 == b.serial && pb.kdone) { const bool ended = hipEventQuery(pb.kdone) == hipSuccess; (void)hipGetLastError();
            b.launched_dry = ended;
            if (ended) { std::lock_guard<std::mutex> lk(sm_); st_.lane_dry[li]++; } }
    }
    LANE_TRACE("launch lane %d batch %llu pics %d\n", li, b.serial, n);
    int max_mbs = 0, max_mb_h = 0, max_mb_w = 0, max_w = 0, max_h = 0, stages = 0;
    bool wait_pack = false, any_hevc = false;
    HevcBatchDims hd;
    const EnginePic *last_upload[4] = {nullptr, nullptr, nullptr, nullptr};      // per copy stream (upload k went to stream k % n_copy_)
    b.n_pre = b.n_post = 0; b.pmask = 0;
    for (int k = 0; k < 5; k++) { b.alg[k] = 0; b.npics[k] = 0; }
    // pack jobs: [0, n_pre) before the decode kernels, [2*kMaxBatch, 2*kMaxBatch + n_post) after them
    for (int i = 0; i < n; i++) {
        EnginePic &p = b.pics[i];
        const bool hevc = p.codec == 1;
        if (hevc) { b.h_hpics[i] = p.hp; if (!p.has_picture) b.h_hpics[i].stages = 0; any_hevc = true; }
        b.h_pics[i] = p.pp;
        if (!p.has_picture || hevc) b.h_pics[i].stages = 0;
        else if (b.any_chain && (p.chain_ok || p.chain_intra)) {
            // this picture runs inside k_chain: its block of the control buffer, and which surfaces are decoded by EARLIER pictures of this launch
            PicParams &q = b.h_pics[i];
            q.stages = PS_CHAIN | (p.chain_intra ? PS_CHAIN_INTRA : 0); q.chain_idx = i; q.n_deps = 0;
            // the intra bands read the residuals from the picture's scratch (the stage path only asks for them when intra is dense)
            if (p.chain_intra) q.want_intra_resid = 1;
            for (int k = 0; k < kMaxSurfaces; k++) q.dep_pic[k] = -1;
            bool refs_in_batch = false;                   // does it reference a picture that ANY kernel of this batch decodes?
            for (int j = 0; j < i; j++) {
                const EnginePic &e = b.pics[j];
                if (e.dec != p.dec || !e.has_picture || e.codec != 0) continue;
                if (p.ref_mask & (1u << e.pp.cur)) refs_in_batch = true;
                if (e.chain_ok || e.chain_intra) { q.dep_pic[e.pp.cur] = (int8_t)j; if (p.ref_mask & (1u << e.pp.cur)) q.n_deps++; }
            }
            // A picture whose references were all complete before this launch (the first picture of its stream in the batch) is reconstructed
            // by the stage kernel k_recon_inter, which runs first on the lane's stream: cached reference loads at 5 waves per SIMD instead of
            // cache-bypassing ones at 3 inside k_chain.  k_chain then only deblocks it (its bands find the reconstruction complete).
            // (a reference decoded by the STAGE kernels of this batch -- a picture with intra macroblocks in front of the chain -- rules that out
            // too: k_recon_inter reconstructs all its pictures at once; inside k_chain, which runs after every stage kernel, the order is right)
            // (measured against reconstructing it inside k_chain: 1 stream 3507 / 3476, 4 streams 8810 / 8315, 8 streams 10157 / 9585 frames/s)
            if (!refs_in_batch) q.stages |= PS_RECON;
        }
        stages |= b.h_pics[i].stages;
        if (p.has_picture && !hevc) max_mb_w = std::max(max_mb_w, p.mb_w);
        // a surface this decoder's pictures of the PREVIOUS batch display may still be read by that batch's pack-out (it runs beside this batch's kernels)
        if (p.has_picture && ((1u << (hevc ? p.hp.cur : p.pp.cur)) & p.dec->engine_state().displayed[li])) wait_pack = true;
        if (hevc && p.has_picture) {
            const HevcPicParams &h = p.hp;
            hd.max_pus = std::max(hd.max_pus, h.n_pus); hd.max_tbs = std::max(hd.max_tbs, h.n_tbs); hd.max_itbs = std::max(hd.max_itbs, h.n_itbs);
            hd.max_ctb_w = std::max(hd.max_ctb_w, h.ctb_w); hd.max_ctb_h = std::max(hd.max_ctb_h, h.ctb_h);
            hd.max_w = std::max(hd.max_w, h.w); hd.max_h = std::max(hd.max_h, h.h); hd.any_intra |= (h.stages & HPS_INTRA) != 0;
            hd.any_deblock |= (h.stages & HPS_DEBLOCK) != 0; hd.any_sao |= (h.stages & HPS_SAO) != 0;
        }
        if (p.has_picture) {
            max_mbs = std::max(max_mbs, p.mb_w * p.mb_h); max_mb_h = std::max(max_mb_h, p.mb_h);
            if (p.uploaded) { const EnginePic *&lu = last_upload[p.upload_seq % (unsigned)n_copy_]; if (!lu || p.upload_seq > lu->upload_seq) lu = &p; }
        }
        if (p.wait_prev_pack) wait_pack = true;
        for (auto &j : p.out_before) b.h_jobs[b.n_pre++] = j;                           // form() keeps both tables within 2 * kMaxBatch
        for (auto &j : p.out_after) b.h_jobs[2 * kMaxBatch + b.n_post++] = j;
        if (!p.out_before.empty() || !p.out_after.empty()) { max_w = std::max(max_w, p.disp_w); max_h = std::max(max_h, p.disp_h); }
        int st = b.h_pics[i].stages;
        if (hevc && p.has_picture) { const int hs = p.hp.stages; if (hs & (HPS_MC | HPS_RESID)) { b.alg[0] += p.alg_bytes[0]; b.npics[0]++; }
            if (hs & HPS_INTRA) { b.alg[1] += p.alg_bytes[1]; b.npics[1]++; } if (hs & (HPS_DEBLOCK | HPS_SAO)) { b.alg[2] += p.alg_bytes[2]; b.npics[2]++; } }
        if (st & PS_RECON) { b.alg[0] += p.alg_bytes[0]; b.npics[0]++; }
        if (st & (PS_INTRA_LDS | PS_INTRA_V1)) { b.alg[1] += p.alg_bytes[1]; b.npics[1]++; }
        if (st & (PS_DEBLOCK_LDS | PS_DEBLOCK_V1)) { b.alg[2] += p.alg_bytes[2]; b.npics[2]++; }
        if (st & PS_CHAIN) { b.alg[4] += ((st & PS_RECON) ? 0 : p.alg_bytes[0]) + ((st & PS_CHAIN_INTRA) ? p.alg_bytes[1] : 0) + p.alg_bytes[2]; b.npics[4]++; }
        b.alg[3] += p.alg_bytes[3] * (long long)(p.out_before.size() + p.out_after.size());
        b.npics[3] += (int)(p.out_before.size() + p.out_after.size());
    }
    // surfaces this batch displays, per decoder: a later batch that decodes into one of them must wait for this batch's pack-out
    // (per lane: pictures of a decoder on different lanes are kept apart by Engine::form)
    for (auto &p : b.pics) p.dec->engine_state().displayed[li] = 0;
    for (auto &p : b.pics) p.dec->engine_state().displayed[li] |= p.out_mask;
    b.max_mbs = max_mbs; b.max_mb_h = max_mb_h; b.max_w = max_w; b.max_h = max_h; b.redo = false;
    hipStream_t st = ln.stream, pst = ln.pack_stream;
    // What does not depend on the lane's previous batch -- clearing the control blocks, the parameter / pack-job tables, the deblocking pre-pass (it only
    // reads the job lists) -- is issued on the lane's pre-stream, so it runs WHILE the previous batch's kernels are still busy instead of in the gap behind
    // them.
    // (This batch's tables were last used four batches ago: the ring guarantees that batch has retired.)
    hipStream_t ps = any_hevc ? st : ln.pre_stream;
    // every counter of every picture, and the abort word
    if (!any_hevc && (stages & (PS_INTRA_LDS | PS_DEBLOCK_LDS | PS_CHAIN))) { hipMemsetAsync(b.d_ctl, 0, sizeof(int) * (size_t)n * chain_ctl_ints(), ps);
        hipMemsetAsync(b.d_ctl + (size_t)kMaxBatch * chain_ctl_ints(), 0, chain_tail_ints() * sizeof(int), ps);        // abort word, census, time stamps
        // the launch's wait limit (chain_common.h): ten times the longest healthy wait, which is the launch itself -- about 0.25 ms per picture of a stream's
        // chain at 1080p (1.3-2.1 ms for 8 pictures, profiles/r04_chain_timeline.txt), in proportion to the picture size; never below 10 ms, never above 100
        if (stages & PS_CHAIN) {
            const double launch_ms = 0.25 * b.max_depth * std::max(1.0, max_mbs / 8160.0) + 0.8;
            const unsigned ticks = (unsigned)(std::min(100.0, std::max(10.0, 10.0 * launch_ms)) * 100000.0);
            static const unsigned forced = getenv("JM_AMD_DEC_CHAIN_WAIT_MS") ? (unsigned)(atof(getenv("JM_AMD_DEC_CHAIN_WAIT_MS")) * 100000.0) : 0u;
            hipMemsetD32Async((hipDeviceptr_t)(b.d_ctl + (size_t)kMaxBatch * chain_ctl_ints() + chain_tail_wait_limit()), (int)(forced ? forced : ticks), 1, ps);
        } }
    if (any_hevc) hipMemcpyAsync(b.d_hpics, b.h_hpics, sizeof(HevcPicParams) * n, hipMemcpyHostToDevice, ps);
    else hipMemcpyAsync(b.d_pics, b.h_pics, sizeof(PicParams) * n, hipMemcpyHostToDevice, ps);
    if (b.n_pre) hipMemcpyAsync(b.d_jobs, b.h_jobs, sizeof(PackJob) * b.n_pre, hipMemcpyHostToDevice, ps);
    if (b.n_post) hipMemcpyAsync(b.d_jobs + 2 * kMaxBatch, b.h_jobs + 2 * kMaxBatch, sizeof(PackJob) * b.n_post, hipMemcpyHostToDevice, ps);
    // job lists were copied on the (in-order) copy stream when the pictures were parsed: waiting for the most recently
    // issued one of this batch covers them all without waiting for uploads of later pictures
    for (const EnginePic *lu : last_upload) if (lu) hipStreamWaitEvent(ps, lu->uploaded, 0);
    if (profile_ && !any_hevc) hipEventRecord(b.pev[8], ps);
    bool prep_early = false;
    if (!any_hevc) {
        if (stages & (PS_DEBLOCK_LDS | PS_CHAIN)) { launch_deblock_prep(b.d_pics, n, max_mbs, ps); prep_early = true; }
        if (profile_) hipEventRecord(b.pev[9], ps);
        hipEventRecord(b.pre_done, ps);
        hipStreamWaitEvent(st, b.pre_done, 0);
    }
    // Pack-out of batch k runs on its own stream and overlaps the decode kernels of batch k+1 (PCIe writes vs. compute).
    // The decoder never reuses a displayed surface for the very next picture (DPB cooling, decoder.cpp), so the decode
    // kernels of this batch only have to wait for the pack-out launched TWO batches ago.
    if (ln.pack_hist[1]) hipStreamWaitEvent(st, ln.pack_hist[1], 0);
    if ((wait_pack || b.n_pre) && ln.pack_hist[0]) hipStreamWaitEvent(st, ln.pack_hist[0], 0);
    auto mark = [&](int i, hipStream_t s) { if (profile_) hipEventRecord(b.pev[i], s); };
    mark(0, st);
    // Pack-out: k_packout writes the tight frames into device staging and a copy engine (SDMA) moves them to the pinned slots.
    // Letting the kernel store into host memory directly saves that hop but its PCIe-bound stores share the L2 / fabric write
    // queues with everything else: k_recon_inter of the next batch ran 4x slower next to it (0.56 -> 2.3 ms for 32 pictures).
    auto copy_out = [&](const std::vector<OutSlot *> &slots, hipStream_t s) { for (OutSlot *o : slots) if (o->dev && o->host &&
        !o->fetch) hipMemcpyAsync(o->host, o->dev, o->bytes, hipMemcpyDeviceToHost, s); };
    if (b.n_pre) { launch_packout(b.d_jobs, b.n_pre, max_w, max_h, st); b.pmask |= 1; for (auto &p : b.pics) copy_out(p.slots_before, st); }
    mark(1, st);
    if (any_hevc && (hd.max_pus || hd.max_tbs || hd.any_intra || hd.any_deblock || hd.any_sao)) {
        // HEVC batch (its own lane, so never mixed with H.264 pictures): MC + residual | intra diagonals | deblocking + SAO
        hipEvent_t ev[4] = {b.pev[1], b.pev[2], b.pev[3], b.pev[4]};
        launch_hevc_picture_batch(b.d_hpics, n, hd, b.d_progress, st, profile_ ? ev : nullptr);
        if (hd.max_pus || hd.max_tbs) b.pmask |= 2;
        if (hd.any_intra) b.pmask |= 4;
        if (hd.any_deblock || hd.any_sao) b.pmask |= 8;
        b.last_ev = 4;
    }
    b.any_bipred = b.any_field = false;
    for (auto &p : b.pics) { b.any_bipred |= p.has_picture && p.codec == 0 && p.bipred; b.any_field |= p.has_picture && p.codec == 0 && p.pp.field != 0; }
    if (stages & PS_RECON) { launch_recon_inter(b.d_pics, n, max_mbs, b.any_bipred && !debug_no_bi_, b.any_field, b.d_err, st); b.pmask |= 2; }
    if (!any_hevc) mark(2, st);
    if (stages & PS_INTRA_LDS) { launch_intra_lds(b.d_pics, n, max_mb_h, b.d_ctl, b.d_err, st); b.pmask |= 4; }
    if (stages & PS_INTRA_V1) { launch_recon_intra(b.d_pics, n, st); b.pmask |= 4; }
    if (!any_hevc) mark(3, st);
    if ((stages & (PS_DEBLOCK_LDS | PS_CHAIN)) && !prep_early) launch_deblock_prep(b.d_pics, n, max_mbs, st);
    if (stages & PS_DEBLOCK_LDS) { launch_deblock_lds(b.d_pics, n, max_mb_h, b.d_ctl, b.d_err, debug_stall_ == 1, st); b.pmask |= 8; }
    if (stages & PS_DEBLOCK_V1) { launch_deblock(b.d_pics, n, st); b.pmask |= 8; }
    if (!any_hevc) { mark(4, st); b.last_ev = 4; }
    // pictures that run inside the chain kernel: reconstruction + deblocking of all of them, consecutive pictures of a stream pipelined
    if (stages & PS_CHAIN) {
        // Work list of the chain kernel, ordered along the pipeline's time axis (chain.hip).  The deblocking wavefront of a picture reaches
        // macroblock (x, y) in step x + L * y (L = deblock_row_lag(), 1 since round 4); the band that holds row y trails the band above it by about
        // kBandLag steps (prefetch depth + publishing lag of the ring-row hand-over, deblock_device.h); a picture follows the previous picture of
        // its stream `lag` steps behind, plus what its vectors reach beyond the usual window: L steps per macroblock row further DOWN and one step per
        // macroblock further RIGHT (both from the parser; rounds 2-3 counted only the rows, so a stream with long rightward vectors could break the
        // rule below).  Key of the 8-macroblock segment c of row r:
        //     base(picture) + L * r + 8 * c + kBandLag * (r / band_rows)
        // i.e. (an upper bound of) the step in which the picture's OWN deblocking wants the segment.  Rule: every dependency of a reconstruction
        // group points to a smaller key.  Direct ones: the reference samples of macroblock (x, r) are final after step (x + 1 + reach_x) + L * (r +
        // 1 + reach_y) of the previous picture, published 2 steps late.  Transitive ones: to run that step the band has prefetched 3 steps ahead,
        // i.e. it has waited for the reconstruction bits of every macroblock (x', y') of its rows with x' + L * y' up to that step + 3 -- and the
        // band ABOVE it has run kBandLag steps further, the one above that 2 * kBandLag, ...: exactly the term the key carries, so the bound holds
        // at every band level (rounds 2-3 had no such term: with nine bands at 4K the bound was off by up to 8 * 7 steps for the lowest band).
        // With 8c <= x' the previous picture's groups involved have keys <= base' + x + L * r + L + 8 + kBandLag * band, this group's key is
        // >= base' + lag + x - 7 + L * r + kBandLag * band: smaller whenever lag > L + 16 (one step more since the final store of a macroblock moved
        // behind the next one's vertical edges).  That is what keeps a full machine from deadlocking:
        // the unfinished group with the smallest key is resident (in-order dispatch per XCD) and waits only for finished groups and resident bands.
        // The SLOPE of a picture's keys is the pace of whatever consumes its reconstruction: one step per macroblock row for a picture that is only deblocked
        // (deblock_row_lag()), two for a picture with the intra role (its intra wavefront needs the macroblock above right COMPLETE: x + 2y). Rounds 2-3 and
        // the
        // first form of round 4 gave a whole launch ONE slope (2 as soon as it held an intra-role picture): with one-row deblocking that puts a band's needs up
        // to 14 keys beyond a dependent group's own key -- the band moves as a unit over 16 rows -- and chain launches of 4 / 8 streams gave up in 3 runs of 10
        // (profiles/r04_ab6_first_giveup.json).  tools/chain_keys.py checks the rule by brute force for every macroblock of a picture (it found both numbers):
        //   deblock-only picture -> deblock-only successor: largest needed key = key - 1 (holds; + kKeySlack for comfort);
        //   intra-role picture (slope 2, its one-row deblocking bands gated by a two-row intra wavefront) -> any successor: the successor starts
        //   mb_h + kIntraExtra keys later (needed: > mb_h + 7 at 1080p, > mb_h + 4 at 4K).  One such picture per IDR period: nothing measurable.
        const int band_rows = chain_band_rows(), L = deblock_row_lag();
        constexpr int kBandLag = 8, kKeySlack = 2, kIntraExtra = 24;
        std::vector<int> base_of(n, 0), slope_of(n, L);
        size_t n_keys = 0;
        for (int i = 0; i < n; i++) {
            if (!(b.h_pics[i].stages & PS_CHAIN)) continue;
            slope_of[i] = (b.h_pics[i].stages & PS_CHAIN_INTRA) ? 2 : L;
            for (int j = i - 1; j >= 0; j--) if (b.pics[j].dec == b.pics[i].dec && (b.h_pics[j].stages & PS_CHAIN)) {
                base_of[i] = base_of[j] + chain_lag_steps_ + kKeySlack + slope_of[i] * b.pics[i].reach_rows + b.pics[i].reach_cols +
                             kBandLag * ((b.pics[i].reach_rows + 1) / band_rows);
                if (b.h_pics[j].stages & PS_CHAIN_INTRA) base_of[i] += b.h_pics[j].mb_h + kIntraExtra;
                break; }
            n_keys = std::max(n_keys,
                (size_t)(base_of[i] + slope_of[i] * b.h_pics[i].mb_h + b.h_pics[i].mb_w + kBandLag * (b.h_pics[i].mb_h / band_rows + 1) + 2));
        }
        if (group_buckets_.size() < n_keys) group_buckets_.resize(n_keys);
        for (size_t k = 0; k < n_keys; k++) group_buckets_[k].clear();
        for (int i = 0; i < n; i++) {
            if (!(b.h_pics[i].stages & PS_CHAIN) || (b.h_pics[i].stages & PS_RECON)) continue;      // (PS_RECON: reconstructed by the stage kernel)
            const int mb_h = b.h_pics[i].mb_h, segs = (b.h_pics[i].mb_w + 7) / 8, base = base_of[i];
            for (int r = 0; r < mb_h; r++) {
                const int kr = base + slope_of[i] * r + kBandLag * (r / band_rows);
                for (int c = 0; c < segs; c++) group_buckets_[kr + 8 * c].push_back((uint32_t)i << 16 | (uint32_t)(r * 32 + c));
            }
        }
        // Every deblocking band of the launch goes FIRST (Engine::form keeps their number at half of what the GPU holds): a band lives for its whole
        // wavefront and waits for reconstruction bits with larger keys, so it must be resident before anything that waits for it is started --
        // then every reconstruction group only ever waits for workgroups that are resident or done, and the one with the smallest key can always run.
        int n_groups = 0;
        bool with_intra = false;
        for (int i = 0; i < n; i++) if (b.h_pics[i].stages & PS_CHAIN) for (int bnd = 0; bnd * band_rows < b.h_pics[i].mb_h; bnd++) {
            // band of the intra wavefront
            if (b.h_pics[i].stages & PS_CHAIN_INTRA) { b.h_groups[n_groups++] = (uint32_t)i << 16 | 0xC000u | (uint32_t)bnd; with_intra = true; }
            b.h_groups[n_groups++] = (uint32_t)i << 16 | 0x8000u | (uint32_t)bnd;
        }
        // (The invariant of chain_common.h is established by construction here -- Engine::form keeps the bands within their budget, base_of is built from
        // kMinChainLag / kKeySlack above -- and checked independently by tools/chain_keys.py, which restates every device wait and tests the keys by brute
        // force, tests/test_chain_keys.py.  A run-time re-check against the same constants, as round 4 had here, could never fire: ADVICE r4.)
        // Groups of ONE key are independent of each other: inside a bucket they are dealt so that a picture's 8-macroblock column always runs on the same XCD
        // pair (chain_order.h; the key rule orders buckets only and does not see it)
        for (size_t k = 0; k < n_keys; k++) append_bucket_by_xcd(group_buckets_[k].data(), group_buckets_[k].size(), b.h_groups, n_groups, bucket_tmp_);
        hipMemcpyAsync(b.d_groups, b.h_groups, sizeof(uint32_t) * (size_t)n_groups, hipMemcpyHostToDevice, st);
        launch_chain(b.d_pics, b.d_groups, n_groups, with_intra, b.d_ctl, b.d_err, debug_stall_ != 0, st);
        b.chain_with_intra = with_intra;
        b.pmask |= 32; mark(7, st); b.last_ev = 7;
    }
    hipEventRecord(b.kdone, st);
    hipStreamWaitEvent(pst, b.kdone, 0);
    mark(5, pst);
    if (b.n_post) { launch_packout(b.d_jobs + 2 * kMaxBatch, b.n_post, max_w, max_h, pst); b.pmask |= 16; }
    mark(6, pst);
    hipEventRecord(b.packed, pst);                            // from here on the displayed surfaces may be decoded into again
    for (auto &p : b.pics) copy_out(p.slots_after, pst);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) fprintf(stderr, "jm_amd_dec: kernel launch failed: %s\n", hipGetErrorString(le));
    hipEventRecord(b.done, pst);
    ln.pack_hist[1] = ln.pack_hist[0]; ln.pack_hist[0] = b.packed;
}

// A wait inside a chain launch gave up: the launch assumed slots that were not there (another process on the GPU, a long kernel of another stream).  Its
// pictures -- and those of the lane's next batch, which read them -- are decoded again by the stage kernels, one picture per stream at a time; chain
// launches then pause for a while.  Synchronous and slow on purpose: it should never happen on a GPU the engine owns.
//
// What the redo may NOT do is claim a clean result it cannot deliver.  The lane's next batch has already run when this batch is recovered, and
// surfaces are reused round robin: if that batch decoded into a surface this batch's pictures reference from OUTSIDE the batch (an older picture), or
// into one this batch displays, the redo reads / packs the newer picture.  Such a decoder is "tainted": its pictures of this batch and of the next one
// keep an error (stat errors, jm_amddec_last_error) instead of being passed off as recovered.  Frames the next batch packed BEFORE its kernels
// (out_before: they show pictures of this batch) are packed again when it is redone, or tainted when their surfaces are gone.
void Engine::recover(Lane &ln, Batch &b, const std::vector<std::pair<Decoder *, uint32_t>> &later) {
    const int n = (int)b.pics.size();
    hipStream_t st = ln.stream;
    hipStreamSynchronize(ln.pack_stream); hipStreamSynchronize(st);
    auto is_tainted = [&](Decoder *d) { return std::find(ln.tainted.begin(), ln.tainted.end(), d) != ln.tainted.end(); };
    auto taint = [&](Decoder *d) { if (!is_tainted(d)) ln.tainted.push_back(d); };
    // 1. which decoders can be redone from intact data
    for (int i = 0; i < n; i++) {
        Decoder *d = b.pics[i].dec;
        bool first = true;
        for (int j = 0; j < i; j++) first &= b.pics[j].dec != d;
        if (!first) continue;
        uint32_t written = 0, ext_refs = 0, shown = 0, shown_before = 0, lw = 0;
        for (int j = i; j < n; j++) {
            const EnginePic &p = b.pics[j];
            if (p.dec != d) continue;
            ext_refs |= p.ref_mask & ~written;                 // references decoded before this batch
            shown |= p.out_mask;
            if (!p.out_before.empty()) shown_before |= p.out_mask;
            if (p.has_picture && p.codec == 0) written |= 1u << p.pp.cur;
        }
        for (auto &e : later) if (e.first == d) lw |= e.second;
        // b.redo: this batch is the "next batch" of a recovered one -- its own first run may have decoded into surfaces its out_before frames show
        if ((lw & (ext_refs | shown)) || (b.redo && (written & shown_before))) taint(d);
    }
    // 2. frames this batch packed before its kernels showed pictures of the recovered batch: again, from the pictures as they are now
    if (b.redo && b.n_pre) {
        launch_packout(b.d_jobs, b.n_pre, b.max_w, b.max_h, st);
        for (auto &p : b.pics) for (OutSlot *o : p.slots_before) if (o->dev && o->host && !o->fetch) hipMemcpyAsync(o->host, o->dev, o->bytes,
            hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
    }
    // 3. the pictures again, one per stream at a time
    std::vector<int> depth(n, 0); int max_depth = 0;
    for (int i = 0; i < n; i++) { for (int j = 0; j < i; j++) if (b.pics[j].dec == b.pics[i].dec && b.pics[j].has_picture) depth[i]++;
        max_depth = std::max(max_depth, depth[i]); }
    for (int d = 0; d <= max_depth; d++) {
        int stages = 0;
        for (int i = 0; i < n; i++) {
            const EnginePic &p = b.pics[i];
            b.h_pics[i] = p.pp;
            b.h_pics[i].stages = (p.has_picture && p.codec == 0 && depth[i] == d) ? p.classic_stages : 0;
            stages |= b.h_pics[i].stages;
            b.h_err[i] = 0;
        }
        if (!stages) continue;
        hipMemsetAsync(b.d_ctl, 0, sizeof(int) * (size_t)n * chain_ctl_ints(), st);
        hipMemcpyAsync(b.d_pics, b.h_pics, sizeof(PicParams) * n, hipMemcpyHostToDevice, st);
        if (stages & PS_RECON) launch_recon_inter(b.d_pics, n, b.max_mbs, b.any_bipred, b.any_field, b.d_err, st);
        if (stages & PS_INTRA_LDS) launch_intra_lds(b.d_pics, n, b.max_mb_h, b.d_ctl, b.d_err, st);
        if (stages & PS_INTRA_V1) launch_recon_intra(b.d_pics, n, st);
        if (stages & PS_DEBLOCK_LDS) { launch_deblock_prep(b.d_pics, n, b.max_mbs, st);
            launch_deblock_lds(b.d_pics, n, b.max_mb_h, b.d_ctl, b.d_err, false, st); }
        if (stages & PS_DEBLOCK_V1) launch_deblock(b.d_pics, n, st);
        hipStreamSynchronize(st);                              // h_pics is rewritten for the next depth
    }
    if (b.n_post) {                                            // the display frames of the batch again, from the pictures as they are now
        launch_packout(b.d_jobs + 2 * kMaxBatch, b.n_post, b.max_w, b.max_h, st);
        for (auto &p : b.pics) for (OutSlot *o : p.slots_after) if (o->dev && o->host && !o->fetch) hipMemcpyAsync(o->host, o->dev, o->bytes,
            hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
    }
    // 4. what could not be redone from intact data stays an error of its handle (complete() reports the words that are set)
    for (int i = 0; i < n; i++) if (is_tainted(b.pics[i].dec) && !b.h_err[i]) b.h_err[i] = kErrNotRecovered;
    { std::lock_guard<std::mutex> lk(sm_); st_.chain_recoveries++; }
    const auto since_epoch = std::chrono::steady_clock::now().time_since_epoch();
    chain_block_until_ns_ = std::chrono::duration_cast<std::chrono::nanoseconds>(since_epoch).count() + 30ll * 1000 * 1000 * 1000;
    static bool said = false;
    if (!said) { said = true;
        fprintf(stderr, "jm_amd_dec: device %d: a chain launch ran out of time waiting for workgroups (is the GPU shared?) -- its pictures were decoded "
                "again by the stage kernels; chain launches pause for 30 s\n", device_);
    }
}

// JM_AMD_DEC_VERBOSE: what a chain launch that gave up looked like when it ended -- the census of its workgroups, and per picture
// the step counters of every band and how far the reconstruction bitmap got.  Read after the launch has retired; `redo` batches reuse the buffer.
void Engine::dump_chain_state(Batch &b) {
    const int n = (int)b.pics.size(), stride = chain_ctl_ints();
    std::vector<int> ctl((size_t)kMaxBatch * stride + chain_tail_ints());
    if (hipMemcpy(ctl.data(), b.d_ctl, ctl.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
    const int *lw = ctl.data() + (size_t)kMaxBatch * stride;
    int n_band_wgs = 0, n_recon_groups = 0;
    for (int i = 0; i < n; i++) if (b.h_pics[i].stages & PS_CHAIN) { const int bands = (b.h_pics[i].mb_h + 15) / 16;
        n_band_wgs += ((b.h_pics[i].stages & PS_CHAIN_INTRA) ? 4 : 2) * bands;
        if (!(b.h_pics[i].stages & PS_RECON)) n_recon_groups += b.h_pics[i].mb_h * ((b.h_pics[i].mb_w + 7) / 8); }
    fprintf(stderr, "  census: abort %d | reconstruction workgroups started %d done %d (work list: %d groups x 2, some empty) | "
        "band workgroups started %d done %d of %d | highest work-list index started %d | band budget %d / %d\n",
        lw[0], lw[1], lw[2], n_recon_groups, lw[3], lw[4], n_band_wgs, lw[5], chain_bands_max_, chain_bands_max_intra_);
    // (sums of ticks: read as unsigned, a large launch passes 2^31)
    if (lw[2] > 0) fprintf(stderr, "  reconstruction workgroups: %.2f us each on average, of which %.2f us in wait_final (wave 0)\n",
        0.01 * (double)(unsigned)lw[7] / lw[2], 0.01 * (double)(unsigned)lw[6] / lw[2]);
    if (lw[8]) fprintf(stderr,
        "  FIRST give-up (of %d): code %d (1 fin: reconstruction waits for the deblocking of picture `pic`; 2 bits: a deblocking band of `pic` waits "
        "for its reconstruction; 4 ring; 8 intra ring; 16 ifin) pic %d where 0x%x (fin: band << 16 | macroblock column; bits: row << 16 | column, bit 31 of "
        "the row = intra band; ring: band << 16 | chroma) needed %d saw %d / %d\n", lw[8], lw[9], lw[10], (unsigned)lw[11], lw[12], lw[13], lw[14]);
    // the second half of the record (round 6): what decides between "the producer stood still" and "its publication was not seen", and where the waiter ran
    if (lw[8]) fprintf(stderr,
        "  ... the same counter read by an atomic read-modify-write when the wait gave up: %d (equal to what the polls saw = the producer stood there; larger = "
        "published and not seen) | waiter: XCC %d, HW_ID1 0x%08x (wave %d SIMD %d CU %d SA %d SE %d), %d looks | its timer skipped %d clock gap(s), the longest "
        "%.2f ms | wait limit of the launch %.1f ms\n", lw[16], lw[17] & 15, (unsigned)lw[18], lw[18] & 31, (lw[18] >> 8) & 3, (lw[18] >> 10) & 15,
        (lw[18] >> 16) & 1, (lw[18] >> 18) & 7, lw[19], lw[20], lw[21] * 1e-5, lw[15] * 1e-5);
    for (int i = 0; i < n; i++) {
        const PicParams &q = b.h_pics[i];
        if (!(q.stages & PS_CHAIN)) continue;
        const int *c = ctl.data() + (size_t)i * stride;
        const int bands = (q.mb_h + 15) / 16;
        fprintf(stderr, "  picture %2d dec %p %dx%d stages 0x%x%s deps %d err %d reach %d/%d |", i, (void *)b.pics[i].dec, q.mb_w, q.mb_h, q.stages,
            (q.stages & PS_RECON) ? " (reconstructed before the launch)" : "", q.n_deps, b.h_err[i], b.pics[i].reach_rows, b.pics[i].reach_cols);
        for (int k = 0; k < bands; k++) fprintf(stderr, " b%d ring %d/%d fin %d/%d", k, c[0 + k], c[32 + k], c[64 + k], c[64 + 32 + k]);
        if (q.stages & PS_CHAIN_INTRA) for (int k = 0; k < bands; k++) fprintf(stderr, " i%d ring %d/%d ifin %d/%d", k, c[128 + k], c[128 + 32 + k], c[192 + k],
            c[192 + 32 + k]);
        int full = 0, first_open = -1, first_open_bits = 0;
        for (int r = 0; r < q.mb_h; r++) { int cnt = 0; for (int w = 0; w < 8; w++) cnt += __builtin_popcount((unsigned)c[256 + r * 8 + w]);
            if (cnt >= q.mb_w) full++; else if (first_open < 0) { first_open = r; first_open_bits = cnt; } }
        fprintf(stderr, " | bitmap: %d of %d rows complete, first open row %d has %d of %d\n", full, q.mb_h, first_open, first_open_bits, q.mb_w);
    }
    // time line (diagnostic launches, JM_AMD_DEC_CENSUS): per picture, microseconds after the launch's first stamp
    const int *ts = lw + chain_tail_head_ints();
    int t0 = 0x7fffffff;
    for (int i = 0; i < n; i++) for (int k = 0; k < 4; k += 2) if (ts[4 * i + k]) t0 = std::min(t0, 0x40000000 - ts[4 * i + k]);
    if (t0 != 0x7fffffff) for (int i = 0; i < n; i++) {
        if (!(b.h_pics[i].stages & PS_CHAIN)) continue;
        auto us = [&](int k) { const int v = ts[4 * i + k]; return v ? ((k & 1) ? v - t0 : 0x40000000 - v - t0) * 0.01 : -1.0; };
        fprintf(stderr, "  time line picture %2d dec %p: reconstruction %.1f .. %.1f us, bands %.1f .. %.1f us\n", i, (void *)b.pics[i].dec, us(0), us(1),
            us(2), us(3));
    }
}

void Engine::complete(Lane &ln, Batch &b, bool failed) {
    LANE_TRACE("retire lane %d batch %llu pics %zu\n", (int)(&ln - lanes_), b.serial, b.pics.size());
    if (!failed && (b.redo || b.any_chain)) {
        bool wait_err = b.redo;
        for (size_t i = 0; i < b.pics.size(); i++) wait_err |= b.h_err[i] != 0 && b.any_chain;
        if (wait_err) {
            // the full state of the first give-ups of a process goes to stderr whatever the verbosity (a give-up nobody can explain
            // afterwards is worth nothing); JM_AMD_DEC_VERBOSE prints all of them
            static std::atomic<int> dumped{0};
            if (getenv("JM_AMD_DEC_VERBOSE") || (!b.redo && dumped.fetch_add(1) < 4)) {
                fprintf(stderr, "jm_amd_dec: chain launch of %zu pictures gave up%s; codes:", b.pics.size(), b.redo ? " (redo of the batch behind one that did)" : "");
                for (size_t i = 0; i < b.pics.size(); i++) fprintf(stderr, " %d", b.h_err[i]); fprintf(stderr, "\n");
                dump_chain_state(b); }
            // the lane's next batch (already launched) decoded from this batch's damaged pictures: let it finish, it is redone when it retires
            std::vector<std::pair<Decoder *, uint32_t>> later;     // ... and remember which surfaces it decoded into meanwhile (Engine::recover)
            if (!b.redo) ln.tainted.clear();                       // (a redo batch inherits the tainted set of the batch it followed)
            if (ln.inflight > 1) {
                Batch &nx = ln.ring[(ln.tail + 1) % kBatchRing]; hipEventSynchronize(nx.done); nx.redo = true;
                for (auto &p : nx.pics) if (p.has_picture && p.codec == 0) later.emplace_back(p.dec, 1u << p.pp.cur);
            }
            recover(ln, b, later);
        }
    }
    if (profile_ && !failed) {
        std::lock_guard<std::mutex> lk(sm_);
        auto add = [&](int cls, int e0, int e1, bool ran) {
            if (!ran) return;
            float ms = 0;
            if (hipEventElapsedTime(&ms, b.pev[e0], b.pev[e1]) == hipSuccess) { st_.ns[cls] += ms * 1e6; st_.launches[cls]++; }
        };
        add(3, 0, 1, b.pmask & 1); add(0, 1, 2, b.pmask & 2); add(1, 2, 3, b.pmask & 4); add(2, 3, 4, b.pmask & 8); add(3, 5, 6, b.pmask & 16);
        add(4, 4, 7, b.pmask & 32);
        for (int k = 0; k < 5; k++) { st_.pics[k] += b.npics[k]; st_.alg_bytes[k] += b.alg[k]; }
        st_.batches++; st_.batch_pics += (long long)b.pics.size();
        // the lane's time line: how long this batch's kernels held the lane's stream, and how long the stream sat idle since the previous batch's last kernel
        const int li = (int)(&ln - lanes_);
        st_.lane_batches[li]++; st_.lane_pics[li] += (long long)b.pics.size();
        if (b.last_ev >= 0) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, b.pev[0], b.pev[b.last_ev]) == hipSuccess) st_.lane_busy_ns[li] += ms * 1e6;
            const Batch &pb = ln.ring[(ln.tail + kBatchRing - 1) % kBatchRing];       // (its events are recorded again three launches from now at the earliest)
            if (pb.serial + 1 == b.serial && pb.last_ev >= 0 && hipEventElapsedTime(&ms, pb.pev[pb.last_ev], b.pev[0]) == hipSuccess && ms > 0) {
                st_.lane_gap_ns[li] += ms * 1e6;
                float mu = 0, mp = 0;
                const bool h264 = !b.pics.empty() && b.pics[0].codec == 0;       // (the two pre-stream events are recorded for H.264 batches)
                // (a batch launched onto a lane that had already run dry waited for nothing: the idle time before it is the lane's, not its uploads')
                if (h264 && !b.launched_dry && hipEventElapsedTime(&mu, pb.pev[pb.last_ev], b.pev[8]) == hipSuccess && mu > 0) st_.lane_upwait_ns[li] += std::min(mu, ms) * 1e6;
                if (h264 && !b.launched_dry && hipEventElapsedTime(&mp, pb.pev[pb.last_ev], b.pev[9]) == hipSuccess && mp > 0) st_.lane_prewait_ns[li] += std::min(mp, ms) * 1e6;
            }
        }
        (void)hipGetLastError();
    }
    // (counted with or without profiling)
    if (b.any_chain && !failed) { static const bool timeline = getenv("JM_AMD_DEC_CHAIN_TIMELINE") != nullptr;
        if (timeline) { fprintf(stderr, "jm_amd_dec: chain launch of %zu pictures\n", b.pics.size()); dump_chain_state(b); } }
    // clock gaps the launch's waits saw (chain_common.h WaitClock): time its waves were not run -- evidence, counted per launch
    if (b.h_err[kMaxBatch]) { std::lock_guard<std::mutex> lk(sm_); st_.wait_gap_launches++; st_.wait_gap_max_ticks = std::max(st_.wait_gap_max_ticks,
        (long long)(unsigned)b.h_err[kMaxBatch + 1]);
        if (getenv("JM_AMD_DEC_VERBOSE")) fprintf(stderr, "jm_amd_dec: a chain launch's waits saw %d clock gap(s), the longest %.2f ms: its waves were not run meanwhile "
            "(chain launch %lld of this process, batch %llu of its lane, %.1f ms after the engine's first trace point, CLOCK_MONOTONIC %.3f s when it retired)\n",
            b.h_err[kMaxBatch], (unsigned)b.h_err[kMaxBatch + 1] * 1e-5, st_.chain_batches + 1, b.serial, trace_ms(),
            std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
        b.h_err[kMaxBatch] = b.h_err[kMaxBatch + 1] = 0; }
    // diagnostic launches (JM_AMD_DEC_CENSUS): how many reconstruction workgroups shared one reference window (tests: the quad path really ran)
    if (b.any_chain && !failed) { static const bool census = getenv("JM_AMD_DEC_CENSUS") != nullptr;
        if (census) { int tail[32];
            if (hipMemcpy(tail, b.d_ctl + (size_t)kMaxBatch * chain_ctl_ints(), sizeof tail, hipMemcpyDeviceToHost) == hipSuccess) {
                std::lock_guard<std::mutex> lk(sm_); st_.quad_windows += tail[22]; st_.private_windows += tail[23]; }
            else (void)hipGetLastError(); } }
    if (b.any_chain && !failed) { std::lock_guard<std::mutex> lk(sm_); st_.chain_batches++; st_.chain_i_batches += b.chain_with_intra;
        for (auto &p : b.pics) st_.chain_pics += p.has_picture && (p.chain_ok || p.chain_intra); }
    { std::lock_guard<std::mutex> lk(m_); const int li = (int)(&ln - lanes_);
      for (auto &p : b.pics) { p.dec->engine_state().inflight--; p.dec->engine_state().lane_inflight[li]--; } pending_gen_++; }
    // a kernel whose bounded wait gave up (damaged hand-over between workgroups) left a code in the picture's error word: the handle reports it
    for (size_t i = 0; i < b.pics.size(); i++) if (b.h_err[i]) { b.pics[i].dec->on_device_wait_error(b.h_err[i]); b.h_err[i] = 0;
        std::lock_guard<std::mutex> lk(sm_); st_.wait_errors++; }
    for (auto &p : b.pics) p.dec->on_engine_done(p, failed);
    b.pics.clear();
}

// about once a second (every 200 ms while the GPU is shared, and at once after a chain knob was set): does another process have compute queues on this GPU?
// Engine thread only, no lock held.  A point-in-time sample: a process that arrives between two looks can still make a chain launch give up, which
// recover() then repairs.
void Engine::look_for_other_users() {
    if (!kfd_gpu_id_) return;
    const long long now = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const bool was = gpu_shared_.load(std::memory_order_relaxed);
    if (now - shared_checked_ns_.load() <= (was ? 200ll : 1000ll) * 1000 * 1000) return;
    shared_checked_ns_ = now;
    const bool sh = kfd_gpu_has_other_users(kfd_gpu_id_);
    if (sh != was) fprintf(stderr, "jm_amd_dec: device %d: %s -- chain launches %s\n", device_,
        sh ? "another process has compute queues on this GPU" : "the GPU is no longer shared", sh ? "off" : "on again");
    gpu_shared_.store(sh, std::memory_order_relaxed);
}

void Engine::run() {
    hipSetDevice(device_);
    // This thread's sleeps are its polling period (step 3 below: 20 us while batches are in flight; with the default timer slack of 50 us such a sleep takes ~75).
    // A 1 us slack was tried (JM_AMD_DEC_TIMER_SLACK_NS=1000): the end of a batch is noticed 50 us sooner, but the loop then takes m_ three times as often --
    // one stream and eight streams unchanged, 32 device-resident streams rather worse (profiles/r06_engine_loop.txt).  Default: the slack is left alone.
    { const char *e = getenv("JM_AMD_DEC_TIMER_SLACK_NS"); const unsigned long ns = e ? strtoul(e, nullptr, 10) : 0ul; if (ns) prctl(PR_SET_TIMERSLACK, ns, 0, 0, 0); }
    for (;;) {
        bool progressed = false;
        // 1. retire finished batches (oldest first per lane)
        for (auto &ln : lanes_) {
            while (ln.inflight > 0) {
                // hipErrorNotReady = still running.  Anything else that is not success is a sticky device error (kernel fault, lost device):
                // the batch is retired as FAILED -- its handles report the error and release their slots -- instead of being polled forever.
                const hipError_t q = hipEventQuery(ln.ring[ln.tail].done);
                if (q == hipErrorNotReady) break;
                if (q != hipSuccess && !device_failed_) { device_failed_ = true;
                    fprintf(stderr, "jm_amd_dec: device %d failed: %s -- every handle on it now returns errors\n", device_, hipGetErrorString(q)); }
                { auto t0 = std::chrono::steady_clock::now(); complete(ln, ln.ring[ln.tail], q != hipSuccess);
                    long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                    std::lock_guard<std::mutex> lk(sm_); st_.complete_ns += ns; }
                ln.tail = (ln.tail + 1) % kBatchRing; ln.inflight--;
                progressed = true;
            }
        }
        // 2. launch: at most two batches queued per lane, so that while they run new pictures pile up and batches stay full
        lingered_ = false;
        for (int li = 0; li < kLanes; li++) {
            Lane &ln = lanes_[li];
            static const int max_inflight = getenv("JM_AMD_DEC_INFLIGHT") ? atoi(getenv("JM_AMD_DEC_INFLIGHT")) : 2;
            // the intra lane takes one batch at a time while intra pictures may run ahead of their turn (deep queues): a second batch behind a running one would only fix its
            // membership early (it could not start sooner), and an intra batch costs the lane the same 1.8 ms whether it holds two pictures or twenty --
            // formed when the lane falls idle it takes everything that has become ready meanwhile (r06: 276 batches of 2.3 -> see profiles/r06_lane_fill.txt)
            if (ln.inflight >= (li == kIntraLane && early_intra_ && deep_queues_ ? 1 : max_inflight)) continue;
            Batch &b = ln.ring[ln.head];
            bool have;
            if (li == kOrdinaryLane) { bool any; { std::lock_guard<std::mutex> lk(m_); any = !pending_.empty(); } if (any) look_for_other_users(); }
            { std::lock_guard<std::mutex> lk(m_); have = !pending_.empty() && form(ln, li, b); }
            if (!have) continue;
            if (device_failed_) { complete(ln, b, true); progressed = true; continue; }      // nothing can run any more: fail the pictures right away
            { auto t0 = std::chrono::steady_clock::now(); launch(ln, b);
                long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                std::lock_guard<std::mutex> lk(sm_); st_.launch_ns += ns; }
            ln.head = (ln.head + 1) % kBatchRing; ln.inflight++;
            progressed = true;
        }
        if (progressed) continue;
        // 3. nothing to do right now
        bool busy = false;
        for (auto &ln : lanes_) busy |= ln.inflight > 0;
        // (a batch that waits for its missing streams -- Engine::form's fill linger -- is looked at again after a nap, not in a spin under m_: with pictures pending
        //  the wait below returns at once)
        if (busy || lingered_) std::this_thread::sleep_for(std::chrono::microseconds(20));
        else { std::unique_lock<std::mutex> lk(m_); cv_.wait_for(lk, std::chrono::milliseconds(2), [&] { return !pending_.empty(); }); }
    }
}

}  // namespace jmamd
