// jmcodec_amd/csrc/decoder.cpp -- see decoder.h.
#include "decoder.h"
#include "numa.h"
#include <pthread.h>
#include <time.h>
#include "engine.h"
#include "host_copy.h"
#include "kernels.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace jmamd {

#define HIP_OK(expr) ((expr) == hipSuccess)

// =============================================================================================
// parse worker pools: one per NUMA node that holds a GPU in use (one for the whole process when the host does not say where its GPUs sit)
// =============================================================================================
// A handle's pictures are entropy-decoded on the node of its GPU: the workers run on that node's CPUs and the page-locked job buffers they fill
// (and grow) come from its memory, so a job list crosses the socket interconnect neither on its way into the buffer nor on its way to the device.
// The process-wide thread budget (the CFS quota x 1.25, or JM_AMD_DEC_THREADS) is what ONE pool gets when the process drives one device
// (JM_AMD_DEC_DEVICE: a rank of bench.py, a process per GPU); in the drop-in "one process, handles round robin over every GPU" mode each node's
// pool gets the share of the budget that corresponds to its CPUs.
namespace {
int thread_budget() {
    const char *e = getenv("JM_AMD_DEC_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (!e) {
        // A container may see every CPU of the machine but own a small CFS quota (cgroup v2 cpu.max: "<quota> <period>"): a worker per
        // visible CPU then only buys throttling stalls.  Size the pool to the quota, with a quarter more for workers blocked on job slots or on
        // the collocated picture's motion: measured on a 16-CPU quota (scratch/gpu_threads.sh, frames/s with 16 / 20 / 28 workers) HEVC 1080p
        // 3.98 / 4.29 / 3.85 k, High 9.9 / 8.6 / 8.3 k, High + B 7.5 / 8.3 / 7.1 k, 4K High + B 1.21 / 1.32 / 1.33 k -- with 28 (x 1.75, the
        // former rule) the kernel throttled the process for seconds per second and a picture cost 15-25 % more CPU time.
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            long long q = 0, per = 0; char qs[32] = {0};
            if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0 && per > 0) { q = atoll(qs); int lim = (int)((q * 5 / 4 + per - 1) / per);
                if (lim < 4) lim = 4; if (n > lim) n = lim; }
            fclose(f);
        }
    }
    return std::max(1, std::min(n, 64));
}
struct Pool {
    std::mutex m; std::condition_variable cv;
    std::deque<std::pair<Decoder *, PicTask *>> q;
    std::vector<std::thread> threads;
    int n = 0, node = -1; bool bound = false;
    Pool(int node_, int n_) : n(n_), node(node_) {
        // The workers are bound to the node's CPUs only when this process may use enough of them: in a cpuset-restricted container that owns a couple
        // of CPUs on that node and more elsewhere, binding would pile all n workers onto those few.  Enough = the pool's share of the CPU
        // quota (n is quota x 1.25), at least half the pool.
        bound = (int)numa_cpus_of_node(node).size() >= std::max(2, (n * 4 / 5 + 1) / 2);
        for (int i = 0; i < n; i++) threads.emplace_back([this] { pthread_setname_np(pthread_self(),
            "jm-parse"); if (bound) numa_bind_this_thread(node); run(); });
        for (auto &t : threads) t.detach();
    }
    void run() {
        ParseScratch scratch;
        for (;;) {
            std::pair<Decoder *, PicTask *> job;
            { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return !q.empty(); }); job = q.front(); q.pop_front(); }
            try { job.first->parse_task(job.second, scratch); }
            catch (const std::exception &e) { job.first->parse_exception(job.second, e.what()); }
            catch (...) { job.first->parse_exception(job.second, "unknown exception"); }
        }
    }
};
std::mutex g_pools_m;
std::vector<Pool *> g_pools;                       // intentionally leaked: workers outlive static destruction
Pool &pool_of_node(int node) {
    std::lock_guard<std::mutex> lk(g_pools_m);
    for (Pool *p : g_pools) if (p->node == node) return *p;
    int n = thread_budget();
    if (node >= 0 && !getenv("JM_AMD_DEC_DEVICE")) {                       // many GPUs in one process: this node's share of the budget
        cpu_set_t allowed; CPU_ZERO(&allowed);
        const int total = sched_getaffinity(0, sizeof allowed, &allowed) == 0 ? CPU_COUNT(&allowed) : 0;
        const int mine = (int)numa_cpus_of_node(node).size();
        if (total > 0 && mine > 0 && mine < total) n = std::max(2, (n * mine + total - 1) / total);
    }
    g_pools.push_back(new Pool(node, n));
    return *g_pools.back();
}
std::atomic<int> g_handle_counter{0};
std::atomic<int> g_handle_index{0};
}  // namespace

void pool_submit(Decoder *d, PicTask *t) { Pool &p = pool_of_node(d->numa_node()); { std::lock_guard<std::mutex> lk(p.m); p.q.emplace_back(d, t); }
    p.cv.notify_one(); }
int pool_threads(int node) { return pool_of_node(node).n; }

// =============================================================================================
static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

Decoder::Decoder() { memset(info_, 0, sizeof info_); trace_on_ = getenv("JM_AMD_DEC_TRACE") != nullptr; eng_state_ = new EngineDecoderState(); }

Decoder::~Decoder() {
    // wait until no worker still references this object
    { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0 && parse_pending_ == 0; }); }
    { std::lock_guard<std::mutex> lk(submit_mtx_); }
    if (copier_) { copier_->free_signal(out_sig_); out_sig_ = 0; }
    gpu_close();
    delete eng_state_;
    if (trace_on_ && !trace_.empty()) {
        char name[256]; snprintf(name, sizeof name, "%s.%p.csv", getenv("JM_AMD_DEC_TRACE"), (void *)this);
        if (FILE *f = fopen(name, "w")) {
            fprintf(f, "seq,is_i,dispatch,parsed,submit0,submit1\n");
            for (auto &r : trace_)
                fprintf(f, "%llu,%d,%lld,%lld,%lld,%lld\n", (unsigned long long)r.seq, r.is_i, r.t_dispatch, r.t_parsed, r.t_submit0, r.t_submit1);
            fclose(f);
        }
    }
}

void Decoder::fail(const std::string &msg) {            // any thread
    std::lock_guard<std::mutex> lk(error_m_);
    if (!failed_) { error_ = msg; fprintf(stderr, "jm_amd_dec: %s\n", msg.c_str()); }
    failed_ = true;
}
// jm_amddec_last_error: the text is copied under the lock into a buffer only the calling thread's calls replace, so the pointer stays valid while
// parse workers and the engine thread go on reporting
const char *Decoder::last_error() {
    std::lock_guard<std::mutex> lk(error_m_);
    error_out_ = error_;
    return error_out_.c_str();
}

// remembers the most recent non-fatal error text for jm_amddec_last_error (any thread)
void Decoder::note_error(const std::string &msg) {
    std::lock_guard<std::mutex> lk(error_m_);
    if (!failed_) error_ = msg;
}

int Decoder::set_option(const char *key, long long v) {
    std::string k(key);
    if (k == "parse_only") parse_only_ = v != 0;
    else if (k == "digest") { want_digest_ = v != 0; if (want_digest_) sync_mode_ = true; }
    else if (k == "display_delay") display_delay_ = (int)std::max(0ll, std::min(v, (long long)n_jobs_ - 4));
    else if (k == "job_slots") { if (inited_) return -1; n_jobs_ = (int)std::max(8ll, std::min(v, (long long)kMaxJobSlots)); n_jobs_set_ = true; }
    else if (k == "fast_parse") fast_parse_ = v != 0;        // 0: the general macroblock path only (tests)
    // tests: FNV-1a over every picture's job list as the device gets it
    else if (k == "job_digest") { want_job_digest_ = v != 0; if (want_job_digest_) sync_mode_ = true; }
    else if (k == "sync") sync_mode_ = v != 0;
    else if (k == "device_output") device_output_ = v != 0;        // frames stay in device memory (no D2H); see output_device()
    else if (k == "device") device_ = (int)v;
    else if (k == "profile") { profile_ = v != 0; if (engine_) engine_->set_profile(profile_); }
    else if (k.rfind("chain_", 0) == 0 || k == "debug_stall" || k == "debug_no_bi" || k == "early_intra_ahead") {    // engine-wide knobs (every handle of the device), after init
        if (!engine_ || !engine_->set_knob(k, v)) return -1;
    }
    else if (k == "wait_idle") {      // block until every dispatched picture has been executed by the device (no flush)
        dispatch_pending();
        { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0; }); }
    }
    else return -1;
    return 0;
}
long long Decoder::get_stat(const char *key) const {
    std::string k(key);
    if (k == "inferred_frames") return stat_inferred_frames_.load();
    if (k == "redundant_slices") return stat_redundant_slices_.load();
    if (k == "field_pictures") return stat_field_pics_.load();
    if (k == "lone_fields") return stat_lone_fields_.load();
    if (k == "frames") return num_frames_;
    if (k == "pictures") return stat_pictures_;
    if (k == "job_bytes") return stat_job_bytes_;
    if (k == "job_regrown") return stat_job_regrown_;          // job slots grown on demand (a few per handle, then none)
    if (k == "job_slot_bytes") { long long n = 0; for (auto &j : jobs_) n += (long long)j.cap; return n; }   // page-locked job memory of this handle right now
    if (k == "errors") return stat_errors_;
    if (k == "copy_engines") return copier_ ? (long long)copier_->engine_mask() : 0;      // SDMA engines the direct route uses (bit mask)
    if (k == "direct_frames") return stat_direct_;
    // ... and the time their callers spent waiting for them // frames that went out by the "direct" route (one DMA into the caller's registered buffer)
    if (k == "direct_ns") return stat_direct_ns_;
    if (k == "device_wait_errors") return stat_wait_errors_;
    if (k == "intra_mbs") return stat_intra_mbs_;
    if (k == "coef_int16") return stat_coef_;
    if (k == "syntax_digest") return codec_ == 1 ? (long long)hdigest_.h : (long long)digest_.h;
    if (k == "job_digest") return (long long)job_digest_;
    if (k == "digest_mbs") return codec_ == 1 ? (long long)hdigest_.n_cu : (long long)digest_.mbs;
    if (k == "i_pictures") return stat_i_;
    if (k == "p_pictures") return stat_p_;
    if (k == "b_pictures") return stat_b_;
    // frame rate of the active sequence as a fraction (VUI timing information; 0 / 0 when the stream carries none): H.264 counts FIELD ticks (E.2.1)
    if (k == "fps_num") return codec_ == 1 ? (long long)hsps_.time_scale : (long long)seq_.time_scale;
    if (k == "fps_den") return codec_ == 1 ? (long long)hsps_.num_units_in_tick : 2ll * seq_.num_units_in_tick;
    // display frames decided and not yet made current by a decode / poll call (finished or still on the device)
    if (k == "frames_done_unfetched") return (long long)frames_done_unfetched();
    if (k == "frames_waiting") { std::lock_guard<std::mutex> lk(const_cast<std::mutex &>(mtx_)); return (long long)ready_.size(); }
    // pictures dispatched (being parsed, waiting for the engine, on the device) whose completion the handle has not seen yet
    if (k == "pictures_in_flight") { std::lock_guard<std::mutex> lk(const_cast<std::mutex &>(mtx_)); return (long long)outstanding_; }
    if (k == "coded_width") return mb_w_ * 16;
    if (k == "coded_height") return mb_h_ * 16;
    if (k == "pitch") return pitch_;
    if (k == "device") return device_;
    if (k == "threads") return pool_threads(numa_node_);       // workers of the parse pool this handle uses (the pool of its GPU's NUMA node)
    if (k == "numa_node") return numa_node_;
    if (k == "elapsed_us") return (long long)(elapsed_ms_ * 1000.0);
    if (k == "parse_ns_i") return stat_parse_ns_i_;
    if (k == "submit_ns") return stat_submit_ns_;
    if (k == "wait_slot_ns") return stat_wait_slot_ns_;
    if (k == "parse_ns_p") return stat_parse_ns_p_;
    if (k.rfind("k_", 0) == 0 || k.rfind("eng_", 0) == 0) {          // engine-wide (all handles on this device), profile option
        static const char *kn[5] = {"inter", "intra", "deblock", "packout", "chain"};
        if (!engine_) return 0;
        EngineStats es = engine_->stats();
        for (int i = 0; i < 5; i++) {
            if (k == std::string("k_") + kn[i] + "_ns") return (long long)es.ns[i];
            if (k == std::string("k_") + kn[i] + "_n") return es.launches[i];
            if (k == std::string("k_") + kn[i] + "_pics") return es.pics[i];
            if (k == std::string("k_") + kn[i] + "_alg_bytes") return es.alg_bytes[i];
        }
        if (k == "eng_batches") return es.batches;
        if (k == "eng_forms") return es.forms;
        if (k == "eng_form_decoders") return es.form_decoders;
        if (k == "eng_form_pending") return es.form_pending;
        if (k == "eng_batch_pics") return es.batch_pics;
        if (k == "eng_chain_batches") return es.chain_batches;
        if (k == "eng_chain_pics") return es.chain_pics;
        if (k == "eng_chain_i_batches") return es.chain_i_batches;
        if (k == "eng_wait_errors") return es.wait_errors;
        if (k == "eng_chain_recoveries") return es.chain_recoveries;
        if (k == "eng_gpu_shared") return engine_->gpu_shared() ? 1 : 0;
        if (k == "eng_launch_ns") return es.launch_ns;
        if (k == "eng_wait_gap_launches") return es.wait_gap_launches;
        if (k == "eng_wait_gap_max_us") return es.wait_gap_max_ticks / 100;
        if (k == "eng_quad_windows") return es.quad_windows;
        if (k == "eng_private_windows") return es.private_windows;
        if (k == "eng_rej_other_lane") return es.rej_other_lane;
        if (k == "eng_rej_cross_lane") return es.rej_cross_lane;
        if (k == "eng_rej_tables") return es.rej_tables;
        if (k == "eng_early_intra") return es.early_intra;
        if (k == "eng_blocked_ns") return es.blocked_ns;
        if (k == "eng_blocked_n") return es.blocked_n;
        for (int i = 0; i < 4; i++) {
            const std::string l = std::to_string(i);
            if (k == "eng_lane" + l + "_busy_ns") return (long long)es.lane_busy_ns[i];
            if (k == "eng_lane" + l + "_gap_ns") return (long long)es.lane_gap_ns[i];
            if (k == "eng_lane" + l + "_upwait_ns") return (long long)es.lane_upwait_ns[i];
            if (k == "eng_lane" + l + "_prewait_ns") return (long long)es.lane_prewait_ns[i];
            if (k == "eng_lane" + l + "_dry") return es.lane_dry[i];
            if (k == "eng_lane" + l + "_batches") return es.lane_batches[i];
            if (k == "eng_lane" + l + "_pics") return es.lane_pics[i];
        }
        if (k == "eng_complete_ns") return es.complete_ns;
        return -1;
    }
    if (k.rfind("display_poc:", 0) == 0) { size_t i = (size_t)atoll(k.c_str() + 12); return i < display_pocs_.size() ? display_pocs_[i] : -1; }
    return -1;
}

// nvdec_decode_init (nv_dec.cpp:62-80): the reference always returns 0; we return <0 when no
// MI355X-class device can be opened, because silently continuing would mean a CPU fallback.
int Decoder::init(int codec_type, int out_fmt, const uint8_t *extra, int len) {
    codec_ = codec_type; out_fmt_ = out_fmt ? 1 : 0;
    if (codec_type != 0 && codec_type != 1) { fail("only codec_type 0 (H.264) and 1 (HEVC) are implemented"); return -1; }
    if (getenv("JM_AMD_DEC_SYNC")) sync_mode_ = true;
    if (const char *js = getenv("JM_AMD_DEC_JOB_SLOTS")) { n_jobs_ = std::max(8, std::min(atoi(js), kMaxJobSlots)); n_jobs_set_ = true; }
    if (const char *dd = getenv("JM_AMD_DEC_DISPLAY_DELAY")) display_delay_ = std::max(0, std::min(atoi(dd), n_jobs_ - 4));
    if (getenv("JM_AMD_DEC_PARSE_ONLY")) parse_only_ = true;
    out_via_copy_engine_ = !getenv("JM_AMD_DEC_OUT_DIRECT");
    if (getenv("JM_AMD_DEC_DEVICE_OUTPUT")) device_output_ = true;      // host-side tests of callers that cannot reach set_option (jm_intel_dec_* facade)
    cavlc_init_tables();
    // (no HIP call: only the test override can place a parse-only handle)
    if (parse_only_) numa_node_ = numa_node_of_device(device_ < 0 ? 0 : device_, false);
    if (!parse_only_ && !gpu_open()) return -1;
    // Where a display frame waits for jm_nvdec_output_frame.  k_packout writes the tight frame into a device staging buffer of the output slot; then
    //   direct : (default) it stays there, and jm_nvdec_output_frame moves it into the caller's buffer with ONE copy-engine transfer on the ROCr layer
    //            (host_copy.h): no staging hop, no CPU copy, the caller asleep meanwhile.  With three SDMA engines in turn the PCIe link is what bounds
    //            the rate: 16.8-17.0 k frames/s of 1080p at 0.63 ms of CPU per frame (32 streams), against 14.7 k at 0.89 ms for the mix below;
    //   pinned : a copy engine moves it to the slot's pinned host buffer ahead of time, jm_nvdec_output_frame is one CPU memcpy (0.3-0.57 ms of CPU per frame);
    //   fetch  : it stays in device staging and jm_nvdec_output_frame is a plain synchronous hipMemcpy (the copies of a process queue behind each
    //            other on the one engine the HIP runtime uses for this direction: 11.7 k frames/s).
    // JM_AMD_DEC_OUT_FETCH = "direct" | "a/b" (a of every b handles fetch, the rest pinned: the round-1 default was 2/5) | "auto" (per frame: fetch when
    // no other thread is inside such a copy); JM_AMD_DEC_OUT_PINNED=1 = pinned slots for every handle.  When the ROCr layer cannot be reached the
    // default falls back to 2/5.
    {
        int fa = 2, fb = 5;
        const char *e = getenv("JM_AMD_DEC_OUT_FETCH");
        out_route_ = 3;                                                     // 0 auto (per frame), 1 always fetch, 2 always pinned, 3 direct
        if (getenv("JM_AMD_DEC_OUT_PINNED") || !out_via_copy_engine_) out_route_ = 2;
        else if (e && !strcmp(e, "auto")) out_route_ = 0;
        else if (e && strcmp(e, "direct")) { if (sscanf(e, "%d/%d", &fa, &fb) != 2 || fb <= 0) { fa = 2; fb = 5; }
            out_route_ = (handle_index_ % fb) < fa ? 1 : 2; }
        if (out_route_ == 3 && !parse_only_) {
            copier_ = HostCopier::get(device_);
            if (copier_) out_sig_ = copier_->new_signal();
            if (!copier_ || !out_sig_) { copier_ = nullptr; out_route_ = (handle_index_ % 5) < 2 ? 1 : 2; }
        }
        out_fetch_ = out_route_ == 1 || out_route_ == 3;
        if (const char *l = getenv("JM_AMD_DEC_FETCH_LIMIT")) fetch_limit_ = atoi(l);
    }
    if (engine_) engine_->set_profile(profile_);
    inited_ = true;
    // optional SPS/PPS given up front (nv_dec.cpp:334-360).  FFmpeg hands test_player either Annex-B parameter sets or, for MP4 /
    // MKV sources, an AVCDecoderConfigurationRecord ("avcC", ISO/IEC 14496-15 5.2.4.1); with avcC the packets that follow are
    // length-prefixed NAL units unless a bitstream filter already converted them (test_player.cpp:221-226 uses h264_mp4toannexb).
    if (extra && len > 0 && codec_ == 1 && len >= 23 && extra[0] == 1) {
        // HEVCDecoderConfigurationRecord ("hvcC", ISO/IEC 14496-15 8.3.3.1): what a demuxer hands test_player for HEVC in MP4 / MKV
        avcc_len_size_ = (extra[21] & 3) + 1;
        int o = 23;
        for (int a = 0, n_arrays = extra[22]; a < n_arrays && o + 3 <= len; a++) {
            int n = (extra[o + 1] << 8) | extra[o + 2]; o += 3;
            for (int i = 0; i < n && o + 2 <= len; i++) {
                int l = (extra[o] << 8) | extra[o + 1]; o += 2;
                if (l <= 0 || o + l > len) { o = len; break; }
                handle_nal(extra + o, (size_t)l); o += l;
            }
        }
    } else if (extra && len > 0) {
        if (len >= 7 && extra[0] == 1) {
            avcc_len_size_ = (extra[4] & 3) + 1;
            int o = 5, n_sps = extra[o++] & 31;
            for (int pass = 0; pass < 2; pass++) {
                int n = pass == 0 ? n_sps : (o < len ? extra[o++] : 0);
                for (int i = 0; i < n && o + 2 <= len; i++) {
                    int l = (extra[o] << 8) | extra[o + 1]; o += 2;
                    if (l <= 0 || o + l > len) { o = len; break; }
                    handle_nal(extra + o, (size_t)l); o += l;
                }
            }
        } else { feed(extra, (size_t)len); static const uint8_t sc[4] = {0, 0, 0, 1}; feed(sc, 4); in_.clear(); scan_ = 0; have_start_ = false; }
    }
    return 0;
}

// =============================================================================================
// device resources (all device WORK is issued by the per-device Engine; the decoder only owns memory)
// =============================================================================================
bool Decoder::gpu_open() {
    int n = 0;
    if (!HIP_OK(hipGetDeviceCount(&n)) || n <= 0) { fail("no HIP device available (the HIP backend is mandatory; there is no CPU fallback)"); return false; }
    if (device_ < 0) {
        const char *e = getenv("JM_AMD_DEC_DEVICE");
        device_ = e ? atoi(e) : (g_handle_counter++ % n);
    }
    // (the test override JM_AMD_DEC_FAKE_NUMA is keyed by the device index the CALLER named: "device 1" on a one-GPU box is device 0 again, but keeps the
    //  node the override gives device 1 -- the many-GPUs-in-one-process mode, two parse pools and all, on real hardware with one GPU)
    const int named_device = device_;
    if (device_ >= n) device_ %= n;
    mem_trace("handle: before hipSetDevice");
    if (!HIP_OK(hipSetDevice(device_))) { fail("hipSetDevice failed"); return false; }
    mem_trace("handle: hipSetDevice");
    handle_index_ = g_handle_index++;
    numa_node_ = numa_node_of_device(getenv("JM_AMD_DEC_FAKE_NUMA") ? named_device : device_, true);
    engine_ = Engine::get(device_);
    if (!engine_) { fail("could not start the device engine (stream / buffer creation failed)"); return false; }
    gpu_open_ = true;
    mem_trace("handle: engine ready");
    return true;
}

// job slots and the big buffers they borrow: device build (page-locked + device copies) or parse-only build (plain memory).  Nothing is in flight.
void Decoder::free_job_buffers() {
    for (auto &j : jobs_) {
        if (j.big >= 0) { j.host = j.own_host; j.dev = j.own_dev; j.big = -1; }       // (a borrowed buffer is freed with the others below)
        if (gpu_open_ && !parse_only_) { if (j.host) hipHostFree(j.host); if (j.dev) hipFree(j.dev); if (j.dbrec) hipFree(j.dbrec);
            if (j.resid) hipFree(j.resid); if (j.uploaded) hipEventDestroy(j.uploaded); }
        else free(j.host);
        j = JobSlot();
    }
    for (auto &b : big_) {
        if (gpu_open_ && !parse_only_) { if (b.host) hipHostFree(b.host); if (b.dev) hipFree(b.dev); } else free(b.host);
        b = BigJobBuf();
    }
}

void Decoder::free_surfaces() {
    if (surf_block_) hipFree(surf_block_);
    surf_block_ = nullptr;
    for (auto &p : surf_) p = nullptr;
}

void Decoder::gpu_free_sequence() {
    if (!gpu_open_) return;
    hipSetDevice(device_);
    free_surfaces();
    if (resid_) { hipFree(resid_); resid_ = nullptr; }
    for (auto &w : hevc_work_) if (w) { hipFree(w); w = nullptr; }
    if (hevc_bs_) { hipFree(hevc_bs_); hevc_bs_ = nullptr; }
    free_job_buffers();
    free_out_slots(true);
}
// releases output slots: every one (teardown) or only those the application is not waiting for (resolution change: frames of the
// old size that were decoded but not fetched yet stay valid; their slots are freed when they come back)
void Decoder::free_out_slots(bool all) {
    std::lock_guard<std::mutex> lk(mtx_);
    std::vector<OutSlot *> keep;
    for (OutSlot *o : all_out_) {
        bool pending = !all && (o == cur_out_ || std::find(ready_.begin(), ready_.end(), o) != ready_.end());
        if (pending) { keep.push_back(o); continue; }
        if (o->host) hipHostFree(o->host);
        if (o->dev) hipFree(o->dev);
        delete o;
    }
    all_out_ = keep; free_out_.clear();
    if (all) { ready_.clear(); done_unfetched_ = 0; cur_out_ = nullptr; }
}
void Decoder::gpu_close() {
    if (!gpu_open_) { for (OutSlot *o : all_out_) delete o; all_out_.clear(); free_job_buffers(); return; }      // (parse-only handles: plain malloc)
    gpu_free_sequence();
    gpu_open_ = false;
}

// nvdec_create_decoder (nv_dec.cpp:496-540): surfaces sized by the coded picture, NV12
bool Decoder::gpu_alloc_sequence() {
    size_t n_mbs = (size_t)mb_w_ * mb_h_;
    frame_bytes_ = (size_t)disp_w_ * disp_h_ * 3 / 2;
    // MbRec + worst-case coefficients + motion records (16 vectors; 72 int16 for B / weighted slices) + slice tables
    // (always the Main / High layout: a later SPS of the same size may switch profile without re-activation, and a PPS may enable weighted
    //  prediction under any profile_idc)
    job_cap_max_ = n_mbs * (sizeof(MbRec) + 816 + kBiRecInt16 * 2) + 256 * (sizeof(SliceRec) + sizeof(SliceWp)) + 4096;
    // The 24 slots are sized for an ordinary picture -- the fixed records plus 128 bytes per macroblock of levels and motion (config C1 measures
    // 0.63 MB per 1080p picture) -- and three worst-case buffers per handle are LENT to I pictures, which need several times that
    // (acquire_job_slot).  Nothing is allocated while pictures are in flight: hipHostMalloc / hipFree cost milliseconds and synchronise the device,
    // and one stream's chain of pictures stalls for every one of them (a first version grew slots on demand: single stream 4.0 k -> 3.1 k frames/s).
    // A P / B picture that outgrows its slot all the same is parsed again into a grown slot (parse_task): rare, and then the slot stays bigger.
    // 24 worst-case slots were 195 MB of page-locked memory per 1080p handle (14 GB for the bench's 32 handles: VERDICT r2 weak 11); now 55 MB.
    // High streams get 320 bytes per macroblock: their P / B pictures average 125 (High I/P, CABAC + 8x8 transform, QP 28) to 195 bytes (with B
    // pictures: a 144-byte motion record per bi-predicted macroblock) -- with the Baseline allowance every second picture of such a stream was parsed twice
    // and its slot re-allocated mid-stream (`job_slots_grown` 344 / 760 in a bench run of High / High + B: 0.2 ms of kernel time per picture in the
    // parse workers, 10.9 -> 9.6 k frames/s against round 2's worst-case slots on the same box).
    // (Main / Extended: 224 -- B pictures, no 8x8 transform.)
    const size_t per_mb = seq_.profile_idc == 66 ? 128 : (seq_.profile_idc < 100 ? 224 : 320);
    job_cap_ = std::min(job_cap_max_, n_mbs * (sizeof(MbRec) + per_mb) + 256 * (sizeof(SliceRec) + sizeof(SliceWp)) + 4096);
    if (getenv("JM_AMD_DEC_JOB_WORST_CASE")) job_cap_ = job_cap_max_;
    if (codec_ == 1) job_cap_ = n_mbs * 128 + (1u << 20);            // HEVC job lists vary a lot in size: start small, grow on demand (ensure_job_cap)
    if (!n_jobs_set_) n_jobs_ = codec_ == 0 && n_mbs <= 8704 ? kJobSlotsSmall : kJobSlots;      // (decoder.h)
    const bool lend_big = codec_ == 0 && job_cap_ < job_cap_max_;
    if (parse_only_) {
        for (int i = 0; i < n_jobs_; i++) { jobs_[i].host = (uint8_t *)malloc(job_cap_); jobs_[i].cap = job_cap_; }
        if (lend_big) for (auto &b : big_) b.host = (uint8_t *)malloc(job_cap_max_);
        return true;
    }
    hipSetDevice(device_);
    pitch_ = (mb_w_ * 16 + 127) & ~127;
    chroma_off_ = pitch_ * mb_h_ * 16;
    surf_bytes_ = (size_t)pitch_ * mb_h_ * 16 * 3 / 2;
    {   // every surface in one allocation (256-byte aligned strides): a picture's references then lie within one 32-bit offset range
        const size_t stride = (surf_bytes_ + 255) & ~(size_t)255, n = (size_t)(n_surf_ + extra_surf_);
        if (!HIP_OK(hipMalloc((void **)&surf_block_, stride * n))) { fail("hipMalloc(surfaces) failed"); return false; }
        hipMemset(surf_block_, 128, stride * n);
        for (size_t i = 0; i < n; i++) surf_[i] = surf_block_ + i * stride;
    }
    // the banded LDS wavefronts hold pictures of up to 512 macroblock rows (16 rows x 32 bands); the one-workgroup spin-wait kernels k_deblock / k_recon_intra
    // remain for taller ones (a legal H.264 picture may have up to 1,055 rows), k_recon_intra also for P / B pictures with a few scattered intra macroblocks
    use_lds_deblock_ = deblock_lds_supported(mb_w_, mb_h_);
    use_lds_intra_ = intra_lds_supported(mb_w_, mb_h_);
    lds_intra8_ = true;
    chain_ok_ = codec_ == 0 && use_lds_deblock_ && chain_supported(mb_w_, mb_h_);
    chain_intra_on_ = !getenv("JM_AMD_DEC_NO_CHAIN_INTRA");
    // HEVC: kHevcWorkSets residual scratches and pre-SAO work surfaces per handle (H.264: a scratch per job slot)
    if (codec_ == 1) {
        if (!HIP_OK(hipMalloc((void **)&resid_, n_mbs * 768 * kHevcWorkSets))) { fail("hipMalloc(scratch) failed"); return false; }
        for (auto &w : hevc_work_) { if (!HIP_OK(hipMalloc((void **)&w, surf_bytes_))) { fail("hipMalloc(work surface) failed"); return false; }
            hipMemset(w, 128, surf_bytes_); }
        {   // boundary strengths on the device: pu_map (4 B per 4x4 cell), four flag planes (1 B each), bs_v, bs_h -- every part 256-byte aligned
            const size_t cells = (size_t)mb_w_ * 4 * mb_h_ * 4, al = 255;
            hevc_bs_off_[0] = 0; hevc_bs_off_[1] = (cells * 4 + al) & ~al; hevc_bs_off_[2] = hevc_bs_off_[1] + ((cells * 4 + al) & ~al);
            hevc_bs_off_[3] = hevc_bs_off_[2] + ((cells / 2 + al) & ~al);                     // bs_v: (w / 8) x (h / 4) = cells / 2; bs_h the same
            hevc_bs_set_bytes_ = hevc_bs_off_[3] + ((cells / 2 + al) & ~al);
            if (!HIP_OK(hipMalloc((void **)&hevc_bs_, hevc_bs_set_bytes_ * kHevcWorkSets))) { fail("hipMalloc(strength maps) failed"); return false; }
            hipMemset(hevc_bs_, 0, hevc_bs_set_bytes_ * kHevcWorkSets);
        }
    }
    NumaPreferred on_gpu_node(numa_node_);          // the page-locked job buffers (and output slots) of this handle: memory of the GPU's node
    for (int ji = 0; ji < n_jobs_; ji++) {
        JobSlot &j = jobs_[ji];
        if (!HIP_OK(hipHostMalloc((void **)&j.host, job_cap_, hipHostMallocDefault)) || !HIP_OK(hipMalloc((void **)&j.dev, job_cap_)) ||
            (codec_ == 0 && (!HIP_OK(hipMalloc((void **)&j.dbrec, n_mbs * 96)) || !HIP_OK(hipMalloc((void **)&j.resid, n_mbs * 768)))) ||
            !HIP_OK(hipEventCreateWithFlags(&j.uploaded, hipEventDisableTiming))) { fail("job buffer allocation failed"); return false; }
        j.cap = job_cap_;
    }
    if (lend_big) for (auto &b : big_) if (!HIP_OK(hipHostMalloc((void **)&b.host, job_cap_max_, hipHostMallocDefault)) || !HIP_OK(hipMalloc((void **)&b.dev,
        job_cap_max_))) { fail("job buffer allocation failed"); return false; }
    // output slots: allocate the steady-state population now (hipHostMalloc costs milliseconds and serialises
    // inside the runtime; it must never happen while pictures are in flight)
    {
        std::lock_guard<std::mutex> lk(mtx_);
        std::vector<OutSlot *> tmp;
        for (int i = 0; i < n_jobs_ + 4; i++) tmp.push_back(alloc_out_slot());
        for (OutSlot *o : tmp) free_out_.push_back(o);
    }
    return !failed_;
}

OutSlot *Decoder::alloc_out_slot() {   // mtx_ held
    while (!free_out_.empty()) {
        OutSlot *o = free_out_.back(); free_out_.pop_back();
        if (o->bytes == frame_bytes_ || parse_only_) { o->ready = false; o->has_data = false; o->w = disp_w_; o->h = disp_h_; return o; }
        // a slot of the previous resolution came back: release it
        all_out_.erase(std::remove(all_out_.begin(), all_out_.end(), o), all_out_.end());
        if (o->host) hipHostFree(o->host);
        if (o->dev) hipFree(o->dev);
        delete o;
    }
    OutSlot *o = new OutSlot();
    o->w = disp_w_; o->h = disp_h_;
    if (!parse_only_) {
        hipSetDevice(device_);
        if (!device_output_ && out_route_ != 1 && out_route_ != 3 && !HIP_OK(hipHostMalloc((void **)&o->host, frame_bytes_,
            hipHostMallocDefault))) fail("output buffer allocation failed");
        if ((out_via_copy_engine_ || device_output_) && !HIP_OK(hipMalloc((void **)&o->dev, frame_bytes_))) fail("output staging allocation failed");
        o->bytes = frame_bytes_;
    }
    all_out_.push_back(o);
    return o;
}

// =============================================================================================
// front end: Annex-B splitting (replaces the NAL scanning half of cuvidParseVideoData, nv_dec.cpp:394)
// =============================================================================================
void Decoder::feed(const uint8_t *buf, size_t len) {
    if (avcc_len_size_ && !have_start_ && in_.empty()) {
        // avcC mode: a packet is a whole number of length-prefixed NAL units -- unless it starts with a start code (already converted)
        // (a 4-byte length of 256..511 reads 00 00 01 xx, so the first bytes alone cannot tell the two forms apart: the packet is
        //  length-prefixed exactly when its length fields chain to its end)
        bool prefixed = false;
        { size_t o = 0; while (o + (size_t)avcc_len_size_ <= len) { size_t l = 0; for (int i = 0; i < avcc_len_size_; i++) l = (l << 8) | buf[o + i];
            o += (size_t)avcc_len_size_; if (l == 0 || l > len - o) { o = len + 1; break; } o += l; } prefixed = o == len; }
        if (prefixed) {
            size_t o = 0;
            while (o + (size_t)avcc_len_size_ <= len) {
                size_t l = 0;
                for (int i = 0; i < avcc_len_size_; i++) l = (l << 8) | buf[o + i];
                o += (size_t)avcc_len_size_;
                if (l == 0 || l > len - o) break;
                handle_nal(buf + o, l); o += l;
            }
            return;
        }
    }
    in_.insert(in_.end(), buf, buf + len);
    const uint8_t *p = in_.data();
    size_t n = in_.size();
    for (;;) {
        // find next start code 00 00 01 at or after scan_
        size_t pos = std::string::npos;
        for (size_t i = scan_; i + 3 <= n;) {
            const uint8_t *q = (const uint8_t *)memchr(p + i + 2, 1, n - (i + 2));
            if (!q) { i = n; break; }
            size_t k = (size_t)(q - p);
            if (p[k - 1] == 0 && p[k - 2] == 0) { pos = k - 2; break; }
            i = k - 1;
        }
        if (pos == std::string::npos) { scan_ = n >= 2 ? n - 2 : 0; if (have_start_ && scan_ < nal_start_) scan_ = nal_start_; break; }
        if (have_start_) {
            size_t end = pos;
            while (end > nal_start_ && p[end - 1] == 0) end--;
            if (end > nal_start_) handle_nal(p + nal_start_, end - nal_start_);
        }
        have_start_ = true; nal_start_ = pos + 3; scan_ = pos + 3;
    }
    // drop consumed bytes
    if (have_start_ && nal_start_ > (1u << 16)) {
        size_t drop = nal_start_ - 3;
        in_.erase(in_.begin(), in_.begin() + (long)drop);
        nal_start_ -= drop; scan_ = scan_ > drop ? scan_ - drop : 0;
        if (scan_ < nal_start_) scan_ = nal_start_;
    } else if (!have_start_ && in_.size() > 4) {
        in_.erase(in_.begin(), in_.end() - 3); scan_ = 0;
    }
}

// ENDOFSTREAM packet (nv_dec.cpp:389-392)
void Decoder::flush_stream() {
    if (have_start_) {
        size_t end = in_.size();
        while (end > nal_start_ && in_[end - 1] == 0) end--;
        if (end > nal_start_) handle_nal(in_.data() + nal_start_, end - nal_start_);
    }
    in_.clear(); have_start_ = false; scan_ = 0;
    dispatch_pending();
    auto t = std::make_unique<PicTask>();
    t->out_before = std::move(carry_out_); carry_out_.clear();
    flush_dpb(t->out_before);
    push_task(std::move(t));
}

static bool same_picture(const SliceHeader &a, const SliceHeader &b) {      // 7.4.1.2.4
    if (a.frame_num != b.frame_num || a.pps_id != b.pps_id) return false;
    if (a.field_pic != b.field_pic || a.bottom_field != b.bottom_field) return false;
    if ((a.nal_ref_idc == 0) != (b.nal_ref_idc == 0)) return false;
    if (a.poc_lsb != b.poc_lsb || a.delta_poc_bottom != b.delta_poc_bottom) return false;
    if (a.delta_poc[0] != b.delta_poc[0] || a.delta_poc[1] != b.delta_poc[1]) return false;
    if (a.idr != b.idr) return false;
    if (a.idr && a.idr_pic_id != b.idr_pic_id) return false;
    return true;
}

void Decoder::handle_nal(const uint8_t *nal, size_t len) {
    if (codec_ == 1) { hevc_handle_nal(nal, len); return; }
    if (failed_ || len < 1 || (nal[0] & 0x80)) return;
    int ref_idc = (nal[0] >> 5) & 3, type = nal[0] & 31;
    if (type != 1 && type != 5 && type != 7 && type != 8) {
        if (type == 9 || type == 6 || type == 10 || type == 11) dispatch_pending();
        return;
    }
    std::vector<uint8_t> rbsp(len + Rbsp::kSlack);
    size_t n = Rbsp::unescape(nal + 1, len - 1, rbsp.data());
    BitReader br(rbsp.data(), n);
    if (type == 7 || type == 8) {
        dispatch_pending();
        std::string e = type == 7 ? ps_.parse_sps(br) : ps_.parse_pps(br);
        if (!e.empty()) { stat_errors_++; note_error(e); }
        return;
    }
    SliceHeader sh;
    std::string e = ps_.parse_slice_header(br, type, ref_idc, sh);
    if (!e.empty()) { stat_errors_++; note_error(e); return; }
    // a redundant coded picture repeats (part of) the primary one for the case that it was lost; the primary picture is what is decoded (7.4.3: "the
    // decoding process ... of redundant coded pictures is not specified"), its redundant copies are dropped
    if (sh.redundant_pic_cnt > 0) { stat_redundant_slices_++; return; }
    const PicParamSet &pps = ps_.pps[sh.pps_id];
    const SeqParams &sps = ps_.sps[pps.sps_id];
    if (pending_ && !same_picture(first_sh_, sh)) dispatch_pending();
    if (!pending_) { if (!start_picture(sh, sps, pps)) return; }
    add_slice(sh, std::move(rbsp), n);
}

// =============================================================================================
// sequence activation, POC, DPB
// =============================================================================================
bool Decoder::activate(const SeqParams &sps) {
    // a new coded video sequence needs new device resources when the coded size changes, and new output slots / stream_info when only the
    // cropping does (hevc_activate compares the display size too); job buffers are sized for every profile, so a Baseline -> High switch
    // at the same size needs nothing
    bool changed = !seq_active_ || sps.mb_w != mb_w_ || sps.mb_h != mb_h_ || sps.disp_w() != disp_w_ || sps.disp_h() != disp_h_;
    seq_ = sps;
    dpb_size_ = sps.dpb_frames();
    // display order == decode order when POC type 2 (8.2.1.3): no bumping delay needed
    reorder_depth_ = sps.poc_type == 2 ? 0 : (sps.max_num_reorder_frames >= 0 ? sps.max_num_reorder_frames : dpb_size_);
    if (!changed) return true;
    if (seq_active_) {
        // resolution change: drain everything that still refers to the old surfaces
        auto t = std::make_unique<PicTask>();
        t->out_before = std::move(carry_out_); carry_out_.clear();
        push_task(std::move(t));
        { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0 && parse_pending_ == 0; }); }
        // a new coded video sequence with another picture size (cuvid_handle_video_sequence re-creates the decoder, nv_dec.cpp:23-30):
        // nothing is in flight any more, so surfaces, job rings and scratch are rebuilt; frames already decoded keep their slots
        if (gpu_open_) {
            hipSetDevice(device_);
            free_surfaces();
            if (resid_) { hipFree(resid_); resid_ = nullptr; }
            for (auto &w : hevc_work_) if (w) { hipFree(w); w = nullptr; }
            if (hevc_bs_) { hipFree(hevc_bs_); hevc_bs_ = nullptr; }
            free_job_buffers();
            free_out_slots(false);
        } else free_job_buffers();
    }
    mb_w_ = sps.mb_w; mb_h_ = sps.mb_h; disp_w_ = sps.disp_w(); disp_h_ = sps.disp_h();
    n_surf_ = 18;                                  // 16 (max DPB) + current + one spare; also covers later SPSs with a larger DPB
    for (auto &d : dpb_) d = DpbPic();
    if (!gpu_alloc_sequence()) return false;
    seq_active_ = true;
    if (!timer_started_) { t0_ = std::chrono::steady_clock::now(); timer_started_ = true; }   // nv_dec.cpp:537
    return true;
}

// 8.2.1: TopFieldOrderCnt and / or BottomFieldOrderCnt of the current frame or field (into fpoc[] of its store); returns PicOrderCnt of the picture:
// Min(top, bottom) of a frame, a field's own count
int Decoder::compute_poc(const SliceHeader &sh, DpbPic &store) {
    const SeqParams &s = seq_;
    int max_fn = 1 << s.log2_max_frame_num;
    long long top = 0, bot = 0;
    if (s.poc_type == 0) {
        int max_lsb = 1 << s.log2_max_poc_lsb;
        // prevPicOrderCntMsb / Lsb are those of the previous REFERENCE picture in decoding order (for the second field of a reference frame: its first
        // field); when it carried operation 5 they are 0 and its TopFieldOrderCnt after the operation -- dispatch_pending() stores that, and
        // non-reference pictures in between leave it alone (8.2.1.1)
        int prev_msb = sh.idr ? 0 : prev_poc_msb_, prev_lsb = sh.idr ? 0 : prev_poc_lsb_;
        long long msb = prev_msb;
        if (sh.poc_lsb < prev_lsb && prev_lsb - sh.poc_lsb >= max_lsb / 2) msb = (long long)prev_msb + max_lsb;
        else if (sh.poc_lsb > prev_lsb && sh.poc_lsb - prev_lsb > max_lsb / 2) msb = (long long)prev_msb - max_lsb;
        if (msb > (1ll << 30) || msb < -(1ll << 30)) msb = 0;                     // > 2^14 wraps of a 16-bit lsb without an IDR picture: not a real stream
        if (sh.nal_ref_idc) { prev_poc_msb_ = (int)msb; prev_poc_lsb_ = sh.poc_lsb; }
        // a frame or a top field: TopFieldOrderCnt = msb + lsb, and a frame's bottom field lies delta_pic_order_cnt_bottom from it; a bottom field
        // picture: BottomFieldOrderCnt = msb + lsb
        top = bot = msb + sh.poc_lsb;
        if (!sh.field_pic) bot = top + sh.delta_poc_bottom;
    } else {
        // 64-bit arithmetic: offsets are se(v) of a hostile stream, and sums of them must not overflow (the result is truncated, never UB)
        long long prev_off = prev_mmco5_ ? 0 : prev_frame_num_offset_, prev_fn = prev_mmco5_ ? 0 : prev_frame_num_;
        long long off = sh.idr ? 0 : (prev_fn > sh.frame_num ? prev_off + max_fn : prev_off);
        if (off > (1ll << 40)) off = 0;
        prev_frame_num_offset_ = off;
        if (s.poc_type == 2) top = bot = sh.idr ? 0 : (sh.nal_ref_idc ? 2 * (off + sh.frame_num) : 2 * (off + sh.frame_num) - 1);
        else {
            long long abs_fn = s.num_ref_frames_in_poc_cycle ? off + sh.frame_num : 0;
            if (!sh.nal_ref_idc && abs_fn > 0) abs_fn--;
            long long expected = 0, cycle = 0;
            for (int i = 0; i < s.num_ref_frames_in_poc_cycle; i++) cycle += s.offset_for_ref_frame[i];
            if (abs_fn > 0) {
                long long cnt = (abs_fn - 1) / s.num_ref_frames_in_poc_cycle; int in_cycle = (int)((abs_fn - 1) % s.num_ref_frames_in_poc_cycle);
                expected = cnt * cycle;
                for (int i = 0; i <= in_cycle; i++) expected += s.offset_for_ref_frame[i];
            }
            if (!sh.nal_ref_idc) expected += s.offset_for_non_ref_pic;
            // 8.2.1.2: a bottom FIELD picture: expected + offset_for_top_to_bottom_field + delta_pic_order_cnt[0]
            top = expected + sh.delta_poc[0];
            bot = sh.field_pic ? expected + s.offset_for_top_to_bottom + sh.delta_poc[0] : top + s.offset_for_top_to_bottom + sh.delta_poc[1];
        }
    }
    cur_top_poc_ = top; cur_bot_poc_ = bot;
    if (!sh.field_pic) { store.fpoc[0] = (int)top; store.fpoc[1] = (int)bot; return (int)std::min(top, bot); }
    store.fpoc[sh.bottom_field] = (int)(sh.bottom_field ? bot : top);
    return store.fpoc[sh.bottom_field];
}

void Decoder::flush_dpb(std::vector<int> &out) {
    if (codec_ == 1) { for (int i = 0; i < n_surf_; i++) if (i != cur_) dpb_[i].ref = 0; hevc_bump(out, true, true);
        for (int i = 0; i < n_surf_; i++) if (i != cur_ && !dpb_[i].wait_output) dpb_[i].in_use = false; return; }
    if (pending_first_ >= 0) { const int pf = pending_first_; pending_first_ = -1; dpb_[pf].waiting_second = false; store_done(pf, out); }
    for (int i = 0; i < n_surf_; i++) if (i != cur_) dpb_[i].set_ref(0);
    for (;;) {
        int best = -1;
        for (int i = 0; i < n_surf_; i++) if (i != cur_ && dpb_[i].in_use && dpb_[i].wait_output && (best < 0 || dpb_[i].poc < dpb_[best].poc)) best = i;
        if (best < 0) break;
        out.push_back(best | dpb_[best].lone << 8); dpb_[best].wait_output = false; display_pocs_.push_back(dpb_[best].poc);
        dpb_[best].out_at = decode_count_ - 1;
    }
    for (int i = 0; i < n_surf_; i++) if (i != cur_ && dpb_[i].in_use && !dpb_[i].ref && !dpb_[i].wait_output) dpb_[i].in_use = false;
}

// 8.2.5.2: a frame_num that no picture carried stands for a frame that was not sent (gaps_in_frame_num_value_allowed_flag; otherwise lost pictures, counted
// as an error).  The frame is inferred: the sliding window runs as for any reference frame without marking operations, and the frame stays in the buffer as
// a short-term reference "non-existing" -- it is in the reference lists, never displayed, and nothing may predict from it.  The order-count state of types
// 1 and 2 follows the frame numbers.
void Decoder::infer_frame(int fn) {
    const int max_fn = 1 << seq_.log2_max_frame_num;
    int nst = 0, nlt = 0, oldest = -1;
    for (int i = 0; i < n_surf_; i++) {
        DpbPic &p = dpb_[i];
        if (!p.in_use) continue;
        p.frame_num_wrap = p.frame_num > fn ? p.frame_num - max_fn : p.frame_num;
        if (p.any_short()) { nst++; if (oldest < 0 || p.frame_num_wrap < dpb_[oldest].frame_num_wrap) oldest = i; }
        else if (p.any_long()) nlt++;
    }
    if (nst + nlt >= std::max(seq_.max_num_ref_frames, 1) && oldest >= 0) dpb_[oldest].set_ref(0);
    int slot = -1;
    for (;;) {
        for (int i = 0; i < n_surf_; i++) if (dpb_[i].in_use && i != pending_first_ && !dpb_[i].ref && !dpb_[i].wait_output) dpb_[i].in_use = false;
        for (int k = 1; k <= n_surf_ && slot < 0; k++) { const int i = (last_surf_ + k) % n_surf_; if (!dpb_[i].in_use) slot = i; }
        if (slot >= 0) break;
        int best = -1;                                      // C.4.5.3: no empty frame buffer -> the picture first in output order goes
        for (int i = 0; i < n_surf_; i++) if (dpb_[i].wait_output && !dpb_[i].waiting_second && (best < 0 || dpb_[i].poc < dpb_[best].poc)) best = i;
        if (best < 0) { stat_errors_++; return; }
        carry_out_.push_back(best | dpb_[best].lone << 8); display_pocs_.push_back(dpb_[best].poc); dpb_[best].wait_output = false;
        dpb_[best].out_at = decode_count_ - 1;
    }
    DpbPic &c = dpb_[slot];
    c = DpbPic(); c.in_use = true; c.set_ref(1); c.frame_num = fn; c.decode_idx = decode_count_++; c.have = 3; c.non_existing = true;
    if (seq_.poc_type != 0) {
        // 8.2.5.2 (second reading, round 4): with pic_order_cnt_type 1 / 2 the inferred frame GETS order counts -- 8.2.1 as for a reference frame whose
        // delta_pic_order_cnt[] are 0 -- and takes its place by them in the initial lists of later B slices (rounds 1-3 left them 0; only the
        // FrameNumOffset state was carried on, which is all a P-only stream can see).  compute_poc also moves prevFrameNumOffset.
        SliceHeader ih{}; ih.idr = false; ih.nal_ref_idc = 1; ih.frame_num = fn; ih.field_pic = false; ih.bottom_field = false;
        ih.delta_poc[0] = ih.delta_poc[1] = 0;
        c.poc = compute_poc(ih, c);
    }
    prev_frame_num_ = fn; prev_mmco5_ = false;
    prev_ref_frame_num_ = fn;
    stat_inferred_frames_++;
}

bool Decoder::start_picture(const SliceHeader &sh, const SeqParams &sps, const PicParamSet &pps) {
    // 3.30 / 7.4.1.2.4: is this the SECOND field of the frame whose first field was the picture before it?  Opposite parity, the same frame_num, not an
    // IDR picture, and a reference field exactly if the first one is.  A first field that does not get its partner is complete as it stands.
    const int pf = pending_first_;
    const bool second = pf >= 0 && sh.field_pic && !sh.idr && dpb_[pf].waiting_second && dpb_[pf].frame_num == sh.frame_num &&
                        dpb_[pf].have == (sh.bottom_field ? 1 : 2) && dpb_[pf].first_was_ref == (sh.nal_ref_idc != 0) && seq_active_ &&
                        sps.mb_w == mb_w_ && sps.mb_h == mb_h_;
    if (pf >= 0 && !second) { pending_first_ = -1; dpb_[pf].waiting_second = false; stat_lone_fields_++; store_done(pf, carry_out_); }
    if (sh.idr || !seq_active_) {
        // pfnSequenceCallback moment (nv_dec.cpp:23-30): a new coded video sequence starts.
        if (seq_active_) flush_dpb(carry_out_);
        if (!activate(sps)) return false;
    } else if (sps.mb_w != mb_w_ || sps.mb_h != mb_h_) { stat_errors_++; return false; }
    if (!second && !sh.idr && decode_count_ > 0) {
        // 7.4.3: frame_num is PrevRefFrameNum or the one after it; anything else is a gap (a stream that STARTS without an IDR picture has no gap yet)
        const int max_fn = 1 << seq_.log2_max_frame_num;
        if (sh.frame_num != prev_ref_frame_num_ && sh.frame_num != (prev_ref_frame_num_ + 1) % max_fn) {
            if (!seq_.gaps_allowed) stat_errors_++;
            int guard = 0;
            for (int fn = (prev_ref_frame_num_ + 1) % max_fn; fn != sh.frame_num && guard < max_fn; fn = (fn + 1) % max_fn, guard++) infer_frame(fn);
        }
    }
    int slot = -1;
    bool wait_pack = false;
    if (second) slot = pf;
    else {
        // Surface for the new picture.  A surface that was displayed after picture n is still being packed out while picture
        // n+1 decodes (the engine overlaps pack-out with the next batch), so prefer one that has "cooled" for a picture.
        int warm = -1;
        // Round robin over the free surfaces (starting behind the one chosen last), so that a surface is reused as LATE as possible: the engine
        // runs consecutive pictures of a stream in one launch only while none of them decodes into a surface an earlier one still reads or displays.
        for (int k = 1; k <= n_surf_; k++) { const int i = (last_surf_ + k) % n_surf_; if (!dpb_[i].in_use) { if (decode_count_ >= dpb_[i].out_at + 2) {
            slot = i;
            break; } if (warm < 0) warm = i; } }
        if (slot < 0 && warm >= 0) { slot = warm; wait_pack = true; }
        while (slot < 0) {
            // C.4.5.3: no empty frame buffer -> the pictures first in output order go until one is free (frames inferred from gaps in frame_num take buffers
            // without passing through the output process, so this happens in conforming streams too)
            int best = -1;
            for (int i = 0; i < n_surf_; i++) if (dpb_[i].wait_output && !dpb_[i].waiting_second && (best < 0 || dpb_[i].poc < dpb_[best].poc)) best = i;
            if (best < 0) {                  // nothing left to display, every buffer a reference: non-conformant stream -- the oldest reference goes
                for (int i = 0; i < n_surf_; i++) if (best < 0 || dpb_[i].frame_num_wrap < dpb_[best].frame_num_wrap) best = i;
                dpb_[best].set_ref(0); dpb_[best].in_use = false; slot = best; stat_errors_++;
                break;
            }
            carry_out_.push_back(best | dpb_[best].lone << 8); display_pocs_.push_back(dpb_[best].poc); dpb_[best].wait_output = false;
            dpb_[best].out_at = decode_count_ - 1;
            if (!dpb_[best].ref) { dpb_[best].in_use = false; slot = best; wait_pack = true; }
        }
        last_surf_ = slot;
        DpbPic &c = dpb_[slot];
        c = DpbPic(); c.in_use = true; c.frame_num = sh.frame_num; c.decode_idx = decode_count_++;
        c.first_was_ref = sh.nal_ref_idc != 0; c.coded_as_fields = sh.field_pic;
        if (sps.profile_idc != 66 && sh.nal_ref_idc && !sh.field_pic) c.mf = std::make_shared<MotionField>();   // a later B picture may use this one as its
                                                                                                              // colocated picture
    }
    cur_ = slot;
    DpbPic &c = dpb_[slot];
    const int poc = compute_poc(sh, c);
    if (!second) c.poc = poc;
    cur_field_ = sh.field_pic ? 1 + (int)sh.bottom_field : 0; cur_second_ = second;
    pending_ = std::make_unique<PicTask>();
    if (sh.field_pic && seq_.profile_idc != 66 && sh.nal_ref_idc) c.mf_fld[sh.bottom_field] = std::make_shared<MotionField>();
    pending_->mf = sh.field_pic ? c.mf_fld[sh.bottom_field] : c.mf;
    pending_->has_picture = true; pending_->cur_slot = slot; pending_->wait_prev_pack = wait_pack; pending_->sps = sps; pending_->pps = pps;
    pending_->field = cur_field_;
    if (sh.field_pic) { pending_->sps.mb_h = sps.mb_h / 2; stat_field_pics_++; }      // from here on the picture is one of half the height
    pending_->out_before = std::move(carry_out_); carry_out_.clear();
    first_sh_ = sh;
    if (sh.type == SL_I) stat_i_++; else if (sh.type == SL_B) stat_b_++; else stat_p_++;
    return true;
}

// 8.2.4.2.5: the fields of an ordered list of frame stores, alternating in parity and beginning with the parity of the current field; a store whose field
// of the wanted parity is not marked `mark` is passed over, and when one parity has run out the rest of the other follows.  Entries: slot | parity << 5.
static int alternate_fields(const DpbPic *dpb, const int *stores, int n, int mark, int par, int cur_slot, int *out, int cnt, int max) {
    auto marked = [&](int i, int q) { return !(stores[i] == cur_slot && q == par) && dpb[stores[i]].fmark[q] == mark; };
    int c[2] = {0, 0}, q = par;
    for (;;) {
        while (c[q] < n && !marked(c[q], q)) c[q]++;
        if (c[q] < n) { if (cnt < max) out[cnt++] = stores[c[q]] | q << 5; c[q]++; }
        else { const int o = q ^ 1; while (c[o] < n && !marked(c[o], o)) c[o]++; if (c[o] >= n) break; }
        q ^= 1;
    }
    return cnt;
}

// 8.2.4.1 (field picture numbers), 8.2.4.2.2 + 8.2.4.2.5, 8.2.4.3 for the P slices of a field picture.  A list entry names a FIELD: the surface slot
// of its frame store with the parity in bit 5 (jobs.h MbRec.ref; kernels take the lines of that parity).
void Decoder::build_field_ref_lists(const SliceHeader &sh, SliceTask &task) {
    SliceRefs &rf = task.refs;
    const int par = sh.bottom_field, max_fn = 1 << seq_.log2_max_frame_num;
    rf.cur_poc = dpb_[cur_].fpoc[par];
    int st[kMaxSurfaces], lt[kMaxSurfaces], nst = 0, nlt = 0;
    for (int i = 0; i < n_surf_; i++) {
        DpbPic &p = dpb_[i];
        if (!p.in_use) continue;
        p.frame_num_wrap = p.frame_num > sh.frame_num ? p.frame_num - max_fn : p.frame_num;
        // the store of the current frame takes part with its FIRST field (8.2.4.2.2)
        const int m0 = (i == cur_ && par == 0) ? 0 : p.fmark[0], m1 = (i == cur_ && par == 1) ? 0 : p.fmark[1];
        if (m0 == 1 || m1 == 1) st[nst++] = i;
        if (m0 == 2 || m1 == 2) lt[nlt++] = i;
    }
    std::sort(st, st + nst, [&](int a, int b) { return dpb_[a].frame_num_wrap > dpb_[b].frame_num_wrap; });
    std::sort(lt, lt + nlt, [&](int a, int b) { return dpb_[a].lt_idx < dpb_[b].lt_idx; });
    const int nlists = sh.type == SL_B ? 2 : 1;
    int lists[2][35], ninit[2] = {0, 0};
    for (auto &row : lists) for (int &v : row) v = -1;
    if (nlists == 1) ninit[0] = alternate_fields(dpb_, st, nst, 1, par, cur_, lists[0], 0, 33);
    else {
        // 8.2.4.2.4: short-term stores by PicOrderCnt around the count of the current FIELD -- list 0: those not above it, descending, then the others
        // ascending; list 1 the other way round.  PicOrderCnt of a store: of the frame / field pair (Min of its fields) or of its only field (DpbPic.poc)
        int ord[2][kMaxSurfaces], nb = 0, na = 0, before[kMaxSurfaces], after[kMaxSurfaces];
        // (pic_order_cnt_type 0: a frame inferred from a gap in frame_num has no order count and is left out, as in the frame lists of 8.2.4.2.3)
        for (int i = 0; i < nst; i++) { if (dpb_[st[i]].non_existing && seq_.poc_type == 0) continue;
            if (dpb_[st[i]].poc <= rf.cur_poc) before[nb++] = st[i]; else after[na++] = st[i]; }
        std::sort(before, before + nb, [&](int a, int b) { return dpb_[a].poc > dpb_[b].poc; });
        std::sort(after, after + na, [&](int a, int b) { return dpb_[a].poc < dpb_[b].poc; });
        for (int i = 0; i < nb; i++) { ord[0][i] = before[i]; ord[1][na + i] = before[i]; }
        for (int i = 0; i < na; i++) { ord[0][nb + i] = after[i]; ord[1][i] = after[i]; }
        for (int l = 0; l < 2; l++) ninit[l] = alternate_fields(dpb_, ord[l], nst, 1, par, cur_, lists[l], 0, 33);
    }
    for (int l = 0; l < nlists; l++) ninit[l] = alternate_fields(dpb_, lt, nlt, 2, par, cur_, lists[l], ninit[l], 33);
    if (nlists == 2 && ninit[1] > 1 && ninit[0] == ninit[1] && std::equal(lists[0], lists[0] + ninit[0], lists[1])) std::swap(lists[1][0], lists[1][1]);
    // PicNum = 2 * FrameNumWrap + 1 for a field of the current parity, 2 * FrameNumWrap for the other; LongTermPicNum the same from LongTermFrameIdx
    auto pic_num = [&](int e) { return 2 * dpb_[e & 31].frame_num_wrap + (((e >> 5) & 1) == par); };
    auto lt_pic_num = [&](int e) { return 2 * dpb_[e & 31].lt_idx + (((e >> 5) & 1) == par); };
    const int cur_pic_num = 2 * sh.frame_num + 1, max_pic_num = 2 * max_fn;
    for (int l = 0; l < nlists; l++) {
        int *list = lists[l];
        const int nact = sh.num_ref_idx[l];
        for (int i = nact; i < 35; i++) list[i] = -1;
        int pred = cur_pic_num, idx = 0;
        for (int k = 0; k < sh.n_mod[l]; k++) {
            const RefMod &m = sh.mod[l][k];
            int target = -1;
            if (m.idc < 2) {
                int nowrap = m.idc == 0 ? pred - (int)(m.val + 1) : pred + (int)(m.val + 1);
                if (nowrap < 0) nowrap += max_pic_num;
                if (nowrap >= max_pic_num) nowrap -= max_pic_num;
                pred = nowrap;
                const int want = nowrap > cur_pic_num ? nowrap - max_pic_num : nowrap;
                for (int i = 0; i < nst; i++) for (int q = 0; q < 2; q++) { const int e = st[i] | q << 5;
                    if (!(st[i] == cur_ && q == par) && dpb_[st[i]].fmark[q] == 1 && pic_num(e) == want) target = e; }
            } else for (int i = 0; i < nlt; i++) for (int q = 0; q < 2; q++) { const int e = lt[i] | q << 5;
                if (!(lt[i] == cur_ && q == par) && dpb_[lt[i]].fmark[q] == 2 && lt_pic_num(e) == (int)m.val) target = e; }
            if (target < 0 || idx >= nact) { stat_errors_++; break; }
            for (int c = nact; c > idx; c--) list[c] = list[c - 1];
            list[idx++] = target;
            int nidx = idx;
            for (int c = idx; c <= nact; c++) if (list[c] != target) list[nidx++] = list[c];
        }
        for (int i = 0; i < nact && i < 32; i++) {
            rf.slot[l][i] = (int8_t)list[i];
            if (list[i] >= 0) { const DpbPic &p = dpb_[list[i] & 31]; const int q = (list[i] >> 5) & 1;
                rf.uid[l][i] = 2 * p.decode_idx + q; rf.poc[l][i] = p.fpoc[q]; rf.is_long[l][i] = p.fmark[q] == 2; }
        }
    }
    if (nlists == 2 && rf.slot[1][0] >= 0) {
        // 8.4.1.2.1: the colocated field's motion -- its own when it was coded as a field picture, otherwise that of the FRAME picture (Frm_To_Fld)
        const DpbPic &c = dpb_[rf.slot[1][0] & 31];
        if (c.coded_as_fields) task.col = c.mf_fld[(rf.slot[1][0] >> 5) & 1];
        else { task.col = c.mf; rf.col_mode = 1; rf.cur_parity = par; }
        rf.col = task.col.get();
    }
}

// 8.2.4.2 + 8.2.4.3: RefPicList0 / RefPicList1 of a slice, expressed as surface slots (+ what direct / weighted prediction need)
void Decoder::build_ref_lists(const SliceHeader &sh, SliceTask &task) {
    SliceRefs &rf = task.refs;
    memset(rf.slot, -1, sizeof rf.slot); memset(rf.uid, -1, sizeof rf.uid); memset(rf.poc, 0, sizeof rf.poc); memset(rf.is_long, 0, sizeof rf.is_long);
    rf.cur_poc = dpb_[cur_].poc;
    rf.track_uid = seq_.profile_idc != 66;                   // Baseline has no B slices: nothing will ever ask for this picture's motion
    rf.bipred_rec = sh.type == SL_B || sh.explicit_wp;
    if (sh.type == SL_I) return;
    if (sh.field_pic) build_field_ref_lists(sh, task);
    else build_frame_ref_lists(sh, task);
    // weighted prediction tables (8.4.2.3)
    const PicParamSet &pps = ps_.pps[sh.pps_id];
    const int nlists = sh.type == SL_B ? 2 : 1;
    int mode = sh.type == SL_P ? (sh.explicit_wp ? 1 : 0) : pps.weighted_bipred_idc;
    task.has_wp = mode != 0;
    if (task.has_wp) {
        if (mode == 1 && (sh.num_ref_idx[0] > 16 || sh.num_ref_idx[1] > 16)) { stat_errors_++;
            fail("explicit weighted prediction with more than 16 list entries"); }
        SliceWp &wp = task.wp;
        memset(&wp, 0, sizeof wp);
        wp.mode = (uint8_t)mode; wp.logwd_y = (uint8_t)sh.luma_log2_wd; wp.logwd_c = (uint8_t)sh.chroma_log2_wd;
        if (mode == 1) {
            for (int l = 0; l < nlists; l++) for (int i = 0; i < 16 && i < sh.num_ref_idx[l]; i++) {
                wp.w[l][i][0] = (int8_t)sh.luma_w[l][i]; wp.o[l][i][0] = (int8_t)sh.luma_o[l][i];
                for (int c = 0; c < 2; c++) { wp.w[l][i][1 + c] = (int8_t)sh.chroma_w[l][i][c]; wp.o[l][i][1 + c] = (int8_t)sh.chroma_o[l][i][c]; }
            }
        } else {
            for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
                int w1 = 32;
                if (rf.slot[0][i] >= 0 && rf.slot[1][j] >= 0 && !rf.is_long[0][i] && !rf.is_long[1][j]) {
                    int tb = std::clamp(rf.cur_poc - rf.poc[0][i], -128, 127), td = std::clamp(rf.poc[1][j] - rf.poc[0][i], -128, 127);
                    if (td != 0) { int tx = (16384 + std::abs(td / 2)) / td, dsf = std::clamp((tb * tx + 32) >> 6, -1024, 1023) >> 2;
                        if (dsf >= -64 && dsf <= 128) w1 = dsf; }
                }
                wp.imp_w1[i][j] = (uint8_t)(64 + w1);
            }
        }
    }
}

// 8.2.4.2.1 / 8.2.4.2.3 + 8.2.4.3 for the slices of a FRAME picture: reference frames are the stores with both fields marked (DpbPic.ref 1 / 2)
void Decoder::build_frame_ref_lists(const SliceHeader &sh, SliceTask &task) {
    SliceRefs &rf = task.refs;
    int max_fn = 1 << seq_.log2_max_frame_num;
    int st[kMaxSurfaces], lt[kMaxSurfaces], nst = 0, nlt = 0;
    for (int i = 0; i < n_surf_; i++) {
        DpbPic &p = dpb_[i];
        if (!p.in_use || i == cur_) continue;
        if (p.ref == 1) { p.frame_num_wrap = p.frame_num > sh.frame_num ? p.frame_num - max_fn : p.frame_num; p.pic_num = p.frame_num_wrap; st[nst++] = i; }
        else if (p.ref == 2) lt[nlt++] = i;
    }
    std::sort(lt, lt + nlt, [&](int a, int b) { return dpb_[a].lt_idx < dpb_[b].lt_idx; });
    const int nlists = sh.type == SL_B ? 2 : 1;
    int init[2][34], ninit[2] = {0, 0};
    for (auto &row : init) for (int &v : row) v = -1;
    if (sh.type == SL_P) {
        std::sort(st, st + nst, [&](int a, int b) { return dpb_[a].pic_num > dpb_[b].pic_num; });
        for (int i = 0; i < nst && ninit[0] < 33; i++) init[0][ninit[0]++] = st[i];
    } else {                                                // 8.2.4.2.3: by POC around the current picture
        int before[kMaxSurfaces], after[kMaxSurfaces], nb = 0, na = 0, cp = dpb_[cur_].poc;
        // (with pic_order_cnt_type 0 a frame inferred from a gap in frame_num has no order count: it is left out of the initial lists of a B slice;
        //  with types 1 / 2 it has one, infer_frame)
        for (int i = 0; i < nst; i++) { if (dpb_[st[i]].non_existing && seq_.poc_type == 0) continue;
            if (dpb_[st[i]].poc < cp) before[nb++] = st[i]; else after[na++] = st[i]; }
        std::sort(before, before + nb, [&](int a, int b) { return dpb_[a].poc > dpb_[b].poc; });
        std::sort(after, after + na, [&](int a, int b) { return dpb_[a].poc < dpb_[b].poc; });
        for (int i = 0; i < nb; i++) init[0][ninit[0]++] = before[i];
        for (int i = 0; i < na; i++) init[0][ninit[0]++] = after[i];
        for (int i = 0; i < na; i++) init[1][ninit[1]++] = after[i];
        for (int i = 0; i < nb; i++) init[1][ninit[1]++] = before[i];
    }
    for (int l = 0; l < nlists; l++) for (int i = 0; i < nlt && ninit[l] < 33; i++) init[l][ninit[l]++] = lt[i];
    if (nlists == 2 && ninit[1] > 1 && ninit[0] == ninit[1] && std::equal(init[0], init[0] + ninit[0], init[1])) std::swap(init[1][0], init[1][1]);
    if (sh.type == SL_P) std::sort(st, st + nst, [&](int a, int b) { return dpb_[a].pic_num > dpb_[b].pic_num; });
    for (int l = 0; l < nlists; l++) {
        int *list = init[l];
        int nact = sh.num_ref_idx[l];
        for (int i = nact; i < 34; i++) list[i] = -1;
        int pred = sh.frame_num, idx = 0;
        for (int k = 0; k < sh.n_mod[l]; k++) {
            const RefMod &m = sh.mod[l][k];
            int target = -1;
            if (m.idc < 2) {
                int nowrap = m.idc == 0 ? pred - (int)(m.val + 1) : pred + (int)(m.val + 1);
                if (nowrap < 0) nowrap += max_fn;
                if (nowrap >= max_fn) nowrap -= max_fn;
                pred = nowrap;
                int pic_num = nowrap > sh.frame_num ? nowrap - max_fn : nowrap;
                for (int i = 0; i < nst; i++) if (dpb_[st[i]].pic_num == pic_num) target = st[i];
            } else for (int i = 0; i < nlt; i++) if (dpb_[lt[i]].lt_idx == (int)m.val) target = lt[i];
            if (target < 0 || idx >= nact) { stat_errors_++; break; }
            for (int c = nact; c > idx; c--) list[c] = list[c - 1];
            list[idx++] = target;
            int nidx = idx;
            for (int c = idx; c <= nact; c++) if (list[c] != target) list[nidx++] = list[c];
        }
        for (int i = 0; i < nact && i < 32; i++) {
            rf.slot[l][i] = (int8_t)list[i];
            if (list[i] >= 0) { rf.uid[l][i] = 2 * dpb_[list[i]].decode_idx; rf.poc[l][i] = dpb_[list[i]].poc; rf.is_long[l][i] = dpb_[list[i]].ref == 2; }
        }
    }
    if (sh.type == SL_B && rf.slot[1][0] >= 0) {
        const DpbPic &c = dpb_[rf.slot[1][0]];
        if (c.coded_as_fields) {
            // 8.4.1.2.1: RefPicList1[0] is a complementary field pair: the motion of the field nearer in order count (the top field only when strictly nearer)
            const int q = std::abs(c.fpoc[0] - rf.cur_poc) < std::abs(c.fpoc[1] - rf.cur_poc) ? 0 : 1;
            task.col = c.mf_fld[q]; rf.col_mode = 2;
            if (c.have != 3 || !task.col) { stat_errors_++; note_error("colocated field pair incomplete"); task.col.reset(); }
        } else task.col = c.mf;
        rf.col = task.col.get();
    }
}

void Decoder::add_slice(const SliceHeader &sh, std::vector<uint8_t> &&rbsp, size_t rbsp_len) {
    if (pending_->slices.size() >= 255) { stat_errors_++; return; }
    pending_->slices.emplace_back();
    SliceTask &s = pending_->slices.back();
    s.sh = sh; s.rbsp = std::move(rbsp); s.rbsp_len = rbsp_len;
    build_ref_lists(sh, s);
}

void Decoder::mark_current(const SliceHeader &sh) {                                 // 8.2.5, frame picture
    DpbPic &cur = dpb_[cur_];
    if (!sh.nal_ref_idc) return;
    int max_fn = 1 << seq_.log2_max_frame_num;
    if (sh.idr) {
        for (int i = 0; i < n_surf_; i++) if (i != cur_) dpb_[i].set_ref(0);
        if (sh.long_term_reference) { cur.set_ref(2); cur.lt_idx = 0; max_lt_idx_ = 0; } else { cur.set_ref(1); max_lt_idx_ = -1; }
        return;
    }
    for (int i = 0; i < n_surf_; i++) { DpbPic &p = dpb_[i]; if (p.in_use && i != cur_ && p.any_short()) {
        p.frame_num_wrap = p.frame_num > sh.frame_num ? p.frame_num - max_fn : p.frame_num; p.pic_num = p.frame_num_wrap; } }
    bool made_long = false;
    if (sh.adaptive_marking) {
        // a frame picture names FRAMES: stores with both fields marked the same way (ref 1 / 2; 8.2.4.1)
        for (int k = 0; k < sh.n_mark; k++) {
            const MarkOp &m = sh.mark[k];
            int pic_num_x = sh.frame_num - (int)(m.a + 1);
            for (int i = 0; i < n_surf_; i++) {
                DpbPic &p = dpb_[i];
                if (!p.in_use || i == cur_) continue;
                switch (m.op) {
                case 1: if (p.ref == 1 && p.pic_num == pic_num_x) p.set_ref(0); break;
                case 2: if (p.ref == 2 && p.lt_idx == (int)m.a) p.set_ref(0); break;
                case 3: if (p.ref == 2 && p.lt_idx == (int)m.b) p.set_ref(0); break;
                case 4: if (p.any_long() && p.lt_idx > (int)m.a - 1) p.set_ref(0); break;
                case 5: p.set_ref(0); break;
                case 6: if (p.ref == 2 && p.lt_idx == (int)m.b) p.set_ref(0); break;
                }
            }
            if (m.op == 3) for (int i = 0; i < n_surf_; i++) { DpbPic &p = dpb_[i]; if (p.in_use && i != cur_ && p.ref == 1 && p.pic_num == pic_num_x) {
                p.set_ref(2); p.lt_idx = (int)m.b; } }
            if (m.op == 4) max_lt_idx_ = (int)m.a - 1;
            if (m.op == 5) { max_lt_idx_ = -1; cur.mmco5 = true; }
            if (m.op == 6) { cur.set_ref(2); cur.lt_idx = (int)m.b; made_long = true; }
        }
    } else {
        // 8.2.5.3: the count is of frames, complementary field pairs and single fields in which ANY field is marked
        int nst = 0, nlt = 0, oldest = -1;
        for (int i = 0; i < n_surf_; i++) {
            DpbPic &p = dpb_[i];
            if (!p.in_use || i == cur_) continue;
            if (p.any_short()) { nst++; if (oldest < 0 || p.frame_num_wrap < dpb_[oldest].frame_num_wrap) oldest = i; }
            else if (p.any_long()) nlt++;
        }
        if (nst + nlt >= std::max(seq_.max_num_ref_frames, 1) && oldest >= 0) dpb_[oldest].set_ref(0);
    }
    if (!made_long) cur.set_ref(1);
}

// 8.2.5 for a FIELD picture: operations name fields by their field picture numbers (8.2.4.1); the sliding window leaves the second field of a reference
// frame alone (8.2.5.3).  Operation 5 in a field picture is not supported: the handle fails.
void Decoder::mark_current_field(const SliceHeader &sh) {
    DpbPic &cur = dpb_[cur_];
    const int par = sh.bottom_field, max_fn = 1 << seq_.log2_max_frame_num;
    if (!sh.nal_ref_idc) return;
    if (sh.idr) {
        for (int i = 0; i < n_surf_; i++) if (i != cur_) dpb_[i].set_ref(0);
        cur.fmark[par] = sh.long_term_reference ? 2 : 1; cur.fmark[par ^ 1] = 0; cur.sync_ref();
        cur.lt_idx = sh.long_term_reference ? 0 : -1; max_lt_idx_ = sh.long_term_reference ? 0 : -1;
        return;
    }
    for (int i = 0; i < n_surf_; i++) { DpbPic &p = dpb_[i]; if (p.in_use) p.frame_num_wrap = p.frame_num > sh.frame_num ? p.frame_num - max_fn : p.frame_num; }
    bool made_long = false;
    if (sh.adaptive_marking) {
        const int cur_pic_num = 2 * sh.frame_num + 1;
        for (int k = 0; k < sh.n_mark; k++) {
            const MarkOp &m = sh.mark[k];
            if (m.op == 5) { stat_errors_++; fail("memory management operation 5 in a field picture is not supported"); return; }
            const int pic_num_x = cur_pic_num - (int)(m.a + 1);
            if (m.op == 3 || m.op == 6) {
                // 8.2.5.4.3 / 8.2.5.4.6: a field -- the one picNumX names, or the current one -- becomes long-term with LongTermFrameIdx m.b.  Whoever holds
                // that index loses it first, except the other field of the SAME frame (the two fields of a long-term pair share their index)
                int owner = m.op == 6 ? cur_ : -1, owner_q = par;
                if (m.op == 3) for (int i = 0; i < n_surf_; i++) { const DpbPic &p = dpb_[i]; if (!p.in_use) continue;
                    for (int q = 0; q < 2; q++) if (!(i == cur_ && q == par) && p.fmark[q] == 1 && 2 * p.frame_num_wrap + (q == par) == pic_num_x) {
                        owner = i; owner_q = q; } }
                if (owner < 0) { stat_errors_++; continue; }
                for (int i = 0; i < n_surf_; i++) { DpbPic &p = dpb_[i];
                    if (p.in_use && i != owner && p.lt_idx == (int)m.b) { for (int q = 0; q < 2; q++) if (p.fmark[q] == 2) p.fmark[q] = 0; p.sync_ref(); } }
                dpb_[owner].lt_idx = (int)m.b;
                if (m.op == 3) { dpb_[owner].fmark[owner_q] = 2; dpb_[owner].sync_ref(); } else made_long = true;
                continue;
            }
            for (int i = 0; i < n_surf_; i++) {
                DpbPic &p = dpb_[i];
                if (!p.in_use) continue;
                for (int q = 0; q < 2; q++) {
                    if (i == cur_ && q == par) continue;
                    const int same = q == par;
                    if (m.op == 1 && p.fmark[q] == 1 && 2 * p.frame_num_wrap + same == pic_num_x) p.fmark[q] = 0;
                    if (m.op == 2 && p.fmark[q] == 2 && 2 * p.lt_idx + same == (int)m.a) p.fmark[q] = 0;
                    if (m.op == 4 && p.fmark[q] == 2 && p.lt_idx > (int)m.a - 1) p.fmark[q] = 0;
                }
                p.sync_ref();
            }
            if (m.op == 4) max_lt_idx_ = (int)m.a - 1;
        }
    } else if (!(cur_second_ && cur.fmark[par ^ 1] == 1)) {
        int nst = 0, nlt = 0, oldest = -1;
        for (int i = 0; i < n_surf_; i++) {
            DpbPic &p = dpb_[i];
            if (!p.in_use || i == cur_) continue;
            if (p.any_short()) { nst++; if (oldest < 0 || p.frame_num_wrap < dpb_[oldest].frame_num_wrap) oldest = i; }
            else if (p.any_long()) nlt++;
        }
        if (nst + nlt >= std::max(seq_.max_num_ref_frames, 1) && oldest >= 0) dpb_[oldest].set_ref(0);
    }
    cur.fmark[par] = made_long ? 2 : 1; cur.sync_ref();
}

// C.4.5.2 / C.4.5.3 for a frame store that is complete -- a frame, both fields of a frame, or a field whose partner did not come --, with the display
// delay of the reference replaced by the minimum that keeps display order (the YUV file only records ORDER; nv_dec.cpp:341 ulMaxDisplayDelay=2
// only adds latency)
void Decoder::store_done(int slot, std::vector<int> &out) {
    DpbPic &cur = dpb_[slot];
    auto smallest = [&](int exclude) {
        int b = -1;
        for (int i = 0; i < n_surf_; i++)
            if (i != exclude && dpb_[i].in_use && dpb_[i].wait_output && !dpb_[i].waiting_second && (b < 0 || dpb_[i].poc < dpb_[b].poc)) b = i;
        return b;
    };
    if (cur.mmco5) {
        int b;
        while ((b = smallest(slot)) >= 0) {
            out.push_back(b | dpb_[b].lone << 8); display_pocs_.push_back(dpb_[b].poc); dpb_[b].wait_output = false;
            dpb_[b].out_at = decode_count_ - 1;
        }
        const int tmp = std::min(cur.fpoc[0], cur.fpoc[1]);
        cur.fpoc[0] -= tmp; cur.fpoc[1] -= tmp;
        cur.poc = 0; cur.frame_num = 0;
    }
    int w = smallest(slot);
    cur.lone = cur.have == 3 || codec_ == 1 ? 0 : cur.have;
    if (!cur.ref && (w < 0 || dpb_[w].poc > cur.poc)) { out.push_back(slot | cur.lone << 8); display_pocs_.push_back(cur.poc); cur.out_at = decode_count_ - 1;
        cur.in_use = false; cur.wait_output = false; }
    else {
        cur.wait_output = true;
        for (;;) {
            int used = 0, waiting = 0;
            for (int i = 0; i < n_surf_; i++) if (dpb_[i].in_use) { if (dpb_[i].ref || dpb_[i].wait_output) used++; if (dpb_[i].wait_output) waiting++; }
            if (used <= dpb_size_ && waiting <= reorder_depth_) break;
            int b = smallest(-1);
            if (b < 0) break;
            out.push_back(b | dpb_[b].lone << 8); display_pocs_.push_back(dpb_[b].poc); dpb_[b].wait_output = false; dpb_[b].out_at = decode_count_ - 1;
        }
    }
    for (int i = 0; i < n_surf_; i++) if (dpb_[i].in_use && i != pending_first_ && !dpb_[i].ref && !dpb_[i].wait_output) dpb_[i].in_use = false;
}

void Decoder::dispatch_pending() {
    if (codec_ == 1) { hevc_dispatch_pending(); return; }
    if (!pending_) return;
    std::unique_ptr<PicTask> t = std::move(pending_);
    DpbPic &cur = dpb_[cur_];
    const int slot = cur_;
    if (cur_field_) {
        const int par = cur_field_ - 1;
        mark_current_field(first_sh_);
        prev_frame_num_ = cur.frame_num; prev_mmco5_ = false;
        if (first_sh_.nal_ref_idc) prev_ref_frame_num_ = cur.frame_num;
        cur.have |= 1 << par;
        cur_ = -1;
        if (!cur_second_) {
            // the first field of a frame: the store waits for the other one (start_picture decides whether the next picture is that)
            cur.waiting_second = true; cur.wait_output = true; pending_first_ = slot;
        } else {
            cur.waiting_second = false; pending_first_ = -1;
            cur.poc = std::min(cur.fpoc[0], cur.fpoc[1]);                        // 8.2.1: PicOrderCnt of a complementary field pair
            store_done(slot, t->out_after);
        }
    } else {
        cur.have = 3;
        mark_current(first_sh_);
        prev_frame_num_ = cur.frame_num;
        if (first_sh_.nal_ref_idc) prev_ref_frame_num_ = cur.mmco5 ? 0 : cur.frame_num;
        prev_mmco5_ = cur.mmco5;                   // types 1 / 2: "the previous picture in decoding order included operation 5" (8.2.1.2, 8.2.1.3)
        // type 0: tempPicOrderCnt = Min(top, bottom) is subtracted from the picture's order counts (8.2.1); what the following pictures see as
        // prevPicOrderCntLsb is its TopFieldOrderCnt after that (> 0 when the bottom field lies below the top field)
        if (cur.mmco5 && seq_.poc_type == 0) { prev_poc_msb_ = 0; prev_poc_lsb_ = cur.fpoc[0] - std::min(cur.fpoc[0], cur.fpoc[1]); }
        cur_ = -1;
        store_done(slot, t->out_after);
    }
    t->job_slot = acquire_job_slot(first_sh_.type == SL_I);
    push_task(std::move(t));
}

// big: the picture is an I picture -- several times the job list of a P / B picture.  It borrows one of the handle's worst-case buffers for as long as
// it holds the slot, WHEN one is free.  It never waits for one (an all-intra or short-GOP stream had its pipeline depth cut from
// kJobSlots to kBigJobBufs that way, and the P / B pictures behind a waiting I picture could not be dispatched): without a free big buffer it takes
// the roomiest free ordinary slot, which parse_task grows on overflow -- straight to the size the handle's I pictures have shown so far (i_job_peak_) --
// and which stays that size; ordinary pictures take the SMALLEST free slot, so grown slots collect where the next I pictures find them.  An intra-only
// stream therefore settles into kJobSlots slots of I-picture size after a handful of grow events (stat job_regrown) instead of running three deep.
int Decoder::acquire_job_slot(bool big) {
    auto w0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(mtx_);
    const bool lend = big && codec_ == 0 && big_[0].host != nullptr;
    int got = -1, bg = -1;
    cv_.wait(lk, [&] {
        got = -1;
        for (int i = 0; i < n_jobs_; i++) if (!jobs_[i].busy && (got < 0 || (big ? jobs_[i].cap > jobs_[got].cap : jobs_[i].cap < jobs_[got].cap))) got = i;
        return got >= 0;
    });
    if (lend && jobs_[got].cap < job_cap_max_) for (int i = 0; i < kBigJobBufs && bg < 0; i++) if (!big_[i].busy) bg = i;
    if (bg >= 0) {
        JobSlot &j = jobs_[got];
        j.own_host = j.host; j.own_dev = j.dev; j.own_cap = j.cap;
        j.host = big_[bg].host; j.dev = big_[bg].dev; j.cap = job_cap_max_; j.big = bg; big_[bg].busy = true;
    }
    jobs_[got].busy = true;
    stat_wait_slot_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
    return got;
}

void Decoder::push_task(std::unique_ptr<PicTask> t) {
    PicTask *raw = t.get();
    // sync / digest mode: strictly one picture in flight (the end-of-stream flush can dispatch two pictures in one call)
    if (sync_mode_) { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0; }); }
    raw->t_dispatch = now_ns();
    {
        std::lock_guard<std::mutex> lk(mtx_);
        raw->seq = next_seq_++;
        outstanding_++;
        if (raw->has_picture) parse_pending_++;
        inflight_.push_back(std::move(t));
    }
    if (raw->has_picture) pool_submit(this, raw);
    else { raw->state.store(1, std::memory_order_release); submit_ready(); }
}

// =============================================================================================
// worker: entropy decode one picture into its job buffer
// =============================================================================================
void Decoder::parse_task(PicTask *t, ParseScratch &scratch) {
    if (t->hevc) { hevc_parse_task(t); return; }
    auto pt0 = std::chrono::steady_clock::now();
    JobSlot &js = jobs_[t->job_slot];
    const int n_mbs = t->sps.mb_w * t->sps.mb_h;
    scratch.resize(t->sps.mb_w, t->sps.mb_h);
    scratch.begin_picture();
    // job buffer layout: MbRec[n_mbs] | SliceRec[256] | coef ... | mv_ext (appended after parsing)
    static thread_local std::vector<int16_t> mv_ext_buf;
    const bool big_rec = t->sps.profile_idc != 66;
    mv_ext_buf.resize((size_t)n_mbs * (big_rec ? kBiRecInt16 : 32));
    const size_t fixed = (size_t)n_mbs * sizeof(MbRec) + 256 * sizeof(SliceRec);
    // The slot may be smaller than the worst-case picture (gpu_alloc_sequence).  Levels are written in place, so a picture that outgrows its slot is
    // parsed AGAIN into a bigger one (twice the size, at most the worst case): rare -- I pictures borrow worst-case buffers (acquire_job_slot).
    MbRec *mbs = nullptr; SliceRec *srec = nullptr; int16_t *coef = nullptr;
    MbRec blank; memset(&blank, 0, sizeof blank); blank.kind = MB_INTER;
    // default record = concealment for macroblocks no slice delivers (lost / damaged slices): copy the colocated macroblock of the
    // first list-0 reference (zero motion, no residual); grey when the picture has no reference (ref -1 -> 128 in k_recon_inter)
    { int8_t c = -1; for (auto &s : t->slices) if (s.sh.type != SL_I && s.refs.slot[0][0] >= 0) { c = s.refs.slot[0][0]; break; }
      blank.ref[0] = blank.ref[1] = blank.ref[2] = blank.ref[3] = c; }
    JobWriter w;
    SyntaxDigest dg = digest_;
    for (int attempt = 0;; attempt++) {
        scratch.begin_picture();
        mbs = (MbRec *)js.host; srec = (SliceRec *)(js.host + (size_t)n_mbs * sizeof(MbRec)); coef = (int16_t *)(srec + 256);
        w = JobWriter();
        w.mbs = mbs; w.mv_ext = mv_ext_buf.data(); w.mv_ext_cap = (uint32_t)n_mbs * (big_rec ? kBiRecInt16 / 2 : 16);
        w.coef = coef;
        {   // room for levels: what the slot holds behind the fixed records (signed: a buffer that cannot even hold those must fail the picture, not
            // wrap into a huge capacity).  The motion records and weight tables are appended after the parse and get their room then (below).
            long long room = (long long)js.cap - (long long)fixed - 64;
            if (room <= 0) { t->error = "job buffer too small for this picture"; stat_errors_++; fail(t->error); room = 0; }
            w.coef_cap = (uint32_t)(room / 2);
        }
        dg = digest_;
        t->n_intra = 0; t->n_i8x8 = 0; t->any_deblock = false; t->any_wp = false; t->error.clear();
        const char *first_error = nullptr; bool overflow = false;
        for (size_t si = 0; si < t->slices.size(); si++) {
            SliceTask &s = t->slices[si];
            srec[si].alpha_off = (int8_t)s.sh.alpha_off; srec[si].beta_off = (int8_t)s.sh.beta_off; srec[si].disable = (uint8_t)s.sh.disable_deblock;
            srec[si].pad = 0;
            if (s.sh.disable_deblock != 1) t->any_deblock = true;
            BitReader br(s.rbsp.data(), s.rbsp_len);
            br.set_end_from_trailing();
            br.skip_bytes(s.sh.data_bit_offset >> 3); br.skip((int)(s.sh.data_bit_offset & 7));
            if (s.sh.first_mb >= n_mbs) { t->error = "first_mb_in_slice out of range"; continue; }
            if (s.col) s.col->wait();                            // direct prediction reads RefPicList1[0]'s motion: that picture was dispatched earlier
            if (s.has_wp) t->any_wp = true;
            SliceParseResult r = parse_slice_data(t->sps, t->pps, s.sh, br, (int)si, s.refs, scratch, w, want_digest_ ? &dg : nullptr, fast_parse_ &&
                !t->field);
            t->n_intra += r.n_intra; t->n_i8x8 += r.n_i8x8;
            if (r.error) { t->error = r.error; if (!first_error) first_error = r.error; overflow |= strcmp(r.error, "coefficient buffer overflow") == 0; }
        }
        if (overflow && js.cap < job_cap_max_ && attempt < 6 && !failed_) {
            // twice the size -- or, for an I picture in an ordinary slot (no worst-case buffer was free), what this handle's I pictures needed so far
            const bool i_pic = !t->slices.empty() && t->slices[0].sh.type == SL_I;
            const size_t want = std::min(job_cap_max_, std::max(js.cap * 2, i_pic ? i_job_peak_.load() * 5 / 4 : (size_t)0));
            if (ensure_job_cap(js, want)) { stat_job_regrown_++; continue; }
        }
        if (!t->error.empty()) { stat_errors_++; note_error(std::string("slice data: ") + t->error); }
        break;
    }
    for (int i = 0; i < n_mbs; i++) if (scratch.slice_of[i] < 0) mbs[i] = blank;      // only what no slice covered (normally nothing)
    if (want_digest_) digest_ = dg;      // pictures are parsed in order when the digest is requested (sync option)
    if (const char *dm = getenv("JM_AMD_DEC_DUMP_MB")) {      // developer aid: "picture:x:y" prints the records of a macroblock and its neighbours
        int dp = 0, dx = 0, dy = 0;
        if (sscanf(dm, "%d:%d:%d", &dp, &dx, &dy) == 3 && (dp < 0 || dp == (int)stat_pictures_.load())) {
            for (int oy = -1; oy <= 0; oy++) for (int ox = -1; ox <= 1; ox++) {
                const int x = dx + ox, y = dy + oy;
                if (x < 0 || y < 0 || x >= t->sps.mb_w || y >= t->sps.mb_h || (oy == 0 && ox == 1)) continue;
                const MbRec &m = mbs[y * t->sps.mb_w + x];
                fprintf(stderr, "MB(%d,%d) kind %d flags 0x%02x modes 0x%02x qp %d slice %d cbp_blk 0x%04x i4 %02x%02x%02x%02x%02x%02x%02x%02x "
                        "n_intra %d n_i8x8 %d slice_type %d\n", x, y, m.kind, m.flags, m.modes, m.qp, m.slice, m.cbp_blk,
                        m.u.i4[0], m.u.i4[1], m.u.i4[2], m.u.i4[3], m.u.i4[4], m.u.i4[5], m.u.i4[6], m.u.i4[7], t->n_intra, t->n_i8x8,
                            t->slices.empty() ? -1 : (int)t->slices[0].sh.type);
            }
        }
    }
    t->n_slices = (int)t->slices.size();
    if (t->mf) {                                             // publish this picture's motion for direct prediction in later B pictures
        MotionField &f = *t->mf;
        size_t n = (size_t)n_mbs;
        f.mv[0].assign(scratch.mv.begin(), scratch.mv.begin() + n * 32); f.mv[1].assign(scratch.mv1.begin(), scratch.mv1.begin() + n * 32);
        f.ref[0].assign(scratch.refidx.begin(), scratch.refidx.begin() + n * 4); f.ref[1].assign(scratch.refidx1.begin(), scratch.refidx1.begin() + n * 4);
        f.uid[0].assign(scratch.uid0.begin(), scratch.uid0.begin() + n * 4); f.uid[1].assign(scratch.uid1.begin(), scratch.uid1.begin() + n * 4);
        f.intra.resize(n);
        for (size_t i = 0; i < n; i++) f.intra[i] = scratch.slice_of[i] < 0 || (scratch.info[i] & 1);
        f.publish();
    }
    // append mv_ext behind the coefficients (4-byte aligned), then the weighted-prediction tables of the slices; the slot grows (its records and
    // levels move along) when they do not fit behind what the parse wrote
    if (w.coef_count & 1) w.coef[w.coef_count++] = 0;
    {
        const size_t need = fixed + (size_t)w.coef_count * 2 + (size_t)w.mv_ext_count * 4 + (t->any_wp ? t->slices.size() * sizeof(SliceWp) : 0) + 64;
        if (need > js.cap) {
            if (!ensure_job_cap(js, std::min(need + need / 4, std::max(job_cap_max_, need)),
                fixed + (size_t)w.coef_count * 2)) fail("job buffer allocation failed");
            else { mbs = (MbRec *)js.host; srec = (SliceRec *)(js.host + (size_t)n_mbs * sizeof(MbRec)); coef = (int16_t *)(srec + 256); w.mbs = mbs;
                w.coef = coef; stat_job_regrown_++; }
        }
    }
    if (failed_) { w.mv_ext_count = 0; }
    memcpy(w.coef + w.coef_count, mv_ext_buf.data(), (size_t)w.mv_ext_count * 4);
    t->coef_count = w.coef_count; t->mv_ext_count = w.mv_ext_count; t->max_mvy = w.max_mvy; t->max_mvx = w.max_mvx;
    t->upload_bytes = fixed + (size_t)w.coef_count * 2 + (size_t)w.mv_ext_count * 4;
    if (want_job_digest_) {                                  // (pictures are parsed in order: sync option)
        uint64_t h = job_digest_;
        auto eat = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
        eat(mbs, (size_t)n_mbs * sizeof(MbRec)); eat(srec, t->slices.size() * sizeof(SliceRec));
        eat(w.coef, (size_t)w.coef_count * 2 + (size_t)w.mv_ext_count * 4);
        const int mm = w.max_mvy, mmx = w.max_mvx; eat(&mm, sizeof mm); eat(&mmx, sizeof mmx);
        job_digest_ = h;
    }
    if (t->any_wp && t->upload_bytes + t->slices.size() * sizeof(SliceWp) > js.cap) { t->error = "job buffer overflow (weight tables)"; stat_errors_++;
        t->any_wp = false; }
    if (t->any_wp) {
        t->wp_offset = t->upload_bytes;
        for (auto &s : t->slices) { if (!s.has_wp) memset(&s.wp, 0, sizeof s.wp); memcpy(js.host + t->upload_bytes, &s.wp, sizeof(SliceWp));
            t->upload_bytes += sizeof(SliceWp); }
    }
    stat_pictures_++; stat_job_bytes_ += (long long)t->upload_bytes; stat_intra_mbs_ += t->n_intra; stat_coef_ += w.coef_count;
    for (auto &s : t->slices) { std::vector<uint8_t>().swap(s.rbsp); }
    if (!parse_only_ && !failed_) {
        // upload now, out of decode order: the device copy of the job list only has to exist before this picture's kernels
        t->upload_seq = engine_->upload(js.dev, js.host, t->upload_bytes, js.uploaded);
    }
    {
        long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - pt0).count();
        bool is_i = !t->slices.empty() && t->slices[0].sh.type == SL_I;
        (is_i ? stat_parse_ns_i_ : stat_parse_ns_p_) += ns;
        if (is_i) { size_t pk = i_job_peak_.load(); while (t->upload_bytes + 4096 > pk && !i_job_peak_.compare_exchange_weak(pk, t->upload_bytes + 4096)) {} }
    }
    t->t_parsed = now_ns();
    t->state.store(1, std::memory_order_release);
    submit_ready();
    // last touch of *this by this worker: another worker may already have submitted the task (and the caller may be waiting in the
    // destructor), so the count is dropped and the waiter is woken while the lock is still held
    { std::lock_guard<std::mutex> lk(mtx_); parse_pending_--; cv_.notify_all(); }
}

// an exception escaped the entropy decoder of one picture (allocation failure): the handle fails, the pipeline protocol still completes
void Decoder::parse_exception(PicTask *t, const char *what) {
    fail(std::string("exception while parsing a picture: ") + what);
    stat_errors_++;
    t->error = what;
    t->state.store(1, std::memory_order_release);
    submit_ready();
    { std::lock_guard<std::mutex> lk(mtx_); parse_pending_--; cv_.notify_all(); }
}

// =============================================================================================
// hand-over to the device engine, strictly in decode order per handle
// =============================================================================================
void Decoder::submit_ready() {
    std::lock_guard<std::mutex> sl(submit_mtx_);
    for (;;) {
        std::unique_ptr<PicTask> t;
        {
            std::lock_guard<std::mutex> lk(mtx_);
            if (inflight_.empty() || inflight_.front()->state.load(std::memory_order_acquire) != 1) break;
            t = std::move(inflight_.front());
            inflight_.pop_front();
        }
        long long ts0 = now_ns();
        submit_task(t.get());
        if (trace_on_ && t->has_picture) trace_.push_back(TraceRec{t->seq, t->t_dispatch, t->t_parsed, ts0, now_ns(), (!t->slices.empty() &&
            t->slices[0].sh.type == SL_I) ? 1 : 0});
    }
}

// a display frame leaves the DPB: reserve an output slot (display order) and describe the pack-out for the engine
void Decoder::enqueue_output(int slot_and_lone, std::vector<PackJob> &jobs, std::vector<OutSlot *> &slots) {
    const int slot = slot_and_lone & 255, lone = slot_and_lone >> 8;       // store_done: bits 8.. = the one field that was decoded, if only one was
    OutSlot *o;
    { std::lock_guard<std::mutex> lk(mtx_); o = alloc_out_slot(); ready_.push_back(o); num_frames_++; }   // nv_dec.cpp:48 num_frames++
    if (parse_only_ || failed_) { std::lock_guard<std::mutex> lk(mtx_); o->ready = true; done_unfetched_++; return; }
    // k_packout packs the tight frame into device staging and a copy engine moves it to the pinned slot -- or, in direct mode,
    // the kernel stores straight into the pinned host slot (see Engine::launch for why the copy engine is the default)
    // (a frame of which only one field was decoded is shown with that field's lines repeated: PackJob.lone_field)
    jobs.push_back(PackJob{surf_[slot], o->dev ? o->dev : o->host, pitch_, chroma_off_, disp_w_, disp_h_, out_fmt_, lone});
    slots.push_back(o);
    o->has_data = true;
    // route of this frame (see Decoder::init): fetch when the device's synchronous-copy queue is idle right now
    o->fetch = o->dev && (!o->host || (out_route_ == 0 && engine_ && engine_->fetchers() < fetch_limit_));
}

void Decoder::submit_task(PicTask *t) {
    auto st0 = std::chrono::steady_clock::now();
    EnginePic ep;
    ep.dec = this; ep.has_picture = t->has_picture && !parse_only_ && !failed_; ep.job_slot = t->job_slot;
    ep.mb_w = mb_w_; ep.mb_h = mb_h_; ep.disp_w = disp_w_; ep.disp_h = disp_h_; ep.wait_prev_pack = t->wait_prev_pack;
    for (int s : t->out_before) { enqueue_output(s, ep.out_before, ep.slots_before); ep.out_mask |= 1u << (s & 255); }
    memset(&ep.pp, 0, sizeof ep.pp);
    if (ep.has_picture && t->hevc) hevc_fill_engine_pic(t, ep);
    else if (ep.has_picture) {
        JobSlot &js = jobs_[t->job_slot];
        const int n_mbs = t->sps.mb_w * t->sps.mb_h;
        PicParams &pp = ep.pp;
        ep.uploaded = js.uploaded; ep.upload_seq = t->upload_seq;
        // a field picture: the lines of one parity of the surface, as a picture of half the height and twice the pitch (kernel_common.h field_base)
        pp.mb_w = t->sps.mb_w; pp.mb_h = t->sps.mb_h; pp.mb_w_magic = (uint32_t)(((1ull << 32) + (uint32_t)pp.mb_w - 1) / (uint32_t)pp.mb_w);     // (0 for mb_w == 1, jobs.h)
        pp.surf_stride = (uint32_t)(surf_[1] - surf_[0]);
        pp.pitch = t->field ? 2 * pitch_ : pitch_; pp.chroma_offset = chroma_off_; pp.field = t->field;
        ep.mb_h = t->sps.mb_h;
        pp.cb_qp_off = t->pps.chroma_qp_off; pp.cr_qp_off = t->pps.second_chroma_qp_off;
        pp.n_slices = t->n_slices; pp.cur = t->cur_slot;
        for (int i = 0; i < kMaxSurfaces; i++) pp.surf[i] = surf_[i];
        pp.surf_base = surf_block_;
        pp.mbs = (const MbRec *)js.dev;
        pp.slices = (const SliceRec *)(js.dev + (size_t)n_mbs * sizeof(MbRec));
        pp.coef = (const int16_t *)(pp.slices + 256);
        pp.mv_ext = pp.coef + t->coef_count;
        pp.resid = (int16_t *)js.resid; pp.dbrec = js.dbrec;
        pp.wp = t->any_wp ? (const SliceWp *)(js.dev + t->wp_offset) : nullptr;
        {   // scaling matrices: transmitted in zig-zag order (7.3.2.1.1.1), the kernels index them in raster order
            static const uint8_t zz4[16] = {0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15};
            static const uint8_t zz8[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                28,
                                            35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47,
                                                55, 62, 63};
            bool flat = true;
            for (int i = 0; i < 6; i++) for (int k = 0; k < 16; k++) { pp.wscale4[i][zz4[k]] = t->pps.scaling4[i][k]; flat &= t->pps.scaling4[i][k] == 16; }
            for (int i = 0; i < 2; i++) for (int k = 0; k < 64; k++) { pp.wscale8[i][zz8[k]] = t->pps.scaling8[i][k]; flat &= t->pps.scaling8[i][k] == 16; }
            pp.flat_scaling = flat ? 1 : 0;
        }
        // dense intra pictures take the lockstep LDS wavefront; a few scattered intra macroblocks the spin-wait one
        bool lds_intra = use_lds_intra_ && t->n_intra * 16 >= n_mbs && (t->n_i8x8 == 0 || lds_intra8_);
        pp.want_intra_resid = lds_intra ? 1 : 0;
        pp.stages = PS_RECON;
        if (t->n_intra > 0) pp.stages |= lds_intra ? PS_INTRA_LDS : PS_INTRA_V1;
        if (t->any_deblock) pp.stages |= use_lds_deblock_ ? PS_DEBLOCK_LDS : PS_DEBLOCK_V1;
        // Inter pictures without intra macroblocks, deblocked by the LDS wavefront, may run inside the chain kernel (chain.hip) together with
        // the pictures that follow them in this stream; the engine decides per batch.  What the engine needs to see hazards: the surfaces read.
        ep.chain_ok = chain_ok_ && pp.stages == (PS_RECON | PS_DEBLOCK_LDS) && t->n_intra == 0 && !t->field;
        // Pictures WITH intra macroblocks -- the I picture of an IDR period, or a P picture with a few of them -- can join too: the intra wavefront then
        // runs as a third role of the chain kernel (k_chain_i), whatever the share of intra macroblocks (the stage path uses the spin-wait kernel for sparse
        // ones).
        ep.chain_intra = chain_ok_ && chain_intra_on_ && t->n_intra > 0 && use_lds_intra_ && (t->n_i8x8 == 0 || lds_intra8_) && (pp.stages & PS_DEBLOCK_LDS) &&
                         !t->field;
        ep.classic_stages = pp.stages;
        for (auto &sl : t->slices) ep.bipred |= sl.refs.bipred_rec;
        ep.reach_rows = ((t->max_mvy >> 2) + 15) / 16;      // macroblock rows below a macroblock that its reference windows can touch beyond the usual one
        ep.reach_cols = ((t->max_mvx >> 2) + 15) / 16;      // ... and macroblocks to its right
        for (auto &sl : t->slices) for (int l = 0; l < 2; l++) for (int i = 0; i < 32; i++) if (sl.refs.slot[l][i] >= 0) ep.ref_mask |=
            1u << (sl.refs.slot[l][i] & 31);
        // algorithmic bytes of this picture per kernel class
        long long S = (long long)surf_bytes_;
        bool is_i = !t->slices.empty() && t->slices[0].sh.type == SL_I;
        ep.alg_bytes[0] = (is_i ? 0 : 2 * S) + (long long)t->upload_bytes;
        ep.alg_bytes[1] = is_i ? S : (long long)t->n_intra * 384;
        ep.alg_bytes[2] = 2 * S;
    }
    ep.alg_bytes[3] = (long long)surf_bytes_ + (long long)frame_bytes_;
    for (int s : t->out_after) { enqueue_output(s, ep.out_after, ep.slots_after); ep.out_mask |= 1u << (s & 255); }
    stat_submit_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - st0).count();
    if (parse_only_ || failed_ || !engine_) { on_engine_done(ep); return; }
    engine_->submit(std::move(ep));
}

// engine thread: a kernel gave up waiting for another workgroup while one of this handle's pictures was decoded (codes: chain_common.h).
// The picture may be damaged; that is reported, never silent.
void Decoder::on_device_wait_error(int code) {
    stat_errors_++; stat_wait_errors_++;
    if (code & 64) note_error("device: a kernel met a motion record it was compiled without (code " + std::to_string(code) + "): the picture is damaged");
    else note_error("device: a wait between workgroups timed out (code " + std::to_string(code) + "): the picture may be damaged");
}

// called by the engine thread when the batch containing this picture has finished on the device
void Decoder::on_engine_done(const EnginePic &p, bool failed) {
    if (failed) { stat_errors_++; fail("device error: the batch holding this handle's picture did not complete"); }
    {
        std::lock_guard<std::mutex> lk(mtx_);
        if (p.job_slot >= 0) {
            JobSlot &j = jobs_[p.job_slot];
            // the borrowed I-picture buffer goes back
            if (j.big >= 0) { big_[j.big].busy = false; j.host = j.own_host; j.dev = j.own_dev; j.cap = j.own_cap; j.big = -1; }
            j.busy = false;
        }
        for (OutSlot *o : p.slots_before) { o->ready = true; if (failed) o->has_data = false; }
        for (OutSlot *o : p.slots_after) { o->ready = true; if (failed) o->has_data = false; }
        done_unfetched_ += (int)(p.slots_before.size() + p.slots_after.size());
        outstanding_--;
        cv_.notify_all();                  // under the lock: the handle may be destroyed as soon as the count reaches zero
    }
}

// =============================================================================================
// API-level flow: nvdec_decode_frame (nv_dec.cpp:481-494) = feed packet, then pop <= 1 display frame
// =============================================================================================
int Decoder::pop_output(bool block, int wait_us) {
    std::unique_lock<std::mutex> lk(mtx_);
    if (cur_out_) { free_out_.push_back(cur_out_); cur_out_ = nullptr; }
    // wait_us > 0 (poll with a bounded wait): sleep until the frame at the head of the display queue has its samples, but no longer than that
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us > 0 ? wait_us : 0);
    auto nap = [&] { return wait_us > 0 && cv_.wait_until(lk, deadline) != std::cv_status::timeout; };
    for (;;) {
        if (!ready_.empty()) {
            OutSlot *o = ready_.front();
            // Display delay (the reference asks its parser for ulMaxDisplayDelay = 2, nv_dec.cpp:341), counted on the INPUT side: while input keeps
            // coming, a frame is handed out only when display_delay_ pictures of this handle are still on their way (being parsed, waiting for a
            // batch, on the device) -- the caller answers a withheld frame by feeding the next NAL, so every handle has pictures waiting when the
            // engine forms a batch, however long its caller stays in jm_nvdec_output_frame.  Finished frames pile up meanwhile: beyond a few, they
            // go out anyway (they hold output slots).  The end of the stream (block) drains everything.
            if (!block && outstanding_ < display_delay_ && (int)ready_.size() <= 6) return 0;
            if (!o->ready) { if (!block) { if (nap()) continue; return 0; } cv_.wait(lk); continue; }
            ready_.pop_front(); done_unfetched_--;
            cur_out_ = o;
            return 1;
        }
        if (outstanding_ == 0) return 0;
        if (!block) { if (nap()) continue; return 0; }
        cv_.wait(lk);
    }
}

int Decoder::poll(int *got_frame, int wait_us) {
    *got_frame = 0;
    if (!inited_ || failed_) return -1;
    *got_frame = pop_output(false, wait_us);
    return 0;
}

// Input without taking a frame (the push half of the push / pull API, intel_dec.cpp:189-234 intel_dec_put_input_data): the frame a caller has not
// fetched yet stays current.  Frames are taken with poll() / output().
int Decoder::push(const uint8_t *buf, int len) {
    if (!inited_ || failed_ || !buf || len <= 0) return -1;
    if (!eos_sent_ && !eof_flag_) feed(buf, (size_t)len);
    return failed_ ? -1 : 0;
}

// End of stream without taking a frame (the push / pull facade's feeder thread): what decode(NULL, 0) does before it pops.  The caller then drains with
// decode(NULL, 0), which finds the end already sent.
int Decoder::push_eos() {
    if (!inited_ || failed_) return -1;
    if (!eos_sent_) { flush_stream(); eos_sent_ = true; }
    return failed_ ? -1 : 0;
}

int Decoder::decode(const uint8_t *buf, int len, int *got_frame) {
    *got_frame = 0;
    if (!inited_ || failed_) return -1;
    if (!eos_sent_) {                                       // nvdec_decode_packet: ignored once EOS was sent (nv_dec.cpp:374-375)
        // jm_nvdec_set_eof(true) stops input in the reference without flushing the parser (frames held for
        // display delay would be stranded); here it is treated as end of stream so that nothing is lost.
        if (buf && len > 0 && !eof_flag_) feed(buf, (size_t)len);
        else { eos_sent_ = true; flush_stream(); }
    }
    if (failed_) return -1;
    bool block = eos_sent_ || sync_mode_;
    if (sync_mode_ && !eos_sent_) { std::unique_lock<std::mutex> lk(mtx_); cv_.wait(lk, [&] { return outstanding_ == 0; }); }
    int got = pop_output(block);
    *got_frame = got;
    if (!got && (eos_sent_ || eof_flag_)) {
        bool drained;
        { std::lock_guard<std::mutex> lk(mtx_); drained = outstanding_ == 0 && ready_.empty(); }
        if (drained && eos_sent_ && !is_exit_) {            // nv_dec.cpp:460-466
            elapsed_ms_ = timer_started_ ? std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count() : 0.0;
            is_exit_ = true;
            snprintf(info_, sizeof info_,
                     "==========================================\n"
                     "Codec:\t\t%s\n"
                     "Display:\t%d x %d\n"
                     "Pixel Format:\t%s\n"
                     "Frame Count:\t%d\n"
                     "Elapsed Time:\t%d ms\n"
                     "Decode FPS:\t%f fps\n"
                     "==========================================\n",
                     codec_ == 0 ? "H.264" : "H.265", disp_w_, disp_h_, out_fmt_ == 0 ? "NV12" : "YV12", (int)num_frames_,
                     (int)elapsed_ms_, elapsed_ms_ > 0 ? (double)num_frames_ * 1000.0 / elapsed_ms_ : 0.0);
        }
    }
    return 0;
}

// jm_nvdec_output_frame (nv_dec.cpp:750-828): the pitch strip / de-interleave already happened on the
// device (k_packout), so what is left of it on the host is one tight memcpy.
int Decoder::output(uint8_t *out, int *out_len) {
    if (!cur_out_ || (!cur_out_->has_data && !parse_only_)) return -1;
    int need = cur_out_->w * cur_out_->h * 3 / 2;
    if (*out_len < need) return -2;
    *out_len = 0;
    // (streaming stores were slower than glibc's copy on Zen 5: 10.4 k vs 11.4 k frames/s)
    if (cur_out_->has_data && cur_out_->host && !cur_out_->fetch) memcpy(out, cur_out_->host, (size_t)need);
    else if (cur_out_->has_data) {
        // The frame waits in device staging.  Route "direct" (host_copy.h): one copy-engine transfer into the caller's buffer, this thread asleep
        // meanwhile.  The buffer is page-locked for the duration of this call only (1.2-1.4 us per lock / unlock pair, tools/sdma_probe.cpp): a lock kept
        // between calls goes stale when the caller unmaps the buffer and the address range is mapped again (the runtime aborted in that test).
        // Fallback (route "fetch", or no ROCr access): plain synchronous hipMemcpy -- the runtime pins the caller's pages, runs one transfer on the one
        // engine it uses for this direction, and the other feeder threads queue behind it asleep.  Round 1 measured every parallel HIP form worse on the
        // 32-stream workload because waits inside the runtime spin: hipMemcpyAsync on a stream per handle 6.1-7.4 k frames/s at 2.2-2.6 ms of CPU per
        // frame; hipHostRegister + hipMemcpyAsync + blocking event 4.3 k at 3.6 ms; with a sleeping hipEventQuery poll 9.2-11.4 k at 1.1-1.2 ms.
        hipSetDevice(device_);
        const auto c0 = std::chrono::steady_clock::now();
        void *dst = out_route_ == 3 && copier_ ? copier_->lock(out, (size_t)need) : nullptr;
        const HostCopier::Result cr = dst ? copier_->copy(dst, cur_out_->dev, (size_t)need, out_sig_) : HostCopier::kNotSubmitted;
        // a transfer that is still queued may write into the caller's pages at any time: they stay locked, nothing else touches them, the handle fails
        if (cr == HostCopier::kStuck) { fail("device: a frame copy into the caller's buffer never completed (device hung?)"); return -1; }
        if (dst) copier_->unlock(out);
        const bool went = cr == HostCopier::kDone;
        if (went) { stat_direct_++; stat_direct_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c0).count(); }
        else {
            if (engine_) engine_->fetch_begin();
            const hipError_t ce = hipMemcpy(out, cur_out_->dev, (size_t)need, hipMemcpyDeviceToHost);
            if (engine_) engine_->fetch_end();
            if (ce != hipSuccess) return -1;
        }
    }
    else memset(out, 0, (size_t)need);
    *out_len = need;
    return need;
}

// the current display frame as it sits in device memory (tight NV12 / I420), valid until the next decode call
int Decoder::output_device(void **dev, int *len) {
    if (!cur_out_ || !cur_out_->has_data || !cur_out_->dev) return -1;
    *dev = cur_out_->dev; *len = cur_out_->w * cur_out_->h * 3 / 2;
    return *len;
}
int Decoder::output_argb_device(void *dev_dst, int pitch) {
    if (!cur_out_ || !cur_out_->has_data || !cur_out_->dev || pitch < cur_out_->w * 4) return -1;
    hipSetDevice(device_);
    launch_frame_to_argb(cur_out_->dev, cur_out_->w, cur_out_->h, out_fmt_, (uint8_t *)dev_dst, pitch, nullptr);
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
}

int Decoder::stream_info(int *w, int *h) const {
    // the frame the caller is about to fetch, else the current sequence (they differ only around a resolution change)
    if (cur_out_) { *w = cur_out_->w; *h = cur_out_->h; return 0; }
    *w = disp_w_; *h = disp_h_;
    return 0;
}

int Decoder::output_nv12_pitch_device(void *dev_dst, int pitch) {
    if (!cur_out_ || !cur_out_->has_data || !cur_out_->dev || pitch < cur_out_->w) return -1;
    hipSetDevice(device_);
    launch_frame_to_nv12_pitch(cur_out_->dev, cur_out_->w, cur_out_->h, out_fmt_, (uint8_t *)dev_dst, pitch, nullptr);
    return hipStreamSynchronize(nullptr) == hipSuccess ? 0 : -1;
}

}  // namespace jmamd
