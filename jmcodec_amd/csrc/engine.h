// jmcodec_amd/csrc/engine.h -- per-device execution engine: batches one picture per stream into every kernel launch.
//
// Why: single-picture kernels of the serial stages (intra / deblock wavefronts) occupy 2 CUs for ~1 ms, and the GPU
// front end only keeps a handful of HW queues running concurrently, so "one HIP stream per decoder" tops out at ~5
// pictures in flight on a 256-CU part.  The engine instead owns the device: each round it takes the next ready picture
// of every decoder (pictures of one stream depend on each other, pictures of different streams do not -- SURVEY.md 8e)
// and issues ONE batched launch per stage with blockIdx.y = picture.  A 32-stream batch is 32x the work per launch at
// the same latency.
//
// Lanes, each an in-order HIP stream with its own pack-out stream: lane 0 takes ordinary H.264 pictures (P, B, sparse intra), lane 1 those whose
// macroblocks are mostly intra (I pictures: +3 ms of intra wavefront), lanes 2 / 3 the same two classes of HEVC pictures.  Without the separate lanes
// one I picture would hold up every other stream's round.  A decoder may only change lane when its previous pictures
// have completed, so decode order per stream is preserved by stream order inside a lane.
// There is no reference counterpart: the reference drives one NVDEC session synchronously (nv_dec.cpp:33-41).
#pragma once
#include "jobs.h"
#include "hevc_jobs.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ihipStream_t; struct ihipEvent_t;

namespace jmamd {

class Decoder;
struct OutSlot;

constexpr int kMaxBatch = 64;
constexpr int kBatchRing = 4;
constexpr int kMaxChainGroups = 128 * 1024;         // work list of one chain launch (1080p: 1025 groups per picture, 4K: 4059)
// band workgroups of one chain launch, upper bound: half of what a whole MI355X holds at 3 workgroups per CU (k_chain)
constexpr int kMaxChainBands = 384;
// ... at 2 per CU (k_chain_i, the variant with the intra role). The bounds in use come from the device (Engine::Engine)
constexpr int kMaxChainBandsIntra = 256;
// Lane 0 takes ordinary pictures.  (Rounds 1-2 carried an optional second ordinary lane with the streams divided by handle parity; it measured worse
// every time -- 32 streams 12.5 k against 16.8 k frames/s: two half-size batches take as long as one and their kernels get in each other's way --
// and was removed in round 3.)
// smallest spacing of consecutive pictures of a chain in the work list: the rule of Engine::launch (every dependency has a smaller key) needs
// lag > 15 + kBandLag, so the knob cannot go below it
constexpr int kMinChainLag = 24;
constexpr int kOrdinaryLane = 0;
constexpr int kIntraLane = 1;                       // intra-dense H.264 pictures
constexpr int kErrNotRecovered = 32;                // error word set by Engine::recover (chain_common.h CHAIN_ERR_NOT_RECOVERED; the device sets 1..16)
constexpr int kLanes = 4;                           // ordinary, intra-dense H.264, HEVC, HEVC I pictures
constexpr int kHevcLane = 2;
// an I picture's CTB-row wavefront (k_hevc_intra, 2-3 ms at 1080p) would hold up every other stream's P / B batch
constexpr int kHevcIntraLane = 3;
constexpr int kEarlyIntraAhead = 10;               // an intra picture runs ahead of its turn only with at least this many earlier pictures of its stream pending

struct EnginePic {
    Decoder *dec = nullptr;
    bool has_picture = false;
    PicParams pp;                                   // device pointers already resolved by the decoder
    int codec = 0;                                  // 0 = H.264 (pp), 1 = HEVC (hp)
    HevcPicParams hp;
    int job_slot = -1;
    ihipEvent_t *uploaded = nullptr; unsigned long long upload_seq = 0;   // job list copy (copy stream), see Engine::upload
    std::vector<PackJob> out_before, out_after;     // display frames to pack before / after this picture's kernels
    std::vector<OutSlot *> slots_before, slots_after;
    int mb_w = 0, mb_h = 0, disp_w = 0, disp_h = 0;
    bool wait_prev_pack = false;                    // this picture reuses a surface whose pack-out may still be running
    // chain launches (chain.hip): the picture may run inside k_chain, i.e. in the same launch as the pictures before it in its stream
    bool chain_ok = false;
    // a (mostly) intra picture that can run inside k_chain_i with its intra wavefront as a third role (chain_intra.hip)
    bool chain_intra = false;
    int classic_stages = 0;                         // pp.stages when it runs through the stage kernels instead
    uint32_t ref_mask = 0, out_mask = 0;            // surface slots this picture reads as references / displays (pack-out reads them)
    bool bipred = false;                            // some slice of the picture writes two-list / weighted motion records (B slices, weighted prediction)
    int reach_rows = 0;                             // how many macroblock rows further down than usual its vectors reach into the reference pictures
    int reach_cols = 0;                             // ... and how many macroblocks further right (both space the pictures of a chain launch, Engine::launch)
    long long alg_bytes[4] = {0, 0, 0, 0};          // algorithmic bytes of this picture per kernel class (recon, intra, deblock, packout)
    unsigned long long seq = 0;                     // position in its decoder's decode order (Engine::submit)
    // chaining: the engine currently forms chain launches -- an intra picture that can join one stays on the ordinary lane
    int lane(bool chaining = false) const {
        if (codec == 1) return (has_picture && hp.n_pus == 0 && hp.n_itbs > 0) ? kHevcIntraLane : kHevcLane;
        return (has_picture && (pp.stages & PS_INTRA_LDS) && !(chaining && chain_intra)) ? kIntraLane : kOrdinaryLane;
    }
};

// per kernel class: 0 recon_inter, 1 intra, 2 deblock (prep+lds), 3 packout, 4 chain (k_chain: recon + deblock)
struct EngineStats {
    double ns[5] = {0, 0, 0, 0, 0}; long long launches[5] = {0, 0, 0, 0, 0}, pics[5] = {0, 0, 0, 0, 0}, alg_bytes[5] = {0, 0, 0, 0, 0};
    long long batches = 0, batch_pics = 0, chain_batches = 0, chain_pics = 0, wait_errors = 0, chain_recoveries = 0;
    long long chain_i_batches = 0;                  // chain launches that ran k_chain_i (the variant with the intra role); the others ran k_chain
    long long forms = 0, form_decoders = 0, form_pending = 0;   // ordinary-lane batches formed; decoders that had a picture waiting then; pictures waiting then
    long long launch_ns = 0, complete_ns = 0;      // engine thread time spent issuing a batch / retiring it
    // why a decoder with a picture waiting was left out when an ordinary-lane batch was formed (round 6): its oldest picture belongs to another lane (an
    // I picture); it is an ordinary picture but the decoder still has pictures on another lane; the pack-job tables were full
    long long rej_other_lane = 0, rej_cross_lane = 0, rej_tables = 0;
    long long wait_gap_launches = 0, wait_gap_max_ticks = 0;   // chain launches whose waits saw the clock jump (their waves were not run meanwhile), longest jump
    long long quad_windows = 0, private_windows = 0;   // JM_AMD_DEC_CENSUS: reconstruction workgroups of chain launches with one shared / four private reference windows
    long long early_intra = 0;                     // intra pictures launched ahead of their stream's earlier pictures (Engine::form)
    long long blocked_ns = 0, blocked_n = 0;       // time decoders spent left out of ordinary batches between two of their pictures joining one, and how often
    // per lane, from the profile events: time between a batch's first and last kernel, time its stream sat idle before it, batches and pictures
    // where a lane's idle time between two batches went (diagnostic, profile mode): this batch's job lists had not all landed when the previous batch ended /
    // its pre-pass (tables, deblocking pre-pass on the pre-stream) had not finished / it was only launched after the previous batch had ended (count)
    double lane_upwait_ns[4] = {0, 0, 0, 0}, lane_prewait_ns[4] = {0, 0, 0, 0}; long long lane_dry[4] = {0, 0, 0, 0};
    double lane_busy_ns[4] = {0, 0, 0, 0}, lane_gap_ns[4] = {0, 0, 0, 0}; long long lane_batches[4] = {0, 0, 0, 0}, lane_pics[4] = {0, 0, 0, 0};
};

// engine-private state kept inside each Decoder (touched only by the engine thread)
struct EngineDecoderState {
    int lane = -1, inflight = 0;                    // lane of the decoder's most recent in-order picture; pictures in flight (all lanes)
    int lane_inflight[4] = {0, 0, 0, 0};            // ... per lane
    uint32_t displayed[4] = {0, 0, 0, 0};           // per lane: surfaces displayed by this decoder's pictures in the lane's most recently launched batch
    unsigned long long next_seq = 0;                // decode-order numbering of the decoder's pictures (Engine::submit, m_)
    int n_pending = 0;                              // the decoder's pictures in Engine::pending_ (m_)
    unsigned long long form_tag = 0;                // scratch of Engine::form's look for urgent pictures
    unsigned long long scan_tag = 0; uint32_t scan_touched = 0; int scan_ahead = 0; bool scan_closed = false;   // scratch of the look for intra pictures (Engine::form)
    long long blocked_since = 0;                    // diagnostic: when an ordinary-lane batch first left this decoder out (0: not left out)
    // scratch of Engine::form (one batch at a time): what the decoder's pictures already in the batch write / read
    int in_batch = 0; uint32_t batch_written = 0, batch_read = 0; bool batch_chain = false, batch_stop = false, batch_resid = false;
};

void mem_trace(const char *tag);      // developer aid: JM_AMD_DEC_MEMTRACE=1 prints the process's resident memory at set-up steps (engine.cpp)

class Engine {
public:
    static Engine *get(int device);                 // creates the engine (and its thread) on first use; nullptr on HIP failure
    void submit(EnginePic &&p);                     // decode order per decoder; thread-safe
    // async H2D of a parsed job list on the engine's copy stream; records `ev` behind it and returns its sequence number
    // spread = false: always copy stream 0 (HEVC: its few, large job lists gained nothing from a second stream and its parse threads paid 5-12 % more CPU
    // per frame for it, profiles/r06_copy_streams.txt)
    unsigned long long upload(uint8_t *dev, const uint8_t *host, size_t n, ihipEvent_t *ev, bool spread = true);
    void set_profile(bool on) { profile_ = on; }
    // threads currently inside a synchronous device-to-host frame copy (jm_nvdec_output_frame, fetch route): the copies of a device run one after another
    int  fetchers() const { return fetchers_.load(std::memory_order_relaxed); }
    void fetch_begin() { fetchers_.fetch_add(1, std::memory_order_relaxed); }
    void fetch_end() { fetchers_.fetch_sub(1, std::memory_order_relaxed); }
    // engine-wide knobs (all handles of the device): "chain_depth", "chain_lag", "chain_streams", "debug_stall"; false = unknown key
    bool set_knob(const std::string &key, long long v);
    EngineStats stats();
    int device() const { return device_; }
    // another process has compute queues on this GPU (no chain launches then)
    bool gpu_shared() const { return gpu_shared_.load(std::memory_order_relaxed); }

private:
    explicit Engine(int device);
    void run();
    struct Batch {
        PicParams *h_pics = nullptr, *d_pics = nullptr;       // pinned host / device, kMaxBatch entries
        HevcPicParams *h_hpics = nullptr, *d_hpics = nullptr; // the same for HEVC batches
        int *d_progress = nullptr;                            // CTB row progress counters of k_hevc_intra
        int *d_ctl = nullptr;                                 // H.264: kMaxBatch control blocks (chain_common.h), cleared once per batch
        int *h_err = nullptr, *d_err = nullptr;               // error words, one per picture: pinned host memory and its device address
        // redo: an earlier batch of the lane was recovered, this one read its (then damaged) output
        bool any_chain = false, chain_with_intra = false, redo = false; int max_depth = 1;
        int max_mbs = 0, max_mb_h = 0, max_w = 0, max_h = 0; bool any_bipred = false, any_field = false;
        uint32_t *h_groups = nullptr, *d_groups = nullptr;     // work list of k_chain (chain.hip), kMaxChainGroups entries
        PackJob *h_jobs = nullptr, *d_jobs = nullptr;         // 4 * kMaxBatch entries
        // packed: surfaces were read by k_packout (before the copies)
        ihipEvent_t *done = nullptr, *kdone = nullptr, *packed = nullptr, *pre_done = nullptr, *pev[10] = {nullptr};
        std::vector<EnginePic> pics;
        int n_pre = 0, n_post = 0; unsigned pmask = 0;
        long long alg[5] = {0, 0, 0, 0, 0}; int npics[5] = {0, 0, 0, 0, 0};
        int last_ev = -1;                                     // index of the profile event behind the batch's last decode kernel
        bool launched_dry = false;                            // diagnostic: the lane's previous batch had already ended when this one was launched
        unsigned long long serial = 0;                        // position of the batch in its lane's sequence of launches
    };
    struct Lane {
        // pre_stream: what a batch can do before the previous batch is complete
        ihipStream_t *stream = nullptr, *pack_stream = nullptr, *pre_stream = nullptr;
        ihipEvent_t *pack_hist[2] = {nullptr, nullptr};        // 'packed' events of the two most recently launched batches
        Batch ring[kBatchRing];
        int head = 0, tail = 0, inflight = 0;
        unsigned long long launched = 0;                       // batches launched on this lane so far
        long long wait_since_ns = 0;                           // Engine::form: since when the lane's next batch has been waiting for the streams that have nothing pending (0: not waiting)
        std::vector<Decoder *> tainted;                        // decoders whose recovered pictures could not be redone from intact references (Engine::recover)
    };
    bool form(Lane &ln, int lane_idx, Batch &b);              // m_ held
    // Round 6 (Engine::form): cross_lane_ -- a decoder changes lane as soon as the DEVICE is done with its pictures on the other lane (event query), not when
    // the host has retired them; early_intra_ -- an intra-only picture runs ahead of its stream's earlier pictures when nothing they touch is its surface
    bool cross_lane_ = true, early_intra_ = true;
    long long fill_linger_ns_ = 4000 * 1000;        // Engine::form: how long the ordinary lane's next stage batch may wait for the streams that have nothing pending yet, while no picture is urgent (JM_AMD_DEC_FILL_LINGER_US)
    unsigned long long scan_tag_form_ = 0;
    bool lingered_ = false;                         // engine thread: this turn of the loop a batch was held back by the fill linger
    int decoders_pending_ = 0;                      // under m_: decoders with pictures in pending_ (EngineDecoderState::n_pending)
    bool deep_queues_ = false;                      // engine thread: the decoders have many parsed pictures pending -- the engine is what they wait for (Engine::form)
    std::atomic<int> early_intra_ahead_{kEarlyIntraAhead};     // knob "early_intra_ahead" (tests)
    long long early_scan_ns_ = 0; unsigned long long early_scan_tag_ = 0;
    unsigned long long pending_gen_ = 0, early_scanned_gen_ = ~0ull;      // m_: bumped whenever pending_ or the set of pictures in flight changes
    // m_ held: the device has finished every picture of d earlier in decode order than `seq` that is in flight on another lane
    bool others_done(Decoder *d, int lane_idx, unsigned long long seq);
    void inflight_masks(Decoder *d, uint32_t &touched);       // m_ held: surfaces d's pictures in flight read, write or display
    // chains only while at most this many streams have pictures ready (JM_AMD_DEC_CHAIN_STREAMS): a wide batch fills the GPU anyway
    std::atomic<int> chain_max_streams_{16};
    // half of the workgroups THIS device keeps resident (occupancy x compute units)
    int chain_bands_max_ = kMaxChainBands, chain_bands_max_intra_ = kMaxChainBandsIntra;
    // chain_lag_steps_: spacing of consecutive pictures of a chain in the work list, in wavefront steps (JM_AMD_DEC_CHAIN_LAG)
    std::atomic<int> chain_depth_{8}, chain_lag_steps_{24};
    std::atomic<int> chain_depth_few_{16};          // ... with one or two active streams (an explicit chain_depth sets both)
    // H.264 decoders that submitted a picture lately (time of the last one): how many streams are active (m_)
    struct Recent { Decoder *dec; long long t; int mbs; };      // (dec is only compared, never dereferenced: the handle may be gone)
    std::vector<Recent> recent_;
    // scratch of launch() // pictures of one stream per launch at most (JM_AMD_DEC_CHAIN_DEPTH; 1 = off)
    std::vector<std::vector<uint32_t>> group_buckets_;
    std::vector<uint32_t> bucket_tmp_;                                 // scratch of append_bucket_by_xcd (chain_order.h)
    void launch(Lane &ln, Batch &b);
    void launch_hevc(Lane &ln, Batch &b);
    void complete(Lane &ln, Batch &b, bool failed);
    void dump_chain_state(Batch &b);                         // JM_AMD_DEC_VERBOSE: counters of a chain launch that gave up
    // decode a batch's pictures again with the stage kernels (a chain launch's wait gave up).  `later` = surfaces the lane's NEXT batch, which has
    // already run, decoded into, per decoder: a redo that would read one of them cannot be right, and is reported instead of passed off as clean
    void recover(Lane &ln, Batch &b, const std::vector<std::pair<Decoder *, uint32_t>> &later);

    int device_, numa_node_ = -1;
    // another process has queues on this GPU (checked about once a second): no chain launches
    unsigned kfd_gpu_id_ = 0; std::atomic<bool> gpu_shared_{false}; std::atomic<long long> shared_checked_ns_{0};
    void look_for_other_users();                    // engine thread, no lock held
    ihipStream_t *copy_stream_ = nullptr;           // job-list uploads (Engine::upload); copy_streams_[0]
    ihipStream_t *copy_streams_[4] = {nullptr, nullptr, nullptr, nullptr}; int n_copy_ = 2;     // upload k goes to stream k % n_copy_ (JM_AMD_DEC_COPY_STREAMS)
    std::mutex um_; unsigned long long upload_seq_ = 0;
    Lane lanes_[kLanes];
    std::mutex m_; std::condition_variable cv_;
    std::deque<EnginePic> pending_;
    bool profile_ = false, ok_ = false, device_failed_ = false;
    // test hook: 1 = no band publishes its step counter (stage kernels and chain launches), 2 = chain launches only
    std::atomic<int> debug_stall_{0};
    std::atomic<int> debug_no_bi_{0};
    std::atomic<int> fetchers_{0};
    // no chain launches before this time (set when one had to be recovered; setting a chain knob clears it)
    std::atomic<long long> chain_block_until_ns_{0};
    int chain_linger_streams_ = 2;                  // up to this many active streams the next chain launch waits for the running one (Engine::form)
    std::mutex sm_; EngineStats st_;
    std::thread th_;
};

}  // namespace jmamd
