// jmcodec_amd/csrc/decoder.h -- per-handle decode pipeline behind the jm_nvdec_* API.
//
// Re-implements nvdec_ctx and its call flow (/root/reference/nv_dec/nv_dec.h:69-126,
// nv_dec.cpp:62-80, :368-478, :496-540) without CUVID:
//   caller thread : Annex-B splitter -> SPS/PPS/slice headers -> POC, DPB, ref lists, display order
//   worker pool   : CAVLC slice_data() -> macroblock job list in a pinned buffer      (h264_cavlc.cpp)
//   HIP stream    : H2D job list -> k_recon_inter -> k_recon_intra -> k_deblock -> k_packout -> D2H
//   caller thread : jm_nvdec_output_frame = one memcpy out of the pinned output slot
#pragma once
#include "h264_cavlc.h"
#include "h264_syntax.h"
#include "hevc_slice.h"
#include "jobs.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

struct ihipStream_t; struct ihipEvent_t;

namespace jmamd {

constexpr int kMaxJobSlots = 64;  // upper bound of the option "job_slots" / JM_AMD_DEC_JOB_SLOTS
constexpr int kJobSlots = 24;   // pictures in flight per handle (parse + device); deep enough to hide an I picture's entropy decode behind a GOP of device work
// H.264 streams up to 1080p: one stream alone is fed by what fits into its slots -- a chain launch holds n pictures while the next n are parsed, and a launch
// costs ~0.85 ms plus ~0.1 ms per picture, so n decides its rate: 24 slots -> 6.8 pictures per launch, 4.5 k frames/s; 40 slots (and chains of up to 16) ->
// 14.6 per launch, 6.6 k (profiles/r06_single_stream_slots.txt).  1.3 MB of page-locked memory per slot at 1080p Baseline
constexpr int kJobSlotsSmall = 40;

struct DpbPic {                                // one frame store (C.4.5): a frame, or the one or two field pictures of a frame
    bool in_use = false;
    // marking.  fmark[]: each field (top, bottom): 0 not a reference, 1 short-term, 2 long-term.  ref is derived (sync_ref): 1 / 2 = BOTH fields so,
    // i.e. a reference FRAME in the sense of 8.2.4.2.1; 3 = some field is a reference (field pictures can use it, and the store stays occupied)
    int ref = 0; int fmark[2] = {0, 0};
    int fpoc[2] = {0, 0};                      // TopFieldOrderCnt, BottomFieldOrderCnt
    int have = 0;                              // bit 0 / 1: top / bottom field decoded (a frame picture: both)
    bool waiting_second = false, first_was_ref = false, coded_as_fields = false;
    bool non_existing = false;                 // a frame inferred from a gap in frame_num (8.2.5.2): a short-term reference without samples, never displayed
    int lone = 0;                              // set when the store is complete: 1 / 2 = only its top / bottom field was decoded
    void set_ref(int v) { fmark[0] = fmark[1] = v; ref = v; }
    void sync_ref() { ref = (fmark[0] == 1 && fmark[1] == 1) ? 1 : (fmark[0] == 2 && fmark[1] == 2) ? 2 : (fmark[0] || fmark[1]) ? 3 : 0; }
    bool any_short() const { return fmark[0] == 1 || fmark[1] == 1; }
    bool any_long() const { return fmark[0] == 2 || fmark[1] == 2; }
    bool wait_output = false;
    int poc = 0, frame_num = 0, frame_num_wrap = 0, pic_num = 0, lt_idx = -1;
    int decode_idx = 0; bool mmco5 = false;
    std::shared_ptr<MotionField> mf;           // motion of the FRAME picture (colocated data of later B frames)
    std::shared_ptr<MotionField> mf_fld[2];    // ... of the field pictures that filled the store (colocated data of later B fields)
    std::shared_ptr<HevcColMotion> hcol;       // HEVC: motion of this picture for temporal prediction
    int out_at = -100;                         // decode index of the picture after which this surface was displayed (cooling)
};

struct SliceTask {
    SliceHeader sh;
    std::vector<uint8_t> rbsp;                 // unescaped NAL payload (+ slack)
    size_t rbsp_len = 0;
    SliceRefs refs;
    std::shared_ptr<MotionField> col;          // keeps RefPicList1[0]'s motion field alive (B slices)
    SliceWp wp; bool has_wp = false;           // weighted prediction tables of this slice
};

// HEVC picture: the slice segments with their reference lists, and where the packed job lists sit in the job buffer
struct HevcSliceTask { HevcSliceHeader sh; HevcSliceRefs refs; std::vector<uint8_t> rbsp; size_t len = 0; };
struct HevcTask {
    HevcSps sps; HevcPps pps; int poc = 0, work_slot = -1;
    std::vector<HevcSliceTask> slices;
    std::shared_ptr<HevcColMotion> col_out;
    size_t off_ctbs = 0, off_qp8 = 0, off_pus = 0, off_tbs = 0, off_itbs = 0, off_coefs = 0, off_wps = 0;
    int n_pus = 0, n_tbs = 0, n_itbs = 0; bool any_sao = false, any_deblock = false;
};

struct PicTask {
    uint64_t seq = 0;
    bool has_picture = false;
    int cur_slot = -1, job_slot = -1;
    int field = 0;                             // 0 frame picture, 1 / 2 top / bottom field picture (sps.mb_h is then the FIELD's)
    std::shared_ptr<MotionField> mf;           // this picture's motion field (reference pictures of streams that may hold B pictures)
    SeqParams sps; PicParamSet pps;
    std::vector<SliceTask> slices;
    std::unique_ptr<HevcTask> hevc;            // codec_type 1
    std::vector<int> out_before, out_after;    // DPB slots to display before / after this picture
    bool wait_prev_pack = false;               // current surface was displayed by the previous picture (no cooling slack)
    // written by the parse worker
    std::atomic<int> state{0};                 // 0 queued, 1 parsed
    int n_intra = 0, n_i8x8 = 0, n_slices = 0; bool any_deblock = false;
    uint32_t coef_count = 0, mv_ext_count = 0; size_t upload_bytes = 0, wp_offset = 0; bool any_wp = false;
    int max_mvy = 0, max_mvx = 0;              // largest downward / rightward vector component (quarter samples): spacing of chain launches
    unsigned long long upload_seq = 0;
    std::string error;
    long long t_dispatch = 0, t_parsed = 0;    // host steady-clock ns (JM_AMD_DEC_TRACE)
};

struct JobSlot {                       // one picture's job list: pinned host buffer (parse target) + its device copy
    uint8_t *host = nullptr, *dev = nullptr; size_t cap = 0;
    // device scratch: int16 residual of this picture's intra macroblocks (768 B per macroblock), per slot for the same reason
    uint8_t *resid = nullptr;
    // device scratch of k_deblock_prep for this picture (96 B per macroblock): per slot, because pictures of a chain run concurrently
    uint8_t *dbrec = nullptr;
    ihipEvent_t *uploaded = nullptr;   // recorded behind the H2D copy on the engine's copy stream
    bool busy = false;                 // from dispatch until the engine reports the picture done
    int big = -1;                      // >= 0: host / dev / cap are those of borrowed big buffer `big` (an I picture); the slot's own are kept below
    uint8_t *own_host = nullptr, *own_dev = nullptr; size_t own_cap = 0;
};
// HEVC pictures of one handle that may share a batch (independent B pictures of a pyramid): each needs its own pre-SAO work surface and residual scratch
constexpr int kHevcWorkSets = 4;
// worst-case-sized job buffers per H.264 handle, lent to I pictures (one in thirty pictures of config C1; two in flight at most)
constexpr int kBigJobBufs = 3;
struct BigJobBuf { uint8_t *host = nullptr, *dev = nullptr; bool busy = false; };
struct OutSlot {                       // one display frame in pinned host memory, written by k_packout
    uint8_t *host = nullptr;
    uint8_t *dev = nullptr;            // device staging of the packed frame (copy-engine mode, see Engine::launch)
    size_t bytes = 0;
    int w = 0, h = 0;                  // display size of the frame held (a stream may change resolution at an IDR picture)
    bool has_data = false, ready = false;
    bool fetch = false;                // this frame waits in device staging and jm_nvdec_output_frame copies it with one synchronous DMA (Decoder::init)
};

class HostCopier;

class Decoder {
public:
    Decoder();
    ~Decoder();
    int  init(int codec_type, int out_fmt, const uint8_t *extra, int len);
    int  decode(const uint8_t *buf, int len, int *got_frame);
    int  poll(int *got_frame, int wait_us = 0); // pop a finished display frame without feeding input (wait_us > 0: wait that long for one that is on its way)
    int  push(const uint8_t *buf, int len);    // feed input without popping a frame (the current frame stays current)
    int  push_eos();                           // end of stream without popping a frame
    int  output(uint8_t *out, int *out_len);
    int  output_device(void **dev, int *len);
    int  output_argb_device(void *dev_dst, int pitch);
    int  output_nv12_pitch_device(void *dev_dst, int pitch);
    int  stream_info(int *w, int *h) const;
    void set_eof(bool e) { eof_flag_ = e; }
    bool is_exit() const { return is_exit_; }
    char *info() { return info_; }
    const char *last_error();
    int  set_option(const char *key, long long v);
    long long get_stat(const char *key) const;
    void set_device(int d) { device_ = d; }
    int  numa_node() const { return numa_node_; }   // NUMA node of this handle's GPU (-1: unknown / one-node host): which parse pool it uses
    void api_exception(const char *what) { fail(std::string("exception in the decoder: ") + what); }
    void parse_exception(PicTask *t, const char *what);   // worker-pool: an exception escaped parse_task

    // worker-pool entry
    void parse_task(PicTask *t, ParseScratch &scratch);
    // engine completion callback + engine-private per-decoder state
    void on_engine_done(const struct EnginePic &p, bool failed = false);
    void on_device_wait_error(int code);       // engine: a kernel's bounded wait gave up while this handle's picture was decoded
    struct EngineDecoderState &engine_state() { return *eng_state_; }
    // display frames that are decoded and packed and that the caller has not fetched yet (the engine asks: is this handle's next picture urgent?)
    int frames_done_unfetched() const { return done_unfetched_.load(std::memory_order_relaxed); }

private:
    // ---- front end (caller thread) ----
    void feed(const uint8_t *buf, size_t len);
    void flush_stream();
    void handle_nal(const uint8_t *nal, size_t len);
    bool start_picture(const SliceHeader &sh, const SeqParams &sps, const PicParamSet &pps);
    void add_slice(const SliceHeader &sh, std::vector<uint8_t> &&rbsp, size_t rbsp_len);
    void dispatch_pending();
    void build_ref_lists(const SliceHeader &sh, SliceTask &st);
    void mark_current(const SliceHeader &sh);
    int  compute_poc(const SliceHeader &sh, DpbPic &store);
    void build_field_ref_lists(const SliceHeader &sh, SliceTask &task);
    void build_frame_ref_lists(const SliceHeader &sh, SliceTask &task);
    void mark_current_field(const SliceHeader &sh);
    void store_done(int slot, std::vector<int> &out);
    void infer_frame(int frame_num);
    void bump_after_current(std::vector<int> &out);
    void flush_dpb(std::vector<int> &out);
    void push_task(std::unique_ptr<PicTask> t);
    bool activate(const SeqParams &sps);
    int  acquire_job_slot(bool big = false);
    int  pop_output(bool block, int wait_us = 0);
    void fail(const std::string &msg);
    void note_error(const std::string &msg);
    // ---- device side ----
    bool gpu_open();
    bool gpu_alloc_sequence();
    void gpu_free_sequence();
    void free_out_slots(bool all);
    // ---- HEVC front end (hevc_decoder.cpp) ----
    void hevc_handle_nal(const uint8_t *nal, size_t len);
    bool hevc_start_picture(const HevcSliceHeader &sh, int nal_type, int tid);
    bool hevc_activate(const HevcSps &sps);
    bool hevc_build_refs(const HevcSliceHeader &sh, HevcSliceRefs &refs);
    void hevc_dispatch_pending();
    void hevc_bump(std::vector<int> &out, bool all, bool use_fullness);
    void hevc_parse_task(PicTask *t);
    void hevc_fill_engine_pic(PicTask *t, struct EnginePic &ep);
    bool ensure_job_cap(JobSlot &js, size_t bytes, size_t keep = 0);
    void gpu_close();
    void submit_ready();
    void submit_task(PicTask *t);
    void enqueue_output(int slot, std::vector<PackJob> &jobs, std::vector<OutSlot *> &slots);
    OutSlot *alloc_out_slot();

    // configuration
    int codec_ = 0, out_fmt_ = 1, device_ = -1, handle_index_ = 0, last_surf_ = -1, out_route_ = 0, fetch_limit_ = 1;
    bool chain_ok_ = false, chain_intra_on_ = true;
    bool parse_only_ = false, want_digest_ = false, sync_mode_ = false, profile_ = false, out_via_copy_engine_ = true, device_output_ = false,
        out_fetch_ = true;
    std::string error_, error_out_; std::mutex error_m_;     // error_out_: what last_error() last handed out (see there)
    std::atomic<bool> failed_{false}; bool inited_ = false;

    // splitter
    std::vector<uint8_t> in_; size_t scan_ = 0; bool have_start_ = false; size_t nal_start_ = 0;
    int avcc_len_size_ = 0;                    // > 0: init got an avcC record; packets are length-prefixed NAL units of this many bytes

    // parameter sets / sequence
    ParamSets ps_;
    bool seq_active_ = false; SeqParams seq_;
    int mb_w_ = 0, mb_h_ = 0, disp_w_ = 0, disp_h_ = 0, dpb_size_ = 1, reorder_depth_ = 0, n_surf_ = 0;

    // DPB / picture state (front end only)
    DpbPic dpb_[kMaxSurfaces];
    int cur_ = -1;
    int cur_field_ = 0; bool cur_second_ = false;   // the current picture: 0 frame, 1 top field, 2 bottom field; the second field of its frame
    int prev_ref_frame_num_ = 0;                    // PrevRefFrameNum (7.4.3)
    std::atomic<long long> stat_inferred_frames_{0}, stat_redundant_slices_{0};
    int pending_first_ = -1;                        // the frame store that holds a first field and waits for the second
    std::atomic<long long> stat_field_pics_{0}, stat_lone_fields_{0};
    std::unique_ptr<PicTask> pending_;
    SliceHeader first_sh_;
    int prev_poc_msb_ = 0, prev_poc_lsb_ = 0, prev_frame_num_ = 0; long long prev_frame_num_offset_ = 0; bool prev_mmco5_ = false;
    int numa_node_ = -1;
    long long cur_top_poc_ = 0, cur_bot_poc_ = 0;   // TopFieldOrderCnt / BottomFieldOrderCnt of the current picture (pic_order_cnt_type 0)
    int decode_count_ = 0, max_lt_idx_ = -1;
    uint64_t next_seq_ = 0;
    std::vector<int> carry_out_;               // outputs decided before the next picture starts (IDR flush)

    // task pipeline
    std::mutex mtx_;                           // guards inflight_, job slots, out queue
    std::condition_variable cv_;
    std::deque<std::unique_ptr<PicTask>> inflight_;
    std::mutex submit_mtx_;
    JobSlot jobs_[kMaxJobSlots]; int n_jobs_ = kJobSlots;      // the first n_jobs_ are in use (option "job_slots", before init)
    bool n_jobs_set_ = false;                                  // ... chosen by the caller; else by picture size (gpu_alloc_sequence)
    BigJobBuf big_[kBigJobBufs];               // see acquire_job_slot
    void free_job_buffers();
    std::deque<OutSlot *> ready_;              // display order
    std::atomic<int> done_unfetched_{0};       // entries of ready_ whose samples are there (OutSlot::ready)
    std::vector<OutSlot *> free_out_, all_out_;
    OutSlot *cur_out_ = nullptr;
    int outstanding_ = 0, parse_pending_ = 0;                      // tasks pushed and not yet submitted

    // device
    struct EngineDecoderState *eng_state_ = nullptr;
    class Engine *engine_ = nullptr;          // per-device executor (engine.h): the only place device work is issued
    uint8_t *surf_[kMaxSurfaces] = {nullptr};
    uint8_t *surf_block_ = nullptr;            // ONE device allocation holds every surface (surf_[i] = block + i * stride): the chain kernels address all of
                                               // a picture's references through one buffer descriptor (recon_device.h RefBuf)
    void free_surfaces();
    bool use_lds_deblock_ = false;
    uint8_t *resid_ = nullptr; bool use_lds_intra_ = false; bool lds_intra8_ = false;
    // HEVC: pre-SAO work surfaces (resid_ holds as many residual scratches)
    uint8_t *hevc_work_[kHevcWorkSets] = {nullptr, nullptr, nullptr, nullptr}; unsigned hevc_work_rr_ = 0;
    // HEVC boundary strengths on the device (round 5): per work set the 4x4-cell maps and the strength arrays (hevc_jobs.h HevcPicParams)
    uint8_t *hevc_bs_ = nullptr; size_t hevc_bs_set_bytes_ = 0, hevc_bs_off_[4] = {0, 0, 0, 0};
    int pitch_ = 0, chroma_off_ = 0; size_t surf_bytes_ = 0, frame_bytes_ = 0, job_cap_ = 0, job_cap_max_ = 0;
    std::atomic<size_t> i_job_peak_{0};            // largest job list of an I picture of this handle so far (+ slack): what a slot grows to for the next one
    std::atomic<long long> stat_job_regrown_{0};   // job slots grown (a few per handle while the slots reach their working size)
    bool gpu_open_ = false;

    // status / stats
    bool eof_flag_ = false, is_exit_ = false;
    std::atomic<bool> eos_sent_{false};        // (set by the thread that feeds, read by the thread that takes frames: jm_amddec_push_eos)
    uint32_t num_frames_ = 0;
    std::chrono::steady_clock::time_point t0_; bool timer_started_ = false; double elapsed_ms_ = 0;
    char info_[1024];
    std::atomic<long long> stat_parse_ns_i_{0}, stat_parse_ns_p_{0}, stat_submit_ns_{0}, stat_wait_slot_ns_{0};
    std::atomic<long long> stat_pictures_{0}, stat_job_bytes_{0}, stat_errors_{0}, stat_intra_mbs_{0}, stat_coef_{0}, stat_wait_errors_{0};
    SyntaxDigest digest_;
    bool fast_parse_ = true, want_job_digest_ = false;
    // output route "direct" (host_copy.h)
    std::atomic<long long> stat_direct_{0}, stat_direct_ns_{0};
    HostCopier *copier_ = nullptr; uint64_t out_sig_ = 0;
    // a frame goes out only while this many pictures of the handle are still on their way (option "display_delay", JM_AMD_DEC_DISPLAY_DELAY)
    int display_delay_ = 0;
    uint64_t job_digest_ = 1469598103934665603ull;    // option "job_digest" (tests)
    // HEVC state (front end only unless noted)
    HevcParamSets hps_; HevcSps hsps_; HevcPps hpps_;
    int h_poc_tid0_ = 0, h_max_dpb_ = 1, h_reorder_ = 0, extra_surf_ = 0; bool h_first_picture_ = true, h_no_rasl_output_ = false, h_seen_eos_ = false;
    HevcSliceHeader h_last_sh_; bool h_have_last_sh_ = false;
    HevcDigest hdigest_;                       // written by the parse worker (sync option)
    long long stat_i_ = 0, stat_p_ = 0, stat_b_ = 0;
    std::vector<int> display_pocs_;            // diagnostic (get via stats)
    struct TraceRec { uint64_t seq; long long t_dispatch, t_parsed, t_submit0, t_submit1; int is_i; };
    std::vector<TraceRec> trace_; bool trace_on_ = false;
};

// process-wide parse worker pool
void pool_submit(Decoder *d, PicTask *t);
int  pool_threads(int numa_node);

}  // namespace jmamd
