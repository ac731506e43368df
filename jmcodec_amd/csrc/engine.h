// jmcodec_amd/csrc/engine.h -- per-device execution engine: batches one picture per stream into every kernel launch.
//
// Why: single-picture kernels of the serial stages (intra / deblock wavefronts) occupy 2 CUs for ~1 ms, and the GPU
// front end only keeps a handful of HW queues running concurrently, so "one HIP stream per decoder" tops out at ~5
// pictures in flight on a 256-CU part.  The engine instead owns ONE in-order stream per device; each round it takes the
// next ready picture of every decoder (pictures of one stream depend on each other, pictures of different streams do
// not -- SURVEY.md 8e), uploads nothing (job lists were copied when parsed) and issues one batched launch per stage with
// blockIdx.y = picture.  A 16-stream batch is 16x the work per launch at the same latency.
// There is no reference counterpart: the reference drives one NVDEC session synchronously (nv_dec.cpp:33-41).
#pragma once
#include "jobs.h"
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

struct ihipStream_t; struct ihipEvent_t;

namespace jmamd {

class Decoder;
struct OutSlot;

struct EnginePic {
    Decoder *dec = nullptr;
    bool has_picture = false;
    PicParams pp;                                   // device pointers already resolved by the decoder
    int job_slot = -1;
    uint8_t *job_host = nullptr, *job_dev = nullptr; size_t upload_bytes = 0;
    std::vector<PackJob> out_before, out_after;     // display frames to pack before / after this picture's kernels
    std::vector<OutSlot *> slots_before, slots_after;
    int mb_w = 0, mb_h = 0, disp_w = 0, disp_h = 0;
    bool wait_prev_pack = false;                    // this picture reuses a surface whose pack-out may still be running
    long long alg_bytes[4] = {0, 0, 0, 0};          // algorithmic bytes of this picture per kernel class (recon, intra, deblock, packout)
};

constexpr int kMaxBatch = 64;
constexpr int kBatchRing = 4;

struct EngineStats {                                // per kernel class: 0 recon_inter, 1 intra, 2 deblock (prep+lds), 3 packout
    double ns[4] = {0, 0, 0, 0}; long long launches[4] = {0, 0, 0, 0}, pics[4] = {0, 0, 0, 0}, alg_bytes[4] = {0, 0, 0, 0};
    long long batches = 0, batch_pics = 0;
};

class Engine {
public:
    static Engine *get(int device);                 // creates the engine (and its thread) on first use; nullptr on HIP failure
    void submit(EnginePic &&p);                     // decode order per decoder; thread-safe
    void upload(uint8_t *dev, const uint8_t *host, size_t n);   // async H2D on the engine's copy stream (called when a picture is parsed)
    void set_profile(bool on) { profile_ = on; }
    EngineStats stats();
    int device() const { return device_; }

private:
    explicit Engine(int device);
    void run();
    struct Batch {
        PicParams *h_pics = nullptr, *d_pics = nullptr;       // pinned host / device, kMaxBatch entries
        PackJob *h_jobs = nullptr, *d_jobs = nullptr;         // 4 * kMaxBatch entries
        ihipEvent_t *done = nullptr, *kdone = nullptr, *upl = nullptr, *pev[8] = {nullptr};
        std::vector<EnginePic> pics;
        int n_pre = 0, n_post = 0; bool busy = false; unsigned pmask = 0;
        long long alg[4] = {0, 0, 0, 0}; int npics[4] = {0, 0, 0, 0};
    };
    void launch(Batch &b);
    void complete(Batch &b);

    int device_;
    ihipStream_t *stream_ = nullptr, *copy_stream_ = nullptr, *pack_stream_ = nullptr;   // decode kernels | job uploads | pack-out (PCIe writes)
    ihipEvent_t *pack_hist_[2] = {nullptr, nullptr};   // done events of the two most recently launched batches
    std::mutex m_; std::condition_variable cv_;
    std::deque<EnginePic> pending_;
    Batch ring_[kBatchRing];
    int head_ = 0, tail_ = 0, inflight_ = 0;
    bool profile_ = false, ok_ = false;
    std::mutex sm_; EngineStats st_;
    std::thread th_;
};

}  // namespace jmamd
