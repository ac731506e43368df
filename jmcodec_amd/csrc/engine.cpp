// jmcodec_amd/csrc/engine.cpp -- see engine.h.
#include "engine.h"
#include "decoder.h"
#include "kernels.h"
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace jmamd {

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

Engine::Engine(int device) : device_(device) {
    if (hipSetDevice(device_) != hipSuccess) return;
    hipStream_t s, c, p;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&p, hipStreamNonBlocking) != hipSuccess) return;
    stream_ = s; copy_stream_ = c; pack_stream_ = p;
    for (auto &b : ring_) {
        if (hipHostMalloc((void **)&b.h_pics, sizeof(PicParams) * kMaxBatch, hipHostMallocDefault) != hipSuccess) return;
        if (hipMalloc((void **)&b.d_pics, sizeof(PicParams) * kMaxBatch) != hipSuccess) return;
        if (hipHostMalloc((void **)&b.h_jobs, sizeof(PackJob) * 4 * kMaxBatch, hipHostMallocDefault) != hipSuccess) return;
        if (hipMalloc((void **)&b.d_jobs, sizeof(PackJob) * 4 * kMaxBatch) != hipSuccess) return;
        if (hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) return;
        if (hipEventCreateWithFlags(&b.upl, hipEventDisableTiming) != hipSuccess) return;
        if (hipEventCreateWithFlags(&b.kdone, hipEventDisableTiming) != hipSuccess) return;
        for (auto &e : b.pev) if (hipEventCreate(&e) != hipSuccess) return;
    }
    ok_ = true;
    th_ = std::thread([this] { run(); });
    th_.detach();
}

void Engine::upload(uint8_t *dev, const uint8_t *host, size_t n) {
    hipSetDevice(device_);
    hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, copy_stream_);
}

void Engine::submit(EnginePic &&p) {
    { std::lock_guard<std::mutex> lk(m_); pending_.push_back(std::move(p)); }
    cv_.notify_one();
}

EngineStats Engine::stats() { std::lock_guard<std::mutex> lk(sm_); return st_; }

// one batched launch per stage; everything on the engine's single in-order stream
void Engine::launch(Batch &b) {
    const int n = (int)b.pics.size();
    int max_mbs = 0, max_mb_h = 0, max_w = 0, max_h = 0, stages = 0;
    bool wait_pack = false;
    b.n_pre = b.n_post = 0; b.pmask = 0;
    for (int k = 0; k < 4; k++) { b.alg[k] = 0; b.npics[k] = 0; }
    // pack jobs: [0, n_pre) before the decode kernels, [2*kMaxBatch, 2*kMaxBatch + n_post) after them
    for (int i = 0; i < n; i++) {
        EnginePic &p = b.pics[i];
        b.h_pics[i] = p.pp;
        if (!p.has_picture) b.h_pics[i].stages = 0;
        stages |= b.h_pics[i].stages;
        if (p.has_picture) { max_mbs = std::max(max_mbs, p.mb_w * p.mb_h); max_mb_h = std::max(max_mb_h, p.mb_h); }
        if (p.wait_prev_pack) wait_pack = true;
        for (auto &j : p.out_before) if (b.n_pre < 2 * kMaxBatch) b.h_jobs[b.n_pre++] = j;
        for (auto &j : p.out_after) if (b.n_post < 2 * kMaxBatch) b.h_jobs[2 * kMaxBatch + b.n_post++] = j;
        if (!p.out_before.empty() || !p.out_after.empty()) { max_w = std::max(max_w, p.disp_w); max_h = std::max(max_h, p.disp_h); }
        int st = b.h_pics[i].stages;
        if (st & PS_RECON) { b.alg[0] += p.alg_bytes[0]; b.npics[0]++; }
        if (st & (PS_INTRA_LDS | PS_INTRA_V1)) { b.alg[1] += p.alg_bytes[1]; b.npics[1]++; }
        if (st & (PS_DEBLOCK_LDS | PS_DEBLOCK_V1)) { b.alg[2] += p.alg_bytes[2]; b.npics[2]++; }
        b.alg[3] += p.alg_bytes[3] * (long long)(p.out_before.size() + p.out_after.size());
        b.npics[3] += (int)(p.out_before.size() + p.out_after.size());
    }
    hipMemcpyAsync(b.d_pics, b.h_pics, sizeof(PicParams) * n, hipMemcpyHostToDevice, stream_);
    if (b.n_pre) hipMemcpyAsync(b.d_jobs, b.h_jobs, sizeof(PackJob) * b.n_pre, hipMemcpyHostToDevice, stream_);
    if (b.n_post) hipMemcpyAsync(b.d_jobs + 2 * kMaxBatch, b.h_jobs + 2 * kMaxBatch, sizeof(PackJob) * b.n_post, hipMemcpyHostToDevice, stream_);
    // job lists were copied on the copy stream when the pictures were parsed: one event covers them all
    hipEventRecord(b.upl, copy_stream_);
    hipStreamWaitEvent(stream_, b.upl, 0);
    // Pack-out of batch k runs on its own stream and overlaps the decode kernels of batch k+1 (PCIe writes vs. compute).
    // The decoder never reuses a displayed surface for the very next picture (DPB cooling, decoder.cpp), so the decode
    // kernels of this batch only have to wait for the pack-out launched TWO batches ago.
    if (pack_hist_[1]) hipStreamWaitEvent(stream_, pack_hist_[1], 0);
    if ((wait_pack || b.n_pre) && pack_hist_[0]) hipStreamWaitEvent(stream_, pack_hist_[0], 0);
    auto mark = [&](int i, hipStream_t s) { if (profile_) hipEventRecord(b.pev[i], s); };
    mark(0, stream_);
    if (b.n_pre) { launch_packout(b.d_jobs, b.n_pre, max_w, max_h, stream_); b.pmask |= 1; }
    mark(1, stream_);
    if (stages & PS_RECON) { launch_recon_inter(b.d_pics, n, max_mbs, stream_); b.pmask |= 2; }
    mark(2, stream_);
    if (stages & PS_INTRA_LDS) { launch_intra_lds(b.d_pics, n, max_mb_h, stream_); b.pmask |= 4; }
    if (stages & PS_INTRA_V1) { launch_recon_intra(b.d_pics, n, stream_); b.pmask |= 4; }
    mark(3, stream_);
    if (stages & PS_DEBLOCK_LDS) { launch_deblock_lds(b.d_pics, n, max_mbs, max_mb_h, stream_); b.pmask |= 8; }
    if (stages & PS_DEBLOCK_V1) { launch_deblock(b.d_pics, n, stream_); b.pmask |= 8; }
    mark(4, stream_);
    hipEventRecord(b.kdone, stream_);
    hipStreamWaitEvent(pack_stream_, b.kdone, 0);
    mark(5, pack_stream_);
    if (b.n_post) { launch_packout(b.d_jobs + 2 * kMaxBatch, b.n_post, max_w, max_h, pack_stream_); b.pmask |= 16; }
    mark(6, pack_stream_);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) fprintf(stderr, "jm_amd_dec: kernel launch failed: %s\n", hipGetErrorString(le));
    hipEventRecord(b.done, pack_stream_);
    pack_hist_[1] = pack_hist_[0]; pack_hist_[0] = b.done;
    b.busy = true;
}

void Engine::complete(Batch &b) {
    if (profile_) {
        std::lock_guard<std::mutex> lk(sm_);
        auto add = [&](int cls, int e0, int e1, bool ran) {
            if (!ran) return;
            float ms = 0;
            if (hipEventElapsedTime(&ms, b.pev[e0], b.pev[e1]) == hipSuccess) { st_.ns[cls] += ms * 1e6; st_.launches[cls]++; }
        };
        add(3, 0, 1, b.pmask & 1); add(0, 1, 2, b.pmask & 2); add(1, 2, 3, b.pmask & 4); add(2, 3, 4, b.pmask & 8); add(3, 5, 6, b.pmask & 16);
        for (int k = 0; k < 4; k++) { st_.pics[k] += b.npics[k]; st_.alg_bytes[k] += b.alg[k]; }
        st_.batches++; st_.batch_pics += (long long)b.pics.size();
    }
    for (auto &p : b.pics) p.dec->on_engine_done(p);
    b.pics.clear();
    b.busy = false;
}

void Engine::run() {
    hipSetDevice(device_);
    for (;;) {
        Batch *next = nullptr;
        {
            std::unique_lock<std::mutex> lk(m_);
            if (pending_.empty() && inflight_ == 0) cv_.wait(lk, [&] { return !pending_.empty(); });
            if (!pending_.empty() && inflight_ < kBatchRing - 1) {
                // one picture per decoder, in arrival order
                next = &ring_[head_];
                next->pics.clear();
                std::vector<Decoder *> seen;
                for (auto it = pending_.begin(); it != pending_.end() && (int)next->pics.size() < kMaxBatch;) {
                    if (std::find(seen.begin(), seen.end(), it->dec) != seen.end()) { ++it; continue; }
                    seen.push_back(it->dec);
                    next->pics.push_back(std::move(*it));
                    it = pending_.erase(it);
                }
            }
        }
        if (next) {
            launch(*next);
            head_ = (head_ + 1) % kBatchRing; inflight_++;
            // keep at most two batches queued on the device: while they run, new pictures pile up and the next batch is full
            if (inflight_ < 2) continue;
        }
        if (inflight_ > 0) {
            Batch &old = ring_[tail_];
            hipEventSynchronize(old.done);
            complete(old);
            tail_ = (tail_ + 1) % kBatchRing; inflight_--;
        }
    }
}

}  // namespace jmamd
