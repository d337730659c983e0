// fl_queue.cpp -- the persistent request-batching queue behind flgpu_transform and the sharding of every batch across
// the devices of a node.
//
// Reference analogue: one Arc<State> shared by every tokio worker (src/main.rs:108-112), each request processed on the
// worker that received it (src/main.rs:179).  Here concurrent callers of flgpu_transform are packed into shared kernel
// launches; a context that spans several GPUs splits every flushed batch -- and every batch entry point -- into one
// contiguous shard per device, balanced by algorithmic bytes (W*H*C + output bytes), and returns results in
// request order.  Images are independent, so no pixel ever crosses between devices.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>

#include "fl_context.h"

using namespace fl;

namespace fl {

void split_by_weight(const uint64_t *weight, size_t n, uint32_t n_shards, uint32_t *shard_of)
{
    if (n_shards == 0) n_shards = 1;
    unsigned __int128 total = 0;
    for (size_t i = 0; i < n; ++i) total += weight[i] ? weight[i] : 1;
    // image i goes to the shard that the midpoint of its weight interval falls into: monotone, hence contiguous
    unsigned __int128 before = 0;
    uint32_t prev = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint64_t w = weight[i] ? weight[i] : 1;
        const unsigned __int128 mid2 = 2 * before + w; // twice the midpoint
        uint32_t s = (uint32_t)((mid2 * n_shards) / (2 * total));
        if (s >= n_shards) s = n_shards - 1;
        if (s < prev) s = prev;
        shard_of[i] = s;
        prev = s;
        before += w;
    }
}

// Pixel buffers as sources: a flush stops collecting at this many source bytes per device.  Sixteen 1080p buffers are 100 MB of uploads in front of
// the batch's first kernel, and with four lanes' uploads interleaved on the link a request waited for up to 400 MB: the leg is bound by the link either way
// (9.1 k images/s), but its p99 went 12-20 -> 8.7-8.9 ms with batches of three (tools/experiments/queue_bytes_cap.sh: 12 / 16 / 24 / 32 MB / none:
// 8.7 / 8.9 / 9.3-9.6 / 9.7-10.1 / 12-20 ms).  JPEG files do not count: their uploads are small and their decode kernels' time does not depend on the batch.
#ifndef FL_QUEUE_CAP_MB
#define FL_QUEUE_CAP_MB 16
#endif
constexpr uint64_t kBatchSourceBytes = FL_QUEUE_CAP_MB ? (uint64_t)FL_QUEUE_CAP_MB << 20 : ~0ull >> 8;

constexpr size_t kPinPoolMaxBlock = (size_t)64 << 20; // page-locked staging blocks above this size are freed after use, not pooled

PinBlock pin_acquire(flgpu_ctx *c, size_t bytes)
{
    size_t cap = 64 * 1024;
    while (cap < bytes) cap <<= 1;
    {
        std::lock_guard<std::mutex> g(c->pin_mu);
        auto it = c->pin_free.find(cap);
        if (it != c->pin_free.end()) { PinBlock b{it->second, cap}; c->pin_free.erase(it); return b; }
    }
    PinBlock b;
    (void)hipSetDevice(c->device);
    // portable: with several devices any of them may DMA from / into the block
    if (hipHostMalloc(&b.p, cap, c->devices.size() > 1 ? hipHostMallocPortable : hipHostMallocDefault) != hipSuccess) { b.p = nullptr; return b; }
    b.cap = cap;
    return b;
}

void pin_release(flgpu_ctx *c, PinBlock &b)
{
    if (!b.p) return;
    if (b.cap > kPinPoolMaxBlock) { (void)hipHostFree(b.p); b.p = nullptr; return; } // a rare huge picture must not pin its block for the context's lifetime
    std::lock_guard<std::mutex> g(c->pin_mu);
    c->pin_free.emplace(b.cap, b.p);
    b.p = nullptr;
}

} // namespace fl

namespace {

// CPUs this process may really use: the cgroup's quota where there is one (a container's share of a big host), else the affinity mask.
uint32_t usable_cpus()
{
    uint32_t n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = (uint32_t)CPU_COUNT(&set);
    if (!n) n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota> <period>" or "max <period>"
        long long quota = 0, period = 0;
        if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period - 1) / period));
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { // cgroup v1
        long long quota = 0, period = 100000;
        if (fscanf(g, "%lld", &quota) == 1 && quota > 0) {
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
            n = std::min<uint32_t>(n, (uint32_t)std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(g);
    }
    return std::max(1u, n);
}

uint32_t lanes_per_device(const flgpu_ctx *c) { return std::min<uint32_t>(std::max<uint32_t>(c->cfg.queue_lanes ? c->cfg.queue_lanes : 4u, 1u), 8u); }
// Overflow lanes (round 5): with the default shape, four more lanes per device that only ever take a FULL batch that is waiting while the four regular
// lanes are busy.  At 64 callers they sleep (the regular lanes' 4 x 16 places hold everybody: more lanes there only made smaller batches,
// profiles/r05_queue_shape.txt); at 128 callers the queue holds full batches and the device has room for them: 30.3 k -> 36-37 k files/s
// (tools/experiments/jh_capacity.sh).  A caller that sets queue_lanes gets exactly that many lanes and no overflow lanes.
uint32_t overflow_lanes_per_device(const flgpu_ctx *c) { return c->cfg.queue_lanes ? 0u : 4u; }
uint32_t batch_per_device(const flgpu_ctx *c) { return c->cfg.max_batch ? c->cfg.max_batch : 16u; } // measured (round 5, tools/experiments/jh_lanes.sh, 64 callers on 16 cores): 4 lanes x 16 against round 4's 3 x 32 -- the same or more requests per second for files, pixels and blurred pixels, p99 14-19 ms -> 12-14 ms; equal at 128 callers

// One shard of a flushed batch on one lane: sources already sit in pinned blocks (copied there by the calling
// threads), results are left in pinned blocks for the callers to copy out.
int run_batch_queued(flgpu_ctx *c, std::vector<Request *> &batch)
{
    const size_t n = batch.size();
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    hipStream_t st = c->stream;
    // Whatever happens below, nothing of this batch may still be in flight when the requests are marked done: their
    // callers recycle (or free) the pinned blocks the DMA engine reads and writes.
    struct Drain { hipStream_t st; ~Drain() { (void)hipStreamSynchronize(st); } } drain{st};
    std::vector<flgpu_image> dsrc(n), ddst(n);
    std::vector<flgpu_params> ps(n);
    std::vector<uint64_t> dev_out(n);
    std::vector<JpegSrc> jhp(n);
    size_t in_b = 0, out_b = 0;
    for (size_t i = 0; i < n; ++i) {
        dsrc[i] = *batch[i]->src; ddst[i] = *batch[i]->dst; ps[i] = *batch[i]->p;
        flgpu_plan plan;
        int prc = flgpu_plan_output(&ps[i], dsrc[i].width, dsrc[i].height, dsrc[i].channels, &plan);
        if (prc) return prc; // validated by the caller already
        dev_out[i] = staged_out_bytes(ps[i], plan, 0); // an encoded stream gets the format's worst case on the device
        dsrc[i].data = reinterpret_cast<uint8_t *>(in_b); dsrc[i].capacity = batch[i]->src_bytes; in_b += align_up(batch[i]->src_bytes, 256);
        ddst[i].data = reinterpret_cast<uint8_t *>(out_b); ddst[i].capacity = dev_out[i]; out_b += align_up(dev_out[i], 256);
    }
    FL_HIP(c, c->d_in.reserve(in_b), "device input staging");
    FL_HIP(c, c->d_out.reserve(out_b), "device output staging");
    for (size_t i = 0; i < n; ++i) {
        dsrc[i].data = static_cast<uint8_t *>(c->d_in.p) + reinterpret_cast<size_t>(dsrc[i].data);
        ddst[i].data = static_cast<uint8_t *>(c->d_out.p) + reinterpret_cast<size_t>(ddst[i].data);
        FL_HIP(c, hipMemcpyAsync(dsrc[i].data, batch[i]->in.p, batch[i]->src_bytes, hipMemcpyHostToDevice, st), "H2D");
        if (batch[i]->jpeg) {
            jhp[i].hdr = &batch[i]->jhdr; jhp[i].stage = batch[i]->jstage; jhp[i].icc = batch[i]->icc.empty() ? nullptr : batch[i]->icc.data(); jhp[i].icc_len = batch[i]->icc.size();
            c->stats.jpeg_file_bytes += batch[i]->file_bytes;
        }
    }
    { int drc = decode_jpeg_sources(c, n, dsrc.data(), jhp.data(), st); if (drc) return drc; }
    int rc = run_batch_device(c, n, dsrc.data(), ps.data(), false, ddst.data(), st);
    if (rc) return rc;
    // Encoded streams: their lengths are known only on the device (a 300x200 JPEG is ~16 KB of a 183 KB bound).  The first kSpecBytes of every stream
    // are fetched together with the lengths, in front of the wait -- most streams of the sizes a thumbnail service sends are whole then --
    // and only what is longer is fetched behind it.  (Until round 5: lengths, wait, one copy per request, wait -- 0.16 ms of the lane's 1.8 ms
    // per batch spent issuing copies with the device idle, tools/experiments/jh_hiptrace.sh.)
    constexpr uint64_t kSpecBytes = 32u << 10;
    bool encoded = false;
    for (size_t i = 0; i < n; ++i) encoded |= ps[i].front_end == FLGPU_FE_JPEG;
    std::vector<uint64_t> spec(n, 0);
    if (encoded) {
        for (size_t i = 0; i < n; ++i) {
            if (ps[i].front_end != FLGPU_FE_JPEG) continue;
            spec[i] = std::min<uint64_t>({kSpecBytes, dev_out[i], batch[i]->out_bytes});
            if (spec[i]) FL_HIP(c, hipMemcpyAsync(batch[i]->out.p, ddst[i].data, spec[i], hipMemcpyDeviceToHost, st), "D2H");
        }
    }
    const bool jh_fetched = encoded;
    if (jh_fetched) { int frc = entropy_failures_fetch(c, n, st); if (frc) return frc; }
    int rrc = FLGPU_OK;
    if (encoded) rrc = collect_results(c, n, ddst.data(), st);
    if (rrc == FLGPU_ERR_DEVICE) return rrc;
    bool more = false;
    for (size_t i = 0; i < n; ++i) {
        const bool jpeg = ps[i].front_end == FLGPU_FE_JPEG;
        const uint64_t nb = jpeg ? ddst[i].bytes : batch[i]->out_bytes;
        if (jpeg && nb > batch[i]->out_bytes) { batch[i]->status = FLGPU_ERR_BUFFER_TOO_SMALL; ddst[i].bytes = 0; continue; } // the caller's dst really is too small
        if (nb > spec[i]) {
            FL_HIP(c, hipMemcpyAsync(static_cast<uint8_t *>(batch[i]->out.p) + spec[i], ddst[i].data + spec[i], nb - spec[i], hipMemcpyDeviceToHost, st), "D2H");
            more = true;
        }
    }
    if (!encoded) rrc = collect_results(c, n, ddst.data(), st);
    else if (more) FL_HIP(c, hipStreamSynchronize(st), "batch sync");
    for (size_t i = 0; i < n; ++i) {
        batch[i]->dst->width = ddst[i].width; batch[i]->dst->height = ddst[i].height;
        batch[i]->dst->channels = ddst[i].channels; batch[i]->dst->flags = ddst[i].flags;
        batch[i]->dst->bytes = ddst[i].bytes;
        if (ps[i].front_end == FLGPU_FE_JPEG && !ddst[i].bytes && batch[i]->status == FLGPU_OK) batch[i]->status = FLGPU_ERR_BUFFER_TOO_SMALL;
    }
    if (rrc == FLGPU_ERR_DEVICE) return rrc; // the device error word: no result of this batch is valid
    {   // requests whose file the device entropy decoder gave up on go back to their callers, who decode on the host and queue again
        std::vector<uint8_t> bad;
        const int nbad = entropy_failures(c, n, bad, st, jh_fetched);
        if (nbad < 0) return -nbad;
        for (size_t i = 0; i < n && nbad; ++i) if (bad[i]) batch[i]->status = FL_STATUS_RETRY_HOST_HUFFMAN;
    }
    return FLGPU_OK; // otherwise the per-request status above: one oversized stream must not fail its batch mates
}

// `slot` = index of this lane's device in the context's device list (0 for a single-device context)
void worker_main(flgpu_ctx *c, flgpu_ctx *lane, uint32_t slot, bool overflow)
{
    const uint32_t ndev = c->n_dev();
    const size_t max_batch = (size_t)batch_per_device(c) * ndev;
    const auto flush = std::chrono::microseconds(c->cfg.flush_timeout_us ? c->cfg.flush_timeout_us : 200);
    for (;;) {
        std::vector<Request *> batch;
        {
            std::unique_lock<std::mutex> lk(c->qmu);
            c->qcv.wait(lk, [&] { return c->stop || !c->inbox[slot].empty() || (!c->collecting && (overflow ? c->queue.size() >= max_batch : !c->queue.empty())); });
            if (!c->inbox[slot].empty()) {
                // a shard another worker cut for this device
                batch.swap(c->inbox[slot].front());
                c->inbox[slot].pop_front();
            } else {
                if (c->queue.empty()) { if (c->stop) return; continue; }
                if (c->collecting) continue;
                if (overflow && !c->stop && c->queue.size() < max_batch) continue; // (shutting down, it drains the queue like any lane)
                c->collecting = true;
                // a first request arrived: wait for company -- but only while somebody is actually on the way (a caller
                // staging its source), and never beyond the flush timer or a full batch.  A lone caller is served at once.
                const auto deadline = std::chrono::steady_clock::now() + flush;
                while (c->queue.size() < max_batch && !c->stop && c->staging.load(std::memory_order_acquire) > 0) {
                    if (c->qcv.wait_until(lk, deadline) == std::cv_status::timeout) break;
                }
                std::vector<Request *> all;
                // (a batch also ends at kBatchSourceBytes of pixel sources, see above)
                for (uint64_t bytes = 0; !c->queue.empty() && all.size() < max_batch && (all.empty() || bytes < kBatchSourceBytes * ndev);) {
                    if (!c->queue.front()->jpeg) bytes += c->queue.front()->src_bytes;
                    all.push_back(c->queue.front()); c->queue.pop_front();
                }
                c->collecting = false;
                if (ndev <= 1 || all.size() <= 1) batch.swap(all);
                else {
                    // one contiguous shard per device, balanced by algorithmic bytes; this worker keeps its own device's
                    // shard, the others go to the inboxes of the other devices' lanes.  Requests complete individually,
                    // so "request order" is preserved by construction: each caller waits on its own slot.
                    std::vector<uint64_t> w(all.size());
                    std::vector<uint32_t> shard_of(all.size());
                    for (size_t i = 0; i < all.size(); ++i) w[i] = all[i]->weight;
                    split_by_weight(w.data(), all.size(), ndev, shard_of.data());
                    std::vector<std::vector<Request *>> shards(ndev);
                    for (size_t i = 0; i < all.size(); ++i) shards[shard_of[i]].push_back(all[i]);
                    // a small flush may leave this device's shard empty: then take the first non-empty one
                    uint32_t keep = slot;
                    if (shards[keep].empty()) for (uint32_t k = 0; k < ndev; ++k) if (!shards[k].empty()) { keep = k; break; }
                    for (uint32_t k = 0; k < ndev; ++k) {
                        if (shards[k].empty()) continue;
                        if (k == keep) batch.swap(shards[k]);
                        else c->inbox[k].push_back(std::move(shards[k]));
                    }
                }
            }
        }
        c->qcv.notify_all(); // the next batch may be collected (and the other devices' shards picked up) while this one is in flight
        if (batch.empty()) continue;
        int rc;
        {
            std::lock_guard<std::mutex> g(lane->mu);
            rc = run_batch_queued(lane, batch);
            lane->stats.queue_flushes++;
            if (rc) c->set_error(lane->get_error());
        }
        {
            std::lock_guard<std::mutex> lk(c->qmu);
            // (notified under the lock: a caller that has seen `done` returns and its Request, cv included, is gone.  Until round 5 one condition
            // variable for all callers: every finished batch woke all ~50 waiting threads to look at their flags, one after the other on qmu)
            for (Request *r : batch) { if (rc) r->status = rc; r->done = true; r->cv.notify_one(); }
        }
    }
}

// Starts the lanes on first use (qmu held).  Lane i serves device slot i % n_dev.
int start_workers(flgpu_ctx *c)
{
    if (c->worker_started) return FLGPU_OK;
    const uint32_t ndev = c->n_dev(), regular = lanes_per_device(c), per = std::min<uint32_t>(regular + overflow_lanes_per_device(c), 8u);
    for (uint32_t i = 0; i < per * ndev; ++i) {
        const uint32_t slot = i % ndev;
        flgpu_ctx *l = create_child(c, c->devices.empty() ? c->device : c->devices[slot]);
        if (!l) break;
        l->clut_owner = c->shard_ctx.empty() ? c : c->shard_ctx[slot]; // the configured CMYK table of this lane's device
        c->lanes.push_back(l); // capacity reserved at creation: published entries never move
        c->n_lanes.store(c->lanes.size(), std::memory_order_release);
    }
    if (c->lanes.size() < ndev) { // every device needs at least one lane, or its shards would never run
        for (flgpu_ctx *l : c->lanes) flgpu_destroy(l);
        c->n_lanes.store(0, std::memory_order_release);
        c->lanes.clear();
        return FLGPU_ERR_OOM;
    }
    for (size_t i = 0; i < c->lanes.size(); ++i) c->workers.emplace_back(worker_main, c, c->lanes[i], (uint32_t)(i % ndev), i / ndev >= regular);
    c->worker_started = true;
    return FLGPU_OK;
}

// Runs fn(k, first, last) for every non-empty shard on its own thread (shard 0 on the calling thread).
template <typename F> int for_each_shard(flgpu_ctx *c, const std::vector<std::pair<size_t, size_t>> &ranges, F fn)
{
    std::vector<int> rcs(ranges.size(), FLGPU_OK);
    std::vector<std::thread> ts;
    for (size_t k = 1; k < ranges.size(); ++k)
        if (ranges[k].second > ranges[k].first) ts.emplace_back([&, k] { rcs[k] = fn((uint32_t)k, ranges[k].first, ranges[k].second); });
    if (!ranges.empty() && ranges[0].second > ranges[0].first) rcs[0] = fn(0u, ranges[0].first, ranges[0].second);
    for (auto &t : ts) t.join();
    for (size_t k = 0; k < rcs.size(); ++k)
        if (rcs[k]) { c->set_error(c->shard_ctx[k]->get_error()); return rcs[k]; }
    return FLGPU_OK;
}

int shard_ranges(uint32_t n_shards, size_t n, const flgpu_image *srcs, const flgpu_params *ps, bool same_params,
                 std::vector<std::pair<size_t, size_t>> &ranges)
{
    std::vector<uint32_t> shard_of(n);
    int rc = flgpu_plan_shards(n_shards, n, srcs, ps, same_params ? FLGPU_BATCH_SAME_PARAMS : 0u, shard_of.data(), nullptr);
    if (rc) return rc;
    ranges.assign(n_shards, {0, 0});
    for (uint32_t k = 0; k < n_shards; ++k) ranges[k] = {n, n};
    for (size_t i = 0; i < n; ++i) {
        auto &r = ranges[shard_of[i]];
        if (r.first == n) r.first = i;
        r.second = i + 1;
    }
    for (auto &r : ranges) if (r.first == n) r = {0, 0};
    return FLGPU_OK;
}

} // namespace

namespace fl {

void stop_queue(flgpu_ctx *c)
{
    {
        std::lock_guard<std::mutex> lk(c->qmu);
        c->stop = true;
    }
    c->qcv.notify_all();
    for (auto &t : c->workers) t.join();
    c->workers.clear();
    const size_t nl = c->n_lanes.load(std::memory_order_acquire);
    for (size_t i = 0; i < nl; ++i) flgpu_destroy(c->lanes[i]);
    c->n_lanes.store(0, std::memory_order_release);
    c->lanes.clear();
}

} // namespace fl

extern "C" {

int flgpu_plan_shards(uint32_t n_shards, size_t n, const flgpu_image *srcs, const flgpu_params *ps, uint32_t flags,
                      uint32_t *shard_of, uint64_t *shard_bytes)
try {
    if (n_shards == 0 || n_shards > FLGPU_MAX_DEVICES || (n && (!srcs || !ps || !shard_of))) return FLGPU_ERR_INVALID_ARG;
    std::vector<uint64_t> w(n);
    for (size_t i = 0; i < n; ++i) {
        flgpu_plan plan;
        const flgpu_params *p = (flags & FLGPU_BATCH_SAME_PARAMS) ? &ps[0] : &ps[i];
        int rc = flgpu_plan_output(p, srcs[i].width, srcs[i].height, srcs[i].channels, &plan);
        if (rc) return rc;
        w[i] = (uint64_t)srcs[i].width * srcs[i].height * srcs[i].channels + plan.out_bytes;
    }
    split_by_weight(w.data(), n, n_shards, shard_of);
    if (shard_bytes) {
        for (uint32_t k = 0; k < n_shards; ++k) shard_bytes[k] = 0;
        for (size_t i = 0; i < n; ++i) shard_bytes[shard_of[i]] += w[i];
    }
    return FLGPU_OK;
} FL_ABI_CATCH

int flgpu_transform_batch_device(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts,
                                 void *hip_stream, uint32_t flags)
try {
    if (!c) return FLGPU_ERR_INVALID_ARG;
    const bool same = (flags & FLGPU_BATCH_SAME_PARAMS) != 0;
    if (c->shard_ctx.empty()) {
        std::lock_guard<std::mutex> g(c->mu);
        return run_batch_device(c, n, srcs, ps, same, dsts, static_cast<hipStream_t>(hip_stream));
    }
    // several devices: shard k of the batch runs on devices[k], on that shard context's own stream; the call returns
    // when every shard has finished (a caller's stream belongs to one device and cannot order the others)
    if (n == 0) return FLGPU_OK;
    if (!srcs || !ps || !dsts) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    std::vector<std::pair<size_t, size_t>> ranges;
    int rc = shard_ranges((uint32_t)c->shard_ctx.size(), n, srcs, ps, same, ranges);
    if (rc) return rc;
    for (size_t k = 0; k < ranges.size(); ++k) {
        if (ranges[k].second == ranges[k].first) continue;
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, srcs[ranges[k].first].data) == hipSuccess && attr.type == hipMemoryTypeDevice &&
            attr.device != c->shard_ctx[k]->device) {
            int peer = 0;
            (void)hipDeviceCanAccessPeer(&peer, c->shard_ctx[k]->device, attr.device);
            if (!peer) { c->set_error("flgpu_transform_batch_device: an image is not resident on the device of its shard (see flgpu_plan_shards)"); return FLGPU_ERR_INVALID_ARG; }
        }
    }
    rc = for_each_shard(c, ranges, [&](uint32_t k, size_t a, size_t b) {
        flgpu_ctx *s = c->shard_ctx[k];
        std::lock_guard<std::mutex> gs(s->mu);
        int r = run_batch_device(s, b - a, srcs + a, same ? ps : ps + a, same, dsts + a, nullptr);
        if (r) return r;
        hipError_t e = hipStreamSynchronize(s->stream);
        return e == hipSuccess ? FLGPU_OK : s->fail(e, "shard sync");
    });
    c->last_shards = ranges;
    c->last_n = n;
    return rc;
} FL_ABI_CATCH

int flgpu_transform_batch(flgpu_ctx *c, size_t n, const flgpu_image *srcs, const flgpu_params *ps, flgpu_image *dsts)
try {
    if (!c) return FLGPU_ERR_INVALID_ARG;
    if (c->shard_ctx.empty()) {
        std::lock_guard<std::mutex> g(c->mu);
        return run_batch_host(c, n, srcs, ps, dsts);
    }
    if (n == 0) return FLGPU_OK;
    if (!srcs || !ps || !dsts) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    std::vector<std::pair<size_t, size_t>> ranges;
    int rc = shard_ranges((uint32_t)c->shard_ctx.size(), n, srcs, ps, false, ranges);
    if (rc) return rc;
    return for_each_shard(c, ranges, [&](uint32_t k, size_t a, size_t b) {
        flgpu_ctx *s = c->shard_ctx[k];
        std::lock_guard<std::mutex> gs(s->mu);
        return run_batch_host(s, b - a, srcs + a, ps + a, dsts + a);
    });
} FL_ABI_CATCH

int flgpu_batch_results(flgpu_ctx *c, size_t n, flgpu_image *dsts)
try {
    if (!c || (!dsts && n)) return FLGPU_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (n != c->last_n) { c->set_error("flgpu_batch_results: n differs from the last device batch"); return FLGPU_ERR_INVALID_ARG; }
    if (c->shard_ctx.empty()) {
        FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
        return collect_results(c, n, dsts, c->last_stream ? c->last_stream : c->stream);
    }
    int rc = FLGPU_OK;
    for (size_t k = 0; k < c->last_shards.size(); ++k) {
        const auto r = c->last_shards[k];
        if (r.second == r.first) continue;
        flgpu_ctx *s = c->shard_ctx[k];
        std::lock_guard<std::mutex> gs(s->mu);
        if (hipSetDevice(s->device) != hipSuccess) return FLGPU_ERR_DEVICE;
        int r2 = collect_results(s, r.second - r.first, dsts + r.first, s->last_stream ? s->last_stream : s->stream);
        if (r2 && !rc) { rc = r2; c->set_error(s->get_error()); }
    }
    return rc;
} FL_ABI_CATCH

int flgpu_transform(flgpu_ctx *c, const flgpu_image *src, const flgpu_params *p, flgpu_image *dst)
try {
    if (!c || !src || !p || !dst || !src->data || !dst->data) return FLGPU_ERR_INVALID_ARG;
    // validate on the caller's thread so that one bad request cannot fail a shared batch
    flgpu_plan plan;
    int rc = flgpu_plan_output(p, src->width, src->height, src->channels, &plan);
    if (rc) return rc;
    const bool jsrc = (src->flags & FLGPU_IMG_JPEG_SOURCE) != 0;
    JpegInfo jinfo;
    if (jsrc) {
        if (jpeg_parse_info(src->data, (size_t)src->capacity, jinfo) != 0) return FLGPU_ERR_INVALID_ARG;
        if (!jinfo.supported) { c->set_error("JPEG stream not covered by the device decoder"); return FLGPU_ERR_UNSUPPORTED; }
        const int prc = jpeg_source_precheck(c, src, jinfo); // before any block is reserved on the file's say-so
        if (prc) return prc;
    } else if (src->capacity < (uint64_t)src->width * src->height * src->channels) return FLGPU_ERR_INVALID_ARG;
    const bool jpeg = p->front_end == FLGPU_FE_JPEG;
    if (!jpeg && dst->capacity < plan.out_bytes) return FLGPU_ERR_BUFFER_TOO_SMALL;
    Request r{};
    r.src = src; r.p = p; r.dst = dst;
    r.src_bytes = (uint64_t)src->width * src->height * src->channels;
    r.weight = r.src_bytes + plan.out_bytes;
    // an encoded stream is staged with the format's worst case on the device, so only the caller's own capacity can be
    // too small, and that is known once the stream's length is (JpegEncoder into a Vec never fails, handler.rs:274-278)
    r.out_bytes = jpeg ? std::min<uint64_t>(dst->capacity, plan.max_out_bytes) : plan.out_bytes;
    if (r.out_bytes == 0) return FLGPU_ERR_BUFFER_TOO_SMALL;
    {
        const uint32_t limit = 4u * lanes_per_device(c) * batch_per_device(c) * c->n_dev();
        std::unique_lock<std::mutex> lk(c->adm_mu);
        c->adm_cv.wait(lk, [&] { return c->admitted < limit; });
        c->admitted++;
    }
    struct Admission {
        flgpu_ctx *c;
        ~Admission() { { std::lock_guard<std::mutex> lk(c->adm_mu); c->admitted--; } c->adm_cv.notify_one(); }
    } admission{c};
    c->staging.fetch_add(1, std::memory_order_acq_rel);
    // buffers from flgpu_host_alloc are page-locked already: the DMA engine reads / writes them directly, no staging copy
    const bool src_pinned = (src->flags & FLGPU_IMG_PINNED) != 0 && !jsrc, dst_pinned = (dst->flags & FLGPU_IMG_PINNED) != 0 && dst->capacity >= r.out_bytes;
    if (src_pinned) r.in = PinBlock{src->data, 0}; else r.in = pin_acquire(c, jsrc ? jpeg_blob_bound(jinfo) : r.src_bytes);
    if (dst_pinned) r.out = PinBlock{dst->data, 0}; else r.out = pin_acquire(c, r.out_bytes);
    auto give_back = [&] { if (!src_pinned) pin_release(c, r.in); if (!dst_pinned) pin_release(c, r.out); };
    if (!r.in.p || !r.out.p) {
        c->staging.fetch_sub(1, std::memory_order_acq_rel);
        give_back();
        return FLGPU_ERR_OOM;
    }
    if (jsrc) {
        // the serial half of the decoder (parsing + Huffman) on the caller's thread, straight into pinned memory:
        // concurrent requests decode in parallel and only the coefficient blob crosses PCIe
        size_t used = 0;
        int jrc;
        {
            // Who decodes the entropy-coded segment?  A thread with an idle CPU under it does it fastest itself (~1.9 ms, and the
            // request skips six kernel launches); once half of the CPUs this process may use are decoding, further requests are
            // only STAGED (header + the unstuffed segment, ~0.05 ms) and the device decodes them (fl_jpeghuff_dev.hip) -- host and
            // device then work side by side, and a burst of callers no longer queues for CPUs.
            // No more host decoders at once than the process has CPUs: sixty-four runnable decoders on sixteen CPUs all finish late
            // (p99 of the request 60 ms against 20 ms with 32 callers, profiles/r03_latency_jpeg_sources.txt).
            const int policy = device_huffman_policy(c, src->capacity);
            bool on_host = policy == 0;
            {
                std::unique_lock<std::mutex> lk(c->dec_mu);
                if (!c->dec_limit) c->dec_limit = c->cfg.decode_threads ? c->cfg.decode_threads : usable_cpus();
                // (round 5: a quarter of the CPUs, not half -- the device's decode kernels of a batch went from 1.9 to 1.1 ms, and with 64 callers on
                // 16 CPUs every host decode beyond that takes 1.8 ms of CPU from threads that stage files: tools/experiments/jh_policy.sh)
                if (policy == 1 && c->decoding < std::max(1u, c->dec_limit / 4u)) on_host = true;
                if (on_host) {
                    // (bounded: a file that keeps its decoder busy for long -- a huge progressive picture -- must not park every other
                    // JPEG request behind it; after the deadline the caller decodes anyway, one runnable thread more than CPUs)
                    (void)c->dec_cv.wait_for(lk, std::chrono::milliseconds(250), [&] { return c->decoding < c->dec_limit; });
                    c->decoding++;
                }
            }
            struct Turn { flgpu_ctx *c; bool held; ~Turn() { if (held) { { std::lock_guard<std::mutex> lk(c->dec_mu); c->decoding--; } c->dec_cv.notify_one(); } } } turn{c, on_host};
            jrc = jpeg_source_to_blob(c, src, static_cast<uint8_t *>(r.in.p), r.in.cap, &r.jhdr, &used, on_host);
            if (!jrc) stage_of(static_cast<const uint8_t *>(r.in.p), r.jhdr, r.jstage);
        }
        if (jrc) { c->staging.fetch_sub(1, std::memory_order_acq_rel); give_back(); return jrc; }
        r.jpeg = true;
        if (r.jhdr.nc == 4 && c->cfg.use_embedded_profile) r.icc.swap(jinfo.icc);
        r.file_bytes = src->capacity;
        r.src_bytes = used;
    } else
    if (!src_pinned) memcpy(r.in.p, src->data, r.src_bytes); // on the caller's thread: concurrent callers stage in parallel
    {
        std::unique_lock<std::mutex> lk(c->qmu);
        c->staging.fetch_sub(1, std::memory_order_acq_rel);
        if (c->stop) { lk.unlock(); give_back(); return FLGPU_ERR_SHUTDOWN; }
        const int wrc = start_workers(c);
        if (wrc) { lk.unlock(); give_back(); return wrc; }
        c->queue.push_back(&r);
    }
    c->qcv.notify_all();
    {
        std::unique_lock<std::mutex> lk(c->qmu);
        r.cv.wait(lk, [&] { return r.done; });
    }
    if (r.status == FLGPU_OK && !dst_pinned) memcpy(dst->data, r.out.p, std::min<uint64_t>(dst->bytes, r.out_bytes));
    give_back();
    if (r.status == FL_STATUS_RETRY_HOST_HUFFMAN) {
        // the device entropy decoder gave up on this file (its subsequence states did not settle within the rounds it runs, or the
        // stream holds an invalid code word): once more, Huffman-decoded on this thread -- which either works or names the defect
        if (tl_force_host_huffman) return FLGPU_ERR_DEVICE;
        struct Force { Force() { tl_force_host_huffman = true; } ~Force() { tl_force_host_huffman = false; } } force;
        return flgpu_transform(c, src, p, dst);
    }
    if (dst_pinned) dst->flags |= FLGPU_IMG_PINNED;
    return r.status;
} FL_ABI_CATCH

int flgpu_ycck_to_cmyk(flgpu_ctx *c, uint8_t *raw, uint64_t n_pixels)
try {
    if (!c || (!raw && n_pixels)) return FLGPU_ERR_INVALID_ARG;
    if (n_pixels == 0) return FLGPU_OK;
    if (n_pixels >= (1ull << 30)) return FLGPU_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> g(c->mu);
    FL_HIP(c, hipSetDevice(c->device), "hipSetDevice");
    const size_t bytes = (size_t)n_pixels * 4;
    FL_HIP(c, c->d_in.reserve(bytes), "device staging");
    FL_HIP(c, hipMemcpyAsync(c->d_in.p, raw, bytes, hipMemcpyHostToDevice, c->stream), "H2D");
    FL_HIP(c, launch_ycck_to_cmyk(static_cast<uint32_t *>(c->d_in.p), n_pixels, c->stream), "ycck kernel");
    FL_HIP(c, hipMemcpyAsync(raw, c->d_in.p, bytes, hipMemcpyDeviceToHost, c->stream), "D2H");
    FL_HIP(c, hipStreamSynchronize(c->stream), "sync");
    return FLGPU_OK;
} FL_ABI_CATCH

} // extern "C"
