// lws_clone + lws_pool: cross-forward overlap as a library feature (include/lwsnet_hip.h).
//
// A batch-1 forward (/root/reference/models/models.py:106-164) is a chain of ~35 dependent launches whose fixed costs
// are ~40 % of its 0.5 ms, and it costs ~345 us of host time to issue.  Nothing inside ONE forward
// removes that; independent forwards do: with W worker threads, each owning a clone of the model (shared read-only
// parameters, private workspace) and ONE HIP stream, the launch-bound chains of W consecutive forwards overlap on the
// device and their host cost runs W-way parallel.  The reference has no counterpart (single stream, single thread:
// inference.py:105-109); results are the bits lws_forward returns.
#include <condition_variable>
#include <deque>
#include <mutex>
#include <new>
#include <thread>

#include "lws_common.h"

using namespace lws;

namespace {

enum JobState { JOB_FREE = 0, JOB_QUEUED, JOB_RUNNING, JOB_ISSUED, JOB_RECYCLING };

struct Job {
    const float *left = nullptr, *right = nullptr;
    float *out[4] = {nullptr, nullptr, nullptr, nullptr};
    int B = 0, H = 0, W = 0;
    hipEvent_t ready = nullptr;   // recorded on the submitter's stream: the inputs are complete behind it
    hipEvent_t done = nullptr;    // recorded on the worker's stream behind the forward
    int state = JOB_FREE;
    int rc = LWS_OK;
    int64_t ticket = -1;
    std::string err;
};

}  // namespace

struct lws_pool {
    lws_ctx *src = nullptr;
    int device = 0;
    std::vector<lws_ctx *> workers;
    std::vector<hipStream_t> streams;
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<int> queue;        // slots waiting for a worker, in ticket order
    std::vector<Job> slots;       // ring: ticket t lives in slot t % slots.size()
    int64_t next_ticket = 0;
    bool stop = false;
    // sticky first failure of any job: a slot is recycled 4 x workers submits later and its rc / message with it,
    // so a failed forward outside that window would otherwise read as success.  Reported by wait on a recycled ticket,
    // by wait_all and by the next submit; cleared by lws_pool_clear_error.
    int sticky_rc = LWS_OK;
    int64_t sticky_ticket = -1;
    std::string sticky_err;
};

static void worker_main(lws_pool *p, int wi)
{
    (void)hipSetDevice(p->device);
    lws_ctx *h = p->workers[wi];
    hipStream_t st = p->streams[wi];
    for (;;) {
        int slot;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_work.wait(lk, [&] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;          // stop requested and nothing left to run
            slot = p->queue.front();
            p->queue.pop_front();
            p->slots[slot].state = JOB_RUNNING;
        }
        Job &j = p->slots[slot];
        int rc = LWS_OK;
        std::string err;
        if (hipStreamWaitEvent(st, j.ready, 0) != hipSuccess) {
            rc = LWS_ERR_HIP;
            err = "lws_pool: hipStreamWaitEvent on the submitter's event failed";
        }
        if (rc == LWS_OK) {
            rc = lws_forward(h, j.left, j.right, j.B, j.H, j.W, j.out, (void *)st);
            if (rc != LWS_OK) err = lws_last_error();          // thread-local: this worker's message
        }
        if (hipEventRecord(j.done, st) != hipSuccess && rc == LWS_OK) {
            rc = LWS_ERR_HIP;
            err = "lws_pool: hipEventRecord behind the forward failed";
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            j.rc = rc;
            j.err = err;
            j.state = JOB_ISSUED;
            if (rc != LWS_OK) {
                // workers finish out of ticket order: keep the first status and message, but the SMALLEST failed ticket, so
                // that "tickets older than sticky_ticket ran to completion" holds with several workers
                if (p->sticky_rc == LWS_OK) {
                    p->sticky_rc = rc;
                    p->sticky_ticket = j.ticket;
                    p->sticky_err = err;
                } else if (j.ticket < p->sticky_ticket) {
                    p->sticky_ticket = j.ticket;
                }
            }
        }
        p->cv_done.notify_all();
    }
}

static int pool_check_device(const lws_pool *p, const char *what)
{
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) {
        (void)hipGetLastError();
        cur = -1;
    }
    if (cur != p->device) {
        set_error("%s: the pool belongs to HIP device %d but the calling thread's current device is %d", what, p->device, cur);
        return LWS_ERR_INVALID;
    }
    return LWS_OK;
}

extern "C" {

int lws_clone(lws_handle src, lws_handle *out)
{
    LWS_CHECK_ARG(src && out, "lws_clone: null argument");
    if (!src->finalized) {
        set_error("lws_clone: the source handle has not been finalized");
        return LWS_ERR_STATE;
    }
    lws_ctx *h = new (std::nothrow) lws_ctx();
    if (!h) {
        set_error("out of host memory");
        return LWS_ERR_NOMEM;
    }
    h->cfg = src->cfg;
    h->opt = src->opt;
    h->device = src->device;
    h->cu_count = src->cu_count;         // (apply_options copies it into stage[]: a clone must not fall back to the 256 default)
    h->spec = src->spec;
    h->params = src->params;             // shared, read-only; owned by src
    h->params_bytes = src->params_bytes;
    h->owns_params = false;
    for (int i = 0; i < 3; ++i) {
        h->stage[i] = src->stage[i];
        h->stage[i].clk = nullptr;       // (a clock stamp armed on the source stays the source's: lws_clock_stamp)
    }
    h->net2d = src->net2d;
    h->have_2d = src->have_2d;
    h->finalized = true;
    *out = h;
    return LWS_OK;
}

int lws_pool_create(lws_handle src, int workers, int flags, lws_pool_handle *out)
{
    LWS_CHECK_ARG(src && out, "lws_pool_create: null argument");
    LWS_CHECK_ARG(workers >= 1 && workers <= 16, "lws_pool_create: workers must be in 1..16 (got %d)", workers);
    LWS_CHECK_ARG((flags & ~LWS_POOL_SIDE_STREAMS) == 0, "lws_pool_create: unknown flag bits 0x%x", flags);
    if (!src->finalized || !src->have_2d) {
        set_error("lws_pool_create: the model handle must be finalized with the full state dict");
        return LWS_ERR_STATE;
    }
    lws_pool *p = new (std::nothrow) lws_pool();
    if (!p) {
        set_error("out of host memory");
        return LWS_ERR_NOMEM;
    }
    p->src = src;
    p->device = src->device;
    int rc = pool_check_device(p, "lws_pool_create");
    if (rc) {
        delete p;
        return rc;
    }
    p->slots.resize(4 * (size_t)workers);
    auto fail = [&](int code) {
        lws_pool_destroy(p);
        return code;
    };
    for (Job &j : p->slots) {
        // `ready` orders device work only; `done` is what lws_pool_wait blocks the host on: it publishes to the host
        if (hipEventCreateWithFlags(&j.ready, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&j.done, hipEventDisableTiming) != hipSuccess) {
            set_error("lws_pool_create: hipEventCreate failed");
            return fail(LWS_ERR_HIP);
        }
    }
    for (int i = 0; i < workers; ++i) {
        lws_ctx *h = nullptr;
        rc = lws_clone(src, &h);
        if (rc) return fail(rc);
        p->workers.push_back(h);
        // one stream = one hardware queue per worker unless the caller asks for the per-handle side streams too
        h->opt.side_streams = (flags & LWS_POOL_SIDE_STREAMS) ? 1 : 0;
        // several forwards share the CUs: no residency cap on k_conv3d_mid8q (a capped workgroup holds a third of a CU's LDS
        // idle, which the kernels of the other workers' forwards could use)
        h->mid8_balance = 0;
        for (int s_ = 0; s_ < 3; ++s_) h->stage[s_].mid8_balance = 0;
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            set_error("lws_pool_create: hipStreamCreate failed");
            return fail(LWS_ERR_HIP);
        }
        p->streams.push_back(st);
    }
    for (int i = 0; i < workers; ++i) p->threads.emplace_back(worker_main, p, i);
    *out = p;
    return LWS_OK;
}

int lws_pool_workers(lws_pool_handle p) { return p ? (int)p->workers.size() : 0; }

int lws_pool_reserve(lws_pool_handle p, int B, int H, int W)
{
    LWS_CHECK_ARG(p, "lws_pool_reserve: null pool");
    int rc = pool_check_device(p, "lws_pool_reserve");
    if (rc) return rc;
    rc = lws_pool_wait_all(p);         // workspaces may be re-allocated: nothing may be in flight
    if (rc) return rc;
    for (lws_ctx *h : p->workers) {
        rc = lws_reserve(h, B, H, W);
        if (rc) return rc;
    }
    return LWS_OK;
}

int lws_pool_submit(lws_pool_handle p, const float *left, const float *right, int B, int H, int W,
                    float *const pred_out[4], void *after_stream, int64_t *ticket)
{
    LWS_CHECK_ARG(p && left && right && pred_out && ticket, "lws_pool_submit: null argument");
    for (int s = 0; s < 4; ++s) LWS_CHECK_ARG(pred_out[s], "lws_pool_submit: null output for stage %d", s + 1);
    LWS_CHECK_ARG(B >= 1 && H > 0 && W > 0, "lws_pool_submit: bad shape B=%d %dx%d", B, H, W);
    int rc = pool_check_device(p, "lws_pool_submit");
    if (rc) return rc;
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->stop) {
        set_error("lws_pool_submit: the pool is shutting down");
        return LWS_ERR_STATE;
    }
    if (p->sticky_rc != LWS_OK) {
        set_error("lws_pool_submit: an earlier job (ticket %lld) failed: %s", (long long)p->sticky_ticket, p->sticky_err.c_str());
        return p->sticky_rc;
    }
    const int64_t t = p->next_ticket++;          // the ticket (and with it the slot) is this call's from here on
    const int slot = (int)(t % (int64_t)p->slots.size());
    Job &j = p->slots[slot];
    // the slot's previous job (ticket t - capacity) must have left the device before its events are reused
    p->cv_done.wait(lk, [&] { return j.state == JOB_FREE || j.state == JOB_ISSUED; });
    if (j.state == JOB_ISSUED) {
        j.state = JOB_RECYCLING;
        lk.unlock();
        const hipError_t e = hipEventSynchronize(j.done);
        lk.lock();
        if (e != hipSuccess) {
            j.state = JOB_FREE;
            j.ticket = -1;
            p->cv_done.notify_all();
            set_error("lws_pool_submit: hipEventSynchronize on a recycled job failed");
            return LWS_ERR_HIP;
        }
    }
    if (hipEventRecord(j.ready, (hipStream_t)after_stream) != hipSuccess) {
        j.state = JOB_FREE;
        j.ticket = -1;
        p->cv_done.notify_all();
        set_error("lws_pool_submit: hipEventRecord on the caller's stream failed");
        return LWS_ERR_HIP;
    }
    j.left = left;
    j.right = right;
    for (int s = 0; s < 4; ++s) j.out[s] = pred_out[s];
    j.B = B;
    j.H = H;
    j.W = W;
    j.rc = LWS_OK;
    j.err.clear();
    j.ticket = t;
    j.state = JOB_QUEUED;
    p->queue.push_back(slot);
    *ticket = t;
    lk.unlock();
    p->cv_work.notify_one();
    p->cv_done.notify_all();                     // waiters on the recycled ticket of this slot
    return LWS_OK;
}

int lws_pool_wait(lws_pool_handle p, int64_t ticket)
{
    LWS_CHECK_ARG(p, "lws_pool_wait: null pool");
    std::unique_lock<std::mutex> lk(p->mu);
    LWS_CHECK_ARG(ticket >= 0 && ticket < p->next_ticket, "lws_pool_wait: ticket %lld was never issued", (long long)ticket);
    const int slot = (int)(ticket % (int64_t)p->slots.size());
    Job &j = p->slots[slot];
    // the slot has been recycled: that required the job to be complete; its own status is gone, the pool's first failure is not
    auto recycled = [&]() -> int {
        // (tickets older than the smallest failed ticket ran to completion; anything from it on may have failed unrecorded)
        if (p->sticky_rc != LWS_OK && ticket >= p->sticky_ticket) {
            set_error("lws_pool_wait: ticket %lld has been recycled; the pool's first failed job was ticket %lld: %s",
                      (long long)ticket, (long long)p->sticky_ticket, p->sticky_err.c_str());
            return p->sticky_rc;
        }
        return LWS_OK;
    };
    if (j.ticket != ticket) return recycled();
    p->cv_done.wait(lk, [&] { return j.ticket != ticket || j.state == JOB_ISSUED || j.state == JOB_FREE; });
    if (j.ticket != ticket || j.state == JOB_FREE) return recycled();
    const int rc = j.rc;
    const std::string err = j.err;
    hipEvent_t done = j.done;
    lk.unlock();
    if (hipEventSynchronize(done) != hipSuccess) {
        set_error("lws_pool_wait: hipEventSynchronize failed");
        return LWS_ERR_HIP;
    }
    if (rc != LWS_OK) set_error("%s", err.c_str());
    return rc;
}

int lws_pool_wait_all(lws_pool_handle p)
{
    LWS_CHECK_ARG(p, "lws_pool_wait_all: null pool");
    int64_t hi;
    size_t cap;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        hi = p->next_ticket;
        cap = p->slots.size();
    }
    int first_rc = LWS_OK;
    std::string first_err;
    for (int64_t t = hi > (int64_t)cap ? hi - (int64_t)cap : 0; t < hi; ++t) {
        const int rc = lws_pool_wait(p, t);
        if (rc != LWS_OK && first_rc == LWS_OK) {
            first_rc = rc;
            first_err = lws_last_error();
        }
    }
    if (first_rc == LWS_OK) {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->sticky_rc != LWS_OK) {            // a failure older than the 4 x workers tickets scanned above
            first_rc = p->sticky_rc;
            first_err = "ticket " + std::to_string(p->sticky_ticket) + ": " + p->sticky_err;
        }
    }
    if (first_rc != LWS_OK) set_error("%s", first_err.c_str());
    return first_rc;
}

// Per-class kernel timing of the forwards the workers run (the same hipEvent pairs lws_profile_* records for one handle):
// enable on every worker, read the sum.  Call both with nothing in flight (lws_pool_wait_all first): the workers' records
// are not locked against a running forward.
int lws_pool_profile_enable(lws_pool_handle p, int class_mask, int every_n)
{
    LWS_CHECK_ARG(p && every_n >= 1, "lws_pool_profile_enable: bad argument");
    for (lws_ctx *h : p->workers) {
        int rc = lws_profile_enable(h, class_mask);
        if (rc) return rc;
        rc = lws_profile_sample(h, every_n);
        if (rc) return rc;
    }
    return LWS_OK;
}

int lws_pool_profile_read(lws_pool_handle p, double *total_ms, int64_t *launches)
{
    LWS_CHECK_ARG(p && total_ms && launches, "lws_pool_profile_read: null argument");
    int rc = pool_check_device(p, "lws_pool_profile_read");
    if (rc) return rc;
    for (int i = 0; i < LWS_KC_COUNT; ++i) {
        total_ms[i] = 0.0;
        launches[i] = 0;
    }
    double t[LWS_KC_COUNT];
    int64_t n[LWS_KC_COUNT];
    for (lws_ctx *h : p->workers) {
        rc = lws_profile_read(h, t, n);
        if (rc) return rc;
        for (int i = 0; i < LWS_KC_COUNT; ++i) {
            total_ms[i] += t[i];
            launches[i] += n[i];
        }
    }
    return LWS_OK;
}

int lws_pool_clear_error(lws_pool_handle p)
{
    LWS_CHECK_ARG(p, "lws_pool_clear_error: null pool");
    std::lock_guard<std::mutex> lk(p->mu);
    p->sticky_rc = LWS_OK;
    p->sticky_ticket = -1;
    p->sticky_err.clear();
    return LWS_OK;
}

int lws_pool_destroy(lws_pool_handle p)
{
    if (!p) return LWS_OK;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv_work.notify_all();
    for (std::thread &t : p->threads)
        if (t.joinable()) t.join();              // workers drain the queue before they leave
    for (hipStream_t st : p->streams) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamDestroy(st);
    }
    for (lws_ctx *h : p->workers) (void)lws_destroy(h);
    for (Job &j : p->slots) {
        if (j.ready) (void)hipEventDestroy(j.ready);
        if (j.done) (void)hipEventDestroy(j.done);
    }
    delete p;
    return LWS_OK;
}

}  // extern "C"
