// batch.cpp — mpeg::VideoBatch: many streams, one reconstruction call per tick.
//
// Each stream keeps its own, unmodified parser (mpeg::Video); what it would submit to its own device
// store goes through a Port into the batch's descriptor arrays instead (stream index, macroblock and
// coefficient offsets rebased).  Pictures of ONE stream depend on each other, so a stream never has two
// pictures in the same device call: a second one (first reference picture of a stream, which yields
// no frame; the re-submit after a duplicated macroblock address in a damaged stream) flushes the batch
// first — launches are stream ordered, so "last writer in bitstream order" is kept.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <climits>
#include <cmath>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

#include <pthread.h>
#include <sched.h>
#include <stdio.h>

#include <errno.h>

#include "mpeg.hpp"

namespace mpeg {

// A stream's parser talks to the batch through its Port.  Normally a request acts at once; while the
// stream is being parsed on a pool thread (`recording`), requests are only recorded — nothing shared is
// touched — and VideoBatch::DecodeAll replays them on its own thread afterwards.
class VideoBatch::Port : public VideoBackend {
public:
    struct Event {
        enum Kind { Open, Quant, Submit } kind;
        int width = 0, height = 0;
        uint8_t quant[128];
        mpeghip_pic_desc pic;
        std::vector<mpeghip_mb_desc> mbs;
        CoefBytes coefs;
    };
    Port(VideoBatch *b, uint32_t stream) : b_(b), stream_(stream) {}
    // events[0 .. n_events) are this round's; the objects (and the capacity of their vectors: a 1080p picture is
    // 2.5 MB, and fresh allocations of that size are mmap + page faults, which serialise the pool) are reused
    Event &nextEvent()
    {
        if (n_events == events.size())
            events.emplace_back();
        return events[n_events++];
    }
    void open(int width, int height) override
    {
        if (recording) {
            Event &e = nextEvent();
            e.kind = Event::Open;
            e.width = width;
            e.height = height;
            return;
        }
        b_->openStore(width, height);
    }
    void setQuant(const uint8_t intra[64], const uint8_t non_intra[64]) override
    {
        if (recording) {
            Event &e = nextEvent();
            e.kind = Event::Quant;
            memcpy(e.quant, intra, 64);
            memcpy(e.quant + 64, non_intra, 64);
            return;
        }
        b_->Flush(); // the table belongs to pictures not yet queued
        b_->store_->setQuant(stream_, intra, non_intra);
    }
    void submit(const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
                size_t coef_bytes) override
    {
        if (recording) {
            Event &e = nextEvent();
            e.kind = Event::Submit;
            e.pic = pic;
            e.mbs.assign(mbs, mbs + n_mbs);
            e.coefs.assign(coefs, coefs + coef_bytes);
            return;
        }
        b_->queue(stream_, pic, mbs, n_mbs, coefs, coef_bytes);
    }
    void submitOwned(const mpeghip_pic_desc &pic, std::vector<mpeghip_mb_desc> &mbs, CoefBytes &coefs) override
    {
        if (recording) { // keep the parser's arrays, give it the event's old ones: no copy of a 2.5 MB picture
            Event &e = nextEvent();
            e.kind = Event::Submit;
            e.pic = pic;
            e.pic.mb_count = (uint32_t)mbs.size();
            e.mbs.swap(mbs);
            e.coefs.swap(coefs);
            return;
        }
        b_->queue(stream_, pic, mbs.data(), (uint32_t)mbs.size(), coefs.data(), coefs.size());
    }
    void readPlanes(uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        if (recording)
            throw std::logic_error("VideoBatch: frame read during a parallel parse");
        b_->Flush();
        b_->store_->readPlanes(stream_, slot, y, cb, cr);
    }
    void readRGBA(uint32_t slot, uint8_t *dst) override
    {
        if (recording)
            throw std::logic_error("VideoBatch: frame read during a parallel parse");
        b_->Flush();
        b_->store_->readRGBA(stream_, slot, dst);
    }
    void replay(const Event &e) // on the batch's thread
    {
        switch (e.kind) {
        case Event::Open: b_->openStore(e.width, e.height); break;
        case Event::Quant:
            b_->Flush();
            b_->store_->setQuant(stream_, e.quant, e.quant + 64);
            break;
        case Event::Submit:
            b_->queue(stream_, e.pic, e.mbs.data(), (uint32_t)e.mbs.size(), e.coefs.data(), e.coefs.size());
            break;
        }
    }
    bool recording = false;
    std::vector<Event> events;
    size_t n_events = 0;

private:
    VideoBatch *b_;
    uint32_t stream_;
};

// n - 1 parked threads + the caller share the items of one run() at a time.
bool pinThisThreadToNode(int node)
{
    if (node < 0)
        return false;
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f)
        return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    int lo, hi, n = 0;
    while (fscanf(f, "%d", &lo) == 1) { // "0-63,128-191"
        hi = lo;
        int c = fgetc(f);
        if (c == '-') {
            if (fscanf(f, "%d", &hi) != 1)
                break;
            c = fgetc(f);
        }
        for (int k = lo; k <= hi && k < CPU_SETSIZE; k++, n++)
            CPU_SET(k, &set);
        if (c != ',')
            break;
    }
    fclose(f);
    return n > 0 && pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
}

// The CPU time this process really gets, in cores: the affinity mask, capped by the cgroup's CPU-time quota (v2: cpu.max; v1:
// cpu.cfs_quota_us / cpu.cfs_period_us).  A container that shows 256 hardware threads under a 10-core quota runs a 64-thread
// pool 26 % SLOWER than a 16-thread one (BENCH_r04 host_parsed: the threads are throttled in turn and every round waits for the
// last of them), so pools are sized by this, not by the number asked for.
// One cgroup directory's quota in cores (0: none there).  v2: "<quota> <period>" or "max <period>" in cpu.max; v1: cpu.cfs_quota_us
// (-1: none) / cpu.cfs_period_us.
static double quotaOfCgroupDir(const std::string &dir, bool v2)
{
    if (v2) {
        double out = 0;
        if (FILE *f = fopen((dir + "/cpu.max").c_str(), "r")) {
            char q[32] = {0};
            double period = 0;
            if (fscanf(f, "%31s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
                out = atof(q) / period;
            fclose(f);
        }
        return out;
    }
    double q = 0, period = 0;
    if (FILE *fq = fopen((dir + "/cpu.cfs_quota_us").c_str(), "r")) {
        if (fscanf(fq, "%lf", &q) != 1)
            q = 0;
        fclose(fq);
    }
    if (FILE *fp = fopen((dir + "/cpu.cfs_period_us").c_str(), "r")) {
        if (fscanf(fp, "%lf", &period) != 1)
            period = 0;
        fclose(fp);
    }
    return q > 0 && period > 0 ? q / period : 0;
}

// The tightest CPU-time quota that applies to this process: its own cgroup's AND every ancestor's (a quota set on a parent slice
// binds the children; a process without a cgroup namespace — a systemd slice, a pod without cgroupns — sits in a NESTED directory
// that the mount's root files say nothing about).  /proc/self/cgroup names the directory: "0::/a/b" (v2) or "N:cpu,cpuacct:/a/b"
// (v1); the walk goes from there up to the mount's root.  `root` / `proc_file`: the tests' stand-ins for /sys/fs/cgroup and
// /proc/self/cgroup.
double CgroupQuotaCores(const char *root, const char *proc_file)
{
    const std::string base = root ? root : "/sys/fs/cgroup";
    std::string v2_path, v1_path;
    bool have_v2 = false, have_v1 = false;
    if (FILE *f = fopen(proc_file ? proc_file : "/proc/self/cgroup", "r")) {
        char line[4096];
        while (fgets(line, sizeof(line), f)) {
            std::string l(line);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r'))
                l.pop_back();
            const size_t a = l.find(':'), b = a == std::string::npos ? a : l.find(':', a + 1);
            if (b == std::string::npos)
                continue;
            const std::string ctrl = l.substr(a + 1, b - a - 1), path = l.substr(b + 1);
            if (ctrl.empty() && l.compare(0, a, "0") == 0) {
                v2_path = path;
                have_v2 = true;
            } else if (("," + ctrl + ",").find(",cpu,") != std::string::npos) {
                v1_path = path;
                have_v1 = true;
            }
        }
        fclose(f);
    }
    double tightest = 0;
    auto walk = [&](const std::string &mount, std::string path, bool v2) {
        for (;;) { // the process's directory, then its parents, then the mount's root
            const double q = quotaOfCgroupDir(mount + (path == "/" ? "" : path), v2);
            if (q > 0 && (tightest == 0 || q < tightest))
                tightest = q;
            if (path.empty() || path == "/")
                break;
            const size_t cut = path.find_last_of('/');
            path = cut == 0 || cut == std::string::npos ? "/" : path.substr(0, cut);
        }
    };
    // (a directory named by /proc/self/cgroup that the mount does not show — a cgroup namespace — simply has no files: the walk
    // still reaches the mount's root, which is what round 5 looked at)
    walk(base, have_v2 ? v2_path : "/", true);
    walk(base + "/cpu", have_v1 ? v1_path : "/", false);
    return tightest;
}

// The CPU time this process really gets, in cores: the affinity mask, capped by the cgroup CPU-time quota that applies to it
// (CgroupQuotaCores).  A container that shows 256 hardware threads under a 10-core quota runs a 64-thread pool 26 % SLOWER than a
// 16-thread one (BENCH_r04 host_parsed: the threads are throttled in turn and every round waits for the last of them), so pools
// are sized by this, not by the number asked for.
double EffectiveCores()
{
    double cores = (double)std::thread::hardware_concurrency();
    // (a fixed cpu_set_t holds 1 024 CPUs: on a larger host the call fails with EINVAL — ask with a mask sized for the machine)
    for (size_t n = 1024; n <= (1u << 20); n *= 4) {
        cpu_set_t *set = CPU_ALLOC(n);
        if (!set)
            break;
        const size_t bytes = CPU_ALLOC_SIZE(n);
        const int rc = sched_getaffinity(0, bytes, set);
        if (rc == 0)
            cores = (double)CPU_COUNT_S(bytes, set);
        CPU_FREE(set);
        if (rc == 0 || errno != EINVAL)
            break;
    }
    if (cores < 1)
        cores = 1;
    const double quota = CgroupQuotaCores(nullptr, nullptr);
    return quota > 0 && quota < cores ? quota : cores;
}
// threads a pool gets when `asked` are asked for (0 = as many as there is CPU time for): never more than the quota rounded up
static unsigned poolThreads(unsigned asked)
{
    const unsigned fit = (unsigned)std::ceil(EffectiveCores());
    const unsigned cap = fit < 1 ? 1 : fit;
    return asked == 0 ? cap : (asked < cap ? asked : cap);
}

// Workers sleep on a generation counter (a futex word), not on a condition variable: a condition variable hands its mutex
// from one woken thread to the next, so waking 63 workers costs 63 lock hand-overs IN SERIES — milliseconds per run() on a
// 64-thread pool, twice per picture round (parse, puts).  Here a run() bumps the counter and wakes everybody at once; workers
// that finished the last run spin on the counter for about ten microseconds first (rounds follow each other closely), and the
// caller learns of the end from a count of busy workers, the last of which wakes it.
class HostPool {
public:
    HostPool(unsigned n, int numa_node)
    {
        for (unsigned i = 1; i < n; i++)
            workers_.emplace_back([this, numa_node] {
                if (numa_node >= 0) { // (-1: stays where the scheduler puts it)
                    pins_asked_++;
                    if (!pinThisThreadToNode(numa_node))
                        pins_failed_++;
                }
                started_.fetch_add(1, std::memory_order_release);
                work();
            });
        // every worker has tried its binding when the constructor returns: NumaPins() read right after SetThreads /
        // SetNumaNode is the final count, and no run() starts with threads still on the wrong socket
        while (started_.load(std::memory_order_acquire) < workers_.size())
            std::this_thread::yield();
    }
    std::atomic<size_t> started_{0};
    std::atomic<unsigned> pins_asked_{0}, pins_failed_{0};
    ~HostPool()
    {
        stop_.store(true, std::memory_order_release);
        generation_.fetch_add(1, std::memory_order_release);
        futexWake(&generation_, INT_MAX);
        for (std::thread &t : workers_)
            t.join();
    }
    // fn(k) for k in [0, n), each exactly once; returns when all are done; the first exception is rethrown
    void run(size_t n, const std::function<void(size_t)> &fn)
    {
        fn_ = &fn;
        n_ = n;
        next_.store(0, std::memory_order_relaxed);
        error_ = nullptr;
        busy_.store((uint32_t)workers_.size(), std::memory_order_relaxed);
        generation_.fetch_add(1, std::memory_order_release); // (publishes the four stores above)
        futexWake(&generation_, INT_MAX);
        drain();
        for (int spin = 0;;) { // every worker has passed through this generation when busy_ reaches 0
            const uint32_t b = busy_.load(std::memory_order_acquire);
            if (b == 0)
                break;
            if (spin < kSpins) {
                spin++;
                cpuRelax();
            } else {
                futexWait(&busy_, b);
            }
        }
        fn_ = nullptr;
        if (error_)
            std::rethrow_exception(error_);
    }

private:
    static constexpr int kSpins = 300; // x one pause instruction: about ten microseconds (longer spins cost a throttled container its CPU quota)
    static void cpuRelax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    static void futexWait(std::atomic<uint32_t> *word, uint32_t seen)
    { // sleeps only while *word == seen; spurious returns are fine (every caller re-checks)
        syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
    }
    static void futexWake(std::atomic<uint32_t> *word, int n) { syscall(SYS_futex, reinterpret_cast<uint32_t *>(word), FUTEX_WAKE_PRIVATE, n, nullptr, nullptr, 0); }
    void drain()
    {
        for (;;) {
            const size_t k = next_.fetch_add(1, std::memory_order_relaxed);
            if (k >= n_)
                return;
            try {
                (*fn_)(k);
            } catch (...) {
                std::lock_guard<std::mutex> l(m_);
                if (!error_)
                    error_ = std::current_exception();
            }
        }
    }
    void work()
    {
        uint32_t seen = 0;
        for (;;) {
            for (int spin = 0;;) {
                const uint32_t g = generation_.load(std::memory_order_acquire);
                if (g != seen) {
                    seen = g;
                    break;
                }
                if (spin < kSpins) {
                    spin++;
                    cpuRelax();
                } else {
                    futexWait(&generation_, seen);
                }
            }
            if (stop_.load(std::memory_order_acquire))
                return;
            drain();
            if (busy_.fetch_sub(1, std::memory_order_acq_rel) == 1)
                futexWake(&busy_, 1);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_; // error_ only
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    std::atomic<uint32_t> generation_{0}, busy_{0}; // futex words
    std::atomic<bool> stop_{false};
    std::exception_ptr error_;
};

VideoBatch::VideoBatch(Device *dev, uint32_t n_streams) : VideoBatch(dev->newBatchStore(), n_streams) {}

VideoBatch::VideoBatch(std::unique_ptr<BatchStore> store, uint32_t n_streams) : store_(std::move(store)), capacity_(n_streams)
{
    if (n_streams == 0)
        throw std::runtime_error("VideoBatch: n_streams is 0");
    pending_.assign(n_streams, 0);
}

VideoBatch::~VideoBatch() = default;

namespace {
double nowSeconds() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
} // namespace

void VideoBatch::SetThreads(unsigned n)
{
    n = poolThreads(n); // (0: as many as the process has CPU time for; never more than that)
    if (n == threads_)
        return;
    pool_.reset(n > 1 ? new HostPool(n, numa_node_) : nullptr);
    threads_ = n;
}

void VideoBatch::SetNumaNode(int node)
{
    if (node == numa_node_)
        return;
    numa_node_ = node;
    if (threads_ > 1) { // restart the pool where it belongs
        pool_.reset();
        pool_.reset(new HostPool(threads_, numa_node_));
    }
}

void VideoBatch::NumaPins(unsigned out[2]) const
{
    out[0] = pool_ ? pool_->pins_asked_.load() : 0;
    out[1] = pool_ ? pool_->pins_failed_.load() : 0;
}

void VideoBatch::openStore(int width, int height)
{
    if (width_ == 0) {
        width_ = width;
        height_ = height;
        store_->open(width, height, capacity_);
    } else if (width_ != width || height_ != height) {
        throw std::runtime_error("VideoBatch: all streams must have the same picture size");
    }
}

Video *VideoBatch::AddStream(Buffer *buf)
{
    if (videos_.size() >= capacity_)
        throw std::runtime_error("VideoBatch: more streams than the batch was opened for");
    const uint32_t idx = (uint32_t)videos_.size();
    Port *port = new Port(this, idx);
    // the Video owns the port; ports_ only learns about it once the Video exists (its constructor may throw:
    // geometry that does not match the store, a backend error — the port dies with the unique_ptr then)
    std::unique_ptr<Video> v(new Video(buf, std::unique_ptr<VideoBackend>(port)));
    ports_.push_back(port);
    videos_.push_back(std::move(v));
    return videos_.back().get();
}

void VideoBatch::queue(uint32_t stream, const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                       const uint8_t *coefs, size_t coef_bytes)
{
    if (pending_[stream])
        Flush(); // two pictures of one stream never share a device call
    const uint32_t pic_index = (uint32_t)pics_.size(), mb_first = (uint32_t)mbs_.size();
    // coef_off counts 128-byte units, or dwords for a picture in the sparse form (MPEGHIP_PIC_SPARSE); pictures of both
    // forms may share the buffer: a unit-form picture starts on a unit boundary
    const bool sparse = (pic.flags & MPEGHIP_PIC_SPARSE) != 0;
    if (!sparse)
        coefs_.resize((coefs_.size() + MPEGHIP_COEF_UNIT - 1) / MPEGHIP_COEF_UNIT * MPEGHIP_COEF_UNIT);
    else
        any_sparse_queued_ = true;
    const uint32_t unit0 = (uint32_t)(coefs_.size() / (sparse ? 4 : MPEGHIP_COEF_UNIT));
    mpeghip_pic_desc p = pic;
    p.stream = stream;
    p.mb_first = mb_first;
    p.mb_count = n_mbs;
    pics_.push_back(p);
    mbs_.insert(mbs_.end(), mbs, mbs + n_mbs);
    for (uint32_t i = mb_first; i < mb_first + n_mbs; i++) {
        mbs_[i].pic = pic_index;
        mbs_[i].coef_off += unit0;
    }
    coefs_.insert(coefs_.end(), coefs, coefs + coef_bytes);
    pending_[stream] = 1;
    queued_pictures_++;
}

void VideoBatch::Flush()
{
    if (pics_.empty())
        return;
    if (any_sparse_queued_) // (coef_bytes: a multiple of 128 as soon as one picture of the call is in the unit form)
        coefs_.resize((coefs_.size() + MPEGHIP_COEF_UNIT - 1) / MPEGHIP_COEF_UNIT * MPEGHIP_COEF_UNIT);
    any_sparse_queued_ = false;
    store_->submit(pics_.data(), (uint32_t)pics_.size(), mbs_.data(), (uint32_t)mbs_.size(), coefs_.data(), coefs_.size());
    device_submits_++;
    pics_.clear();
    mbs_.clear();
    coefs_.clear();
    std::fill(pending_.begin(), pending_.end(), 0);
}

// the verdict of the device-packed commits so far (their validation; the reconstruction goes on behind it)
void VideoBatch::reapVerdict()
{
    if (!verdict_owed_)
        return;
    verdict_owed_ = false;
    try {
        store_->verdict();
    } catch (...) {
        refused_streams_ = store_->refusedStreams();
        throw;
    }
}

void VideoBatch::Sync()
{
    if (held_refusal_) {
        std::exception_ptr e = held_refusal_;
        held_refusal_ = nullptr;
        std::rethrow_exception(e);
    }
    verdict_owed_ = false;
    try {
        store_->sync();
    } catch (...) {
        refused_streams_ = store_->refusedStreams();
        throw;
    }
}

size_t VideoBatch::DecodeAll(std::vector<Frame *> &frames, bool fetch)
{
    const size_t n = videos_.size();
    frames.assign(n, nullptr);
    // The state of a tick lives in the batch (round_): a call that throws a REFUSAL — the device's verdict on the round before —
    // has parsed its round already and keeps it; the next call takes up exactly there (it commits the held round instead of
    // parsing) and returns the tick's frames.
    if (held_refusal_) { // (learnt while the previous tick's frames were fetched)
        std::exception_ptr e = held_refusal_;
        held_refusal_ = nullptr;
        std::rethrow_exception(e);
    }
    std::vector<uint32_t> &slot = round_.slot, &todo = round_.todo;
    std::vector<double> &time = round_.time;
    std::vector<uint8_t> &got = round_.got;
    std::vector<int> &result = round_.result;
    if (!round_.held) {
        slot.assign(n, 0);
        time.assign(n, 0.0);
        got.assign(n, 0);
        // Rounds of "every stream that still owes a frame parses ONE picture, then one device call": a tick
        // costs as many calls as the neediest stream has pictures in it (1; 2 at a stream's start), not one
        // per stream.
        todo.resize(n);
        for (size_t i = 0; i < n; i++)
            todo[i] = (uint32_t)i;
        result.assign(n, 0);
    } else if (got.size() != n) { // (a stream was added while a round was held: it joins at the next tick)
        slot.resize(n, 0);
        time.resize(n, 0.0);
        got.resize(n, 0);
        result.resize(n, 0);
    }
    while (!todo.empty()) {
        std::vector<uint32_t> again;
        if (pool_ && todo.size() > 1) {
            if (!round_.held) {
            // parse on the pool, every stream recording its own device requests ...
            for (uint32_t i : todo)
                ports_[i]->recording = true;
            std::exception_ptr failed;
            const double p0 = nowSeconds();
            try {
                pool_->run(todo.size(), [&](size_t k) {
                    const uint32_t i = todo[k];
                    result[i] = videos_[i]->DecodeStep(&slot[i], &time[i]);
                });
            } catch (...) {
                failed = std::current_exception();
            }
            t_parse_ += nowSeconds() - p0;
            for (uint32_t i : todo)
                ports_[i]->recording = false;
            if (failed) {
                for (uint32_t i : todo)
                    ports_[i]->n_events = 0;
                std::rethrow_exception(failed);
            }
            }
            round_.held = false;
            // ... the device's verdict on the commit BEFORE this round (its pictures have been there for a whole parse: no wait in
            // practice — asked for before the parse, the same question cost 1.4 ms of idle host per round, 14 % of the
            // pictures per second).  A refusal is thrown here, with this round parsed but not handed over: nothing of it is
            // lost, the next call commits it.  RefusedStreams() names the streams; every other stream's picture of the
            // refused commit was reconstructed.
            try {
                reapVerdict();
            } catch (...) {
                round_.held = true;
                throw;
            }
            // ... then replay them here: the k-th request of every stream, in stream order (requests of
            // different streams commute; two pictures of one stream never share a device call).  Pictures go
            // to the device as one staged submit per step, put into the pinned staging buffer by the pool
            // (validation, record expansion and the copies run in parallel); stores without staging get
            // them merged through queue().
            size_t most = 0;
            for (uint32_t i : todo)
                most = ports_[i]->n_events > most ? ports_[i]->n_events : most;
            std::vector<const Port::Event *> group;
            std::vector<uint32_t> group_stream;
            std::vector<uint8_t> in_group(videos_.size(), 0);
            auto run_group = [&]() {
                if (group.empty())
                    return;
                Flush(); // pictures queued the other way come first
                if (group.size() == 1 || !store_->canStage()) {
                    for (size_t g = 0; g < group.size(); g++)
                        queue(group_stream[g], group[g]->pic, group[g]->mbs.data(), (uint32_t)group[g]->mbs.size(),
                              group[g]->coefs.data(), group[g]->coefs.size());
                    Flush();
                } else {
                    // about kStageSliceBytes of coefficients per staged submit: the device starts on the first slice
                    // while the pool packs the next, and staging buffers stay in the size range that copies fastest
                    constexpr size_t kStageSliceBytes = (size_t)128 << 20;
                    for (size_t g0 = 0, gn = 0; g0 < group.size(); g0 += gn) {
                        size_t sum = 0;
                        for (gn = 0; g0 + gn < group.size() && (gn == 0 || sum < kStageSliceBytes); gn++)
                            sum += group[g0 + gn]->coefs.size() + group[g0 + gn]->mbs.size() * sizeof(mpeghip_mb_desc);
                        std::vector<uint32_t> n_mbs(gn);
                        std::vector<size_t> bytes(gn);
                        for (size_t g = 0; g < gn; g++) {
                            n_mbs[g] = (uint32_t)group[g0 + g]->mbs.size();
                            bytes[g] = group[g0 + g]->coefs.size();
                        }
                        bool all_sparse = true;
                        for (size_t g = 0; g < gn; g++)
                            all_sparse = all_sparse && (group[g0 + g]->pic.flags & MPEGHIP_PIC_SPARSE) != 0;
                        const double s0 = nowSeconds();
                        const bool on_device = device_pack_ && all_sparse;
                        store_->stageBegin(n_mbs, bytes, on_device);
                        const double s1 = nowSeconds();
                        t_begin_ += s1 - s0;
                        std::exception_ptr put_failed;
                        try {
                            pool_->run(gn, [&](size_t g) {
                                const Port::Event *e = group[g0 + g];
                                mpeghip_pic_desc p = e->pic;
                                p.stream = group_stream[g0 + g];
                                if (__builtin_expect(debug_damage_.load(std::memory_order_relaxed) == (int64_t)p.stream && !e->mbs.empty(), 0)) { // (test hook)
                                    debug_damage_.store(-1);
                                    std::vector<mpeghip_mb_desc> bad(e->mbs);
                                    bad[bad.size() / 2].qscale = 0;
                                    store_->stagePut((uint32_t)g, p, bad.data(), e->coefs.data());
                                    return;
                                }
                                store_->stagePut((uint32_t)g, p, e->mbs.data(), e->coefs.data());
                            });
                        } catch (...) {
                            put_failed = std::current_exception();
                        }
                        const double s2 = nowSeconds();
                        t_put_ += s2 - s1;
                        try {
                            store_->stageCommit(); // ends the stage; fails (launching nothing) if a put failed
                        } catch (...) {
                            if (!put_failed)
                                put_failed = std::current_exception();
                        }
                        t_commit_ += nowSeconds() - s2;
                        if (put_failed)
                            std::rethrow_exception(put_failed);
                        verdict_owed_ = verdict_owed_ || on_device;
                        device_submits_++;
                        queued_pictures_ += gn;
                    }
                }
                for (uint32_t st : group_stream)
                    in_group[st] = 0;
                group.clear();
                group_stream.clear();
            };
            try {
                for (size_t k = 0; k < most; k++)
                    for (uint32_t i : todo) {
                        if (k >= ports_[i]->n_events)
                            continue;
                        const Port::Event &e = ports_[i]->events[k];
                        if (e.kind != Port::Event::Submit) {
                            run_group(); // (a quantiser table belongs to pictures not yet on the device)
                            ports_[i]->replay(e);
                            continue;
                        }
                        if (in_group[i])
                            run_group();
                        group.push_back(&e);
                        group_stream.push_back(i);
                        in_group[i] = 1;
                    }
                run_group();
            } catch (...) {
                for (uint32_t i : todo)
                    ports_[i]->n_events = 0;
                throw;
            }
            for (uint32_t i : todo)
                ports_[i]->n_events = 0;
        } else {
            reapVerdict(); // (this path hands over through submit(): a verdict can only be owed from a pooled round before it)
            for (uint32_t i : todo)
                result[i] = videos_[i]->DecodeStep(&slot[i], &time[i]);
        }
        for (uint32_t i : todo) {
            if (result[i] == 1)
                got[i] = 1;
            else if (result[i] == 2)
                again.push_back(i);
        }
        Flush();
        todo.swap(again);
    }
    // With fetch the frames' read-backs below wait for the device, and the first of them would report a refusal of THIS tick's
    // commit from the middle of the loop: ask now, keep the answer for the next call (which throws it before it parses).
    if (fetch) {
        try {
            reapVerdict();
        } catch (...) {
            held_refusal_ = std::current_exception();
        }
    }
    size_t produced = 0;
    for (size_t i = 0; i < n; i++)
        if (got[i]) {
            frames[i] = videos_[i]->Fetch(slot[i], time[i], fetch);
            produced++;
        }
    return produced;
}

// ------------------------------------------------------------------ AudioBatch
namespace {
size_t elemSize(AudioFormat f) { return f == AudioS16 ? 2 : 4; }
int abiFormat(AudioFormat f)
{
    switch (f) {
    case AudioF32NLR: return MPEGHIP_AUDIO_F32NLR;
    case AudioF32: return MPEGHIP_AUDIO_F32;
    case AudioS16: return MPEGHIP_AUDIO_S16;
    default: return MPEGHIP_AUDIO_F32N;
    }
}
} // namespace

// An Audio's synthesis request is only RECORDED (samples copied, destination remembered); the batch's
// Flush() runs all recorded streams in one call and scatters the results.
class AudioBatch::Port : public AudioBackend {
public:
    Port(AudioBatch *b, uint32_t stream) : b_(b), stream_(stream) {}
    void synth(const int32_t *samples, int format, void *out, void *out2) override
    {
        if (format != abiFormat(b_->format_))
            throw std::runtime_error("AudioBatch: a stream changed its output format");
        if (b_->active_[stream_]) {
            if (b_->parallel_)
                throw std::logic_error("AudioBatch: a stream produced two frames in one pooled parse");
            b_->Flush(); // two frames of one stream never share a device call
        }
        memcpy(b_->in_ + (size_t)stream_ * MPEGHIP_AUDIO_FRAME_INTS, samples, MPEGHIP_AUDIO_FRAME_INTS * sizeof(int32_t));
        b_->active_[stream_] = 1;
        b_->dest_[stream_].out = out;
        b_->dest_[stream_].out2 = out2;
    }

private:
    AudioBatch *b_;
    uint32_t stream_;
};

AudioBatch::AudioBatch(Device *dev, uint32_t n_streams, AudioFormat format, int fma_mode)
    : AudioBatch(dev->newAudioBatchStore(), n_streams, format, fma_mode)
{
}

AudioBatch::AudioBatch(std::unique_ptr<AudioBatchStore> store, uint32_t n_streams, AudioFormat format, int fma_mode)
    : store_(std::move(store)), capacity_(n_streams), format_(format)
{
    if (n_streams == 0)
        throw std::runtime_error("AudioBatch: n_streams is 0");
    store_->open(n_streams, fma_mode);
    in_ = static_cast<int32_t *>(store_->allocHost((size_t)n_streams * MPEGHIP_AUDIO_FRAME_INTS * sizeof(int32_t)));
    out_ = static_cast<uint8_t *>(store_->allocHost((size_t)n_streams * 2304 * elemSize(format)));
    if (!in_ || !out_) {
        store_->freeHost(in_);
        store_->freeHost(out_);
        throw std::runtime_error("AudioBatch: no memory for the batch's host arrays");
    }
    active_.assign(n_streams, 0);
    dest_.assign(n_streams, Dest());
}

AudioBatch::~AudioBatch()
{
    pool_.reset();
    store_->freeHost(in_);
    store_->freeHost(out_);
}

Audio *AudioBatch::AddStream(Buffer *buf)
{
    if (audios_.size() >= capacity_)
        throw std::runtime_error("AudioBatch: more streams than the batch was opened for");
    const uint32_t idx = (uint32_t)audios_.size();
    audios_.emplace_back(new Audio(buf, std::unique_ptr<AudioBackend>(new Port(this, idx))));
    audios_.back()->SetLookahead(false); // (a tick parses ONE frame of every stream: the batch is the overlap)
    audios_.back()->SetFormat(format_);
    return audios_.back().get();
}

void AudioBatch::Flush()
{
    bool any = false;
    for (uint8_t a : active_)
        any = any || a;
    if (!any)
        return;
    store_->synth(in_, active_.data(), abiFormat(format_), out_);
    device_calls_++;
    const size_t es = elemSize(format_);
    auto scatter = [&](size_t i) { // the stream's samples from the batch's array into its own Samples
        if (!active_[i])
            return;
        const uint8_t *src = out_ + i * 2304 * es;
        if (format_ == AudioF32NLR) {
            memcpy(dest_[i].out, src, 1152 * es);
            memcpy(dest_[i].out2, src + 1152 * es, 1152 * es);
        } else {
            memcpy(dest_[i].out, src, 2304 * es);
        }
        active_[i] = 0;
    };
    if (pool_ && !parallel_ && capacity_ > 1) {
        pool_->run(capacity_, scatter); // (9 KB per stream: with hundreds of streams the copies are a third of a tick on one thread)
    } else {
        for (uint32_t i = 0; i < capacity_; i++)
            scatter(i);
    }
}

size_t AudioBatch::DecodeAll(std::vector<Samples *> &samples)
{
    const size_t n = audios_.size();
    samples.assign(n, nullptr);
    size_t produced = 0;
    if (pool_ && n > 1) {            // CPU: parse, record — every stream into its own slot, side by side
        Flush();                     // a frame a caller decoded directly (Audio::Decode on a batch stream) goes out first, as the
                                     // one-thread path does it: a slot holds one frame, and the pool cannot flush from inside
        parallel_ = true;
        std::exception_ptr failed;
        try {
            pool_->run(n, [&](size_t i) { samples[i] = audios_[i]->Decode(); });
        } catch (...) {
            failed = std::current_exception();
        }
        parallel_ = false;
        if (failed) {
            std::fill(active_.begin(), active_.end(), 0);
            std::rethrow_exception(failed);
        }
        for (size_t i = 0; i < n; i++)
            produced += samples[i] ? 1 : 0;
    } else {
        for (size_t i = 0; i < n; i++) {
            samples[i] = audios_[i]->Decode();
            produced += samples[i] ? 1 : 0;
        }
    }
    Flush();                         // GPU: one call for all streams
    return produced;
}

void AudioBatch::SetThreads(unsigned n)
{
    n = poolThreads(n);
    if (n == threads_)
        return;
    pool_.reset(n > 1 ? new HostPool(n, -1) : nullptr);
    threads_ = n;
}

// -------------------------------------------------------------------- ShardedVideoBatch
struct ShardedVideoBatch::ShardState {
    std::unique_ptr<VideoBatch> batch;
    std::vector<Frame *> frames;      // the shard's last tick, local stream order
    size_t produced = 0;
    std::exception_ptr error;
    // the shard's host thread: sleeps until a tick is posted
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    uint64_t posted = 0, done = 0;
    bool fetch = true, quit = false;
};

ShardedVideoBatch::ShardedVideoBatch(const std::vector<Device *> &devices, uint32_t n_streams)
{
    if (devices.empty())
        throw std::runtime_error("ShardedVideoBatch: no devices");
    const uint32_t per = (n_streams + (uint32_t)devices.size() - 1) / (uint32_t)devices.size();
    for (Device *d : devices) {
        shards_.emplace_back(new ShardState);
        shards_.back()->batch.reset(new VideoBatch(d, per ? per : 1));
        // the shard's host side — its tick thread and its parse pool — runs on the socket the GPU hangs off
        shards_.back()->batch->SetNumaNode(d->NumaNode());
    }
    start(n_streams);
}

ShardedVideoBatch::ShardedVideoBatch(std::vector<std::unique_ptr<BatchStore>> stores, uint32_t n_streams)
{
    if (stores.empty())
        throw std::runtime_error("ShardedVideoBatch: no stores");
    const uint32_t per = (n_streams + (uint32_t)stores.size() - 1) / (uint32_t)stores.size();
    for (auto &st : stores) {
        shards_.emplace_back(new ShardState);
        shards_.back()->batch.reset(new VideoBatch(std::move(st), per ? per : 1));
    }
    start(n_streams);
}

void ShardedVideoBatch::start(uint32_t n_streams)
{
    if (n_streams == 0)
        throw std::runtime_error("ShardedVideoBatch: n_streams is 0");
    capacity_ = n_streams;
    try {
    for (auto &sp : shards_) {
        ShardState *sh = sp.get();
        sh->worker = std::thread([sh]() {
            pinThisThreadToNode(sh->batch->NumaNode());
            uint64_t seen = 0;
            for (;;) {
                bool fetch;
                {
                    std::unique_lock<std::mutex> l(sh->m);
                    sh->cv.wait(l, [&] { return sh->quit || sh->posted != seen; });
                    if (sh->quit)
                        return;
                    seen = sh->posted;
                    fetch = sh->fetch;
                }
                try { // every libmpeghip call selects its context's device itself: nothing else binds the thread
                    sh->produced = sh->batch->DecodeAll(sh->frames, fetch);
                } catch (...) {
                    sh->error = std::current_exception();
                    sh->produced = 0;
                }
                {
                    std::lock_guard<std::mutex> l(sh->m);
                    sh->done = seen;
                }
                sh->cv.notify_all();
            }
        });
    }
    } catch (...) { // a thread could not be created: the object is never constructed, so no destructor will join the earlier workers
        stopWorkers();
        throw;
    }
}

void ShardedVideoBatch::stopWorkers()
{
    for (auto &sh : shards_) {
        {
            std::lock_guard<std::mutex> l(sh->m);
            sh->quit = true;
        }
        sh->cv.notify_all();
        if (sh->worker.joinable())
            sh->worker.join();
    }
}

ShardedVideoBatch::~ShardedVideoBatch() { stopWorkers(); }

VideoBatch &ShardedVideoBatch::Shard(uint32_t g) { return *shards_.at(g)->batch; }

Video *ShardedVideoBatch::AddStream(Buffer *buf)
{
    if (n_added_ >= capacity_)
        throw std::runtime_error("ShardedVideoBatch: more streams than the batch was opened for");
    Video *v = shards_[n_added_ % Shards()]->batch->AddStream(buf);
    n_added_++;
    return v;
}

// The shards' pools share ONE budget — the CPU time of the process (EffectiveCores): n threads in all (0: as many as fit), divided
// among the shards, at least one each.  (Round 5 forwarded n to every shard: G shards started up to G x the quota, the throttled
// oversubscription the cap exists to prevent.)
void ShardedVideoBatch::SetThreads(unsigned n)
{
    const unsigned shards = (unsigned)shards_.size();
    if (!shards)
        return;
    const unsigned fit = (unsigned)std::ceil(EffectiveCores());
    const unsigned total = n == 0 || n > fit ? (fit < 1 ? 1 : fit) : n;
    for (unsigned i = 0; i < shards; i++) {
        const unsigned share = total / shards + (i < total % shards ? 1 : 0);
        shards_[i]->batch->SetThreads(share < 1 ? 1 : share);
    }
}

unsigned ShardedVideoBatch::Threads() const
{
    unsigned total = 0;
    for (auto &sh : shards_)
        total += sh->batch->Threads();
    return total;
}

void ShardedVideoBatch::SetDevicePack(bool on)
{
    for (auto &sh : shards_)
        sh->batch->SetDevicePack(on);
}

void ShardedVideoBatch::Sync()
{
    std::exception_ptr failed;
    for (auto &sh : shards_) { // (no tick is running: DecodeAll returns only when every shard's has ended)
        try {
            sh->batch->Sync();
        } catch (...) {
            if (!failed)
                failed = std::current_exception();
        }
    }
    if (failed)
        std::rethrow_exception(failed);
}

size_t ShardedVideoBatch::DecodeAll(std::vector<Frame *> &frames, bool fetch)
{
    for (auto &sh : shards_) {
        {
            std::lock_guard<std::mutex> l(sh->m);
            sh->fetch = fetch;
            sh->error = nullptr;
            sh->posted++;
        }
        sh->cv.notify_all();
    }
    size_t produced = 0;
    std::exception_ptr failed;
    for (auto &sh : shards_) {
        std::unique_lock<std::mutex> l(sh->m);
        sh->cv.wait(l, [&] { return sh->done == sh->posted; });
        produced += sh->produced;
        if (sh->error && !failed)
            failed = sh->error;
    }
    if (failed)
        std::rethrow_exception(failed);
    frames.assign(n_added_, nullptr);
    const uint32_t G = Shards();
    for (uint32_t s = 0; s < n_added_; s++) {
        const std::vector<Frame *> &f = shards_[s % G]->frames;
        frames[s] = s / G < f.size() ? f[s / G] : nullptr;
    }
    return produced;
}

} // namespace mpeg
