// staged_rate.cpp — tools/hostbench/libhostbench.so: a host-side measurement driver over the C ABI of libmpeghip.
// Built on demand by tools/hostbench/__init__.py (g++, in-tree); the product libraries do not contain it.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "mpeghip.h"

extern "C" {

// Measurement aid (tools/bench_host_path.py, bench.py --host-fed-seconds) — NOT part of the product libraries: how many pictures per second `threads` host threads can push
// through mpeghip_video_stage_* when every call carries one picture for each of n_streams streams.  The
// pictures are the caller's n_steps synthetic ones (arrays of arrays), the same for every stream — each stream
// still gets its own put (validation + packing into the device format, into pinned staging).  Returns
// pictures/s, < 0 on error; *wire_bytes_per_s = bytes that crossed to the device per second.
double hostbench_staged_submit_rate(int device, uint32_t width, uint32_t height, uint32_t n_streams, uint32_t threads,
                                   double seconds, uint32_t n_steps, const mpeghip_pic_desc *pics,
                                   const mpeghip_mb_desc *const *mbs, const uint32_t *n_mbs, const uint8_t *const *coefs,
                                   const size_t *coef_bytes, int verbose, int device_pack)
{
    // device_pack: 0 = the host validates and packs (mpeghip_video_stage_begin[_sparse]); 1 = a DEVICE-PACKED stage, the puts
    // copy the arrays into pinned staging (mpeghip_video_stage_begin_device + _put_sparse; sparse pictures only); 2 = the same
    // with the arrays written through mpeghip_video_stage_map ONCE per staging buffer and only put_mapped per call afterwards:
    // what is left when a parser writes its pictures in place — commit, PCIe, pack kernel, reconstruction
    try {
        mpeghip_ctx *ctx = nullptr;
        if (mpeghip_ctx_create(device, nullptr, &ctx) != MPEGHIP_OK)
            throw std::runtime_error(mpeghip_last_error());
        mpeghip_video *v = nullptr;
        if (mpeghip_video_open(ctx, width, height, n_streams, &v) != MPEGHIP_OK) {
            mpeghip_ctx_destroy(ctx);
            throw std::runtime_error(mpeghip_last_error());
        }
        struct Shared {
            std::mutex m;
            std::condition_variable go, done;
            uint64_t generation = 0;
            uint32_t busy = 0;
            bool stop = false;
            std::atomic<uint32_t> next{0};
            std::atomic<int> failed{0};
            mpeghip_stage *stage = nullptr;
            uint32_t step = 0;
            bool fill = true;
        } sh;
        threads = threads < 1 ? 1 : threads;
        auto drain = [&]() {
            for (;;) {
                const uint32_t i = sh.next.fetch_add(1);
                if (i >= n_streams)
                    return;
                mpeghip_pic_desc p = pics[sh.step];
                p.stream = i;
                if (device_pack == 2) {
                    if (sh.fill) { // (the first two calls of a step: both staging buffers get the picture's arrays)
                        mpeghip_mb_desc *pm = nullptr;
                        uint32_t *pw = nullptr;
                        if (mpeghip_video_stage_map(sh.stage, i, &pm, &pw) != MPEGHIP_OK) {
                            sh.failed.store(1);
                            continue;
                        }
                        memcpy(pm, mbs[sh.step], (size_t)n_mbs[sh.step] * sizeof(mpeghip_mb_desc));
                        memcpy(pw, coefs[sh.step], coef_bytes[sh.step]);
                    }
                    if (mpeghip_video_stage_put_mapped(sh.stage, i, &p) != MPEGHIP_OK)
                        sh.failed.store(1);
                    continue;
                }
                if (mpeghip_video_stage_put(sh.stage, i, &p, mbs[sh.step], coefs[sh.step]) != MPEGHIP_OK)
                    sh.failed.store(1);
            }
        };
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < threads; t++)
            pool.emplace_back([&]() {
                uint64_t seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> l(sh.m);
                        sh.go.wait(l, [&] { return sh.stop || sh.generation != seen; });
                        if (sh.stop)
                            return;
                        seen = sh.generation;
                    }
                    drain();
                    {
                        std::lock_guard<std::mutex> l(sh.m);
                        sh.busy--;
                    }
                    sh.done.notify_one();
                }
            });
        std::vector<uint32_t> counts(n_streams);
        std::vector<size_t> bytes(n_streams);
        double t_begin = 0, t_put = 0, t_commit = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double>(b - a).count();
        };
        auto one_call = [&](uint32_t step) {
            std::fill(counts.begin(), counts.end(), n_mbs[step]);
            std::fill(bytes.begin(), bytes.end(), coef_bytes[step]);
            mpeghip_stage *st = nullptr;
            const auto c0 = now();
            if (device_pack) {
                for (size_t &b : bytes)
                    b /= 4; // (dwords)
                if (mpeghip_video_stage_begin_device(v, n_streams, counts.data(), bytes.data(), &st) != MPEGHIP_OK)
                    throw std::runtime_error(mpeghip_last_error());
            } else if (mpeghip_video_stage_begin(v, n_streams, counts.data(), bytes.data(), &st) != MPEGHIP_OK)
                throw std::runtime_error(mpeghip_last_error());
            const auto c1 = now();
            {
                std::lock_guard<std::mutex> l(sh.m);
                sh.stage = st;
                sh.step = step;
                sh.next.store(0);
                sh.busy = (uint32_t)pool.size();
                sh.generation++;
            }
            sh.go.notify_all();
            drain();
            {
                std::unique_lock<std::mutex> l(sh.m);
                sh.done.wait(l, [&] { return sh.busy == 0; });
            }
            const auto c2 = now();
            if (mpeghip_video_stage_commit(st) != MPEGHIP_OK || sh.failed.load())
                throw std::runtime_error(mpeghip_last_error());
            const auto c3 = now();
            t_begin += secs(c0, c1);
            t_put += secs(c1, c2);
            t_commit += secs(c2, c3);
        };
        double rate = -1;
        std::exception_ptr err;
        try {
            if (device_pack == 2 && n_steps != 1)
                throw std::runtime_error("hostbench: the in-place mode cycles ONE picture (both staging buffers hold it)");
            for (uint32_t s = 0; s < (device_pack == 2 ? 2 : n_steps); s++)
                one_call(device_pack == 2 ? 0 : s);
            sh.fill = false;
            if (mpeghip_video_sync(v) != MPEGHIP_OK)
                throw std::runtime_error(mpeghip_last_error());
            t_begin = t_put = t_commit = 0;
            const auto t0 = std::chrono::steady_clock::now();
            uint64_t n = 0;
            double dt = 0;
            do {
                for (uint32_t s = 0; s < n_steps; s++) {
                    one_call(s);
                    n += n_streams;
                }
                dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            } while (dt < seconds);
            if (mpeghip_video_sync(v) != MPEGHIP_OK) // (a device-packed commit's deferred verdict)
                throw std::runtime_error(mpeghip_last_error());
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            rate = (double)n / dt;
            if (verbose)
                fprintf(stderr, "staged rate: %u pictures/call, %u threads: of %.3f s, begin (waits for the staging buffer's "
                                "previous use) %.3f, puts %.3f, commit %.3f\n", n_streams, threads, dt, t_begin, t_put, t_commit);
        } catch (...) {
            err = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> l(sh.m);
            sh.stop = true;
        }
        sh.go.notify_all();
        for (std::thread &t : pool)
            t.join();
        mpeghip_video_close(v);
        mpeghip_ctx_destroy(ctx);
        if (err)
            std::rethrow_exception(err);
        return rate;
    } catch (const std::exception &e) {
        fprintf(stderr, "hostbench: %s\n", e.what());
        return -1.0;
    }
}


} // extern "C"
