// capi.cpp — flat C entry points over the C++ host mirror, for ctypes-driven tests
// and for non-C++ callers.  Every function returns NULL / 0 on failure and keeps a
// thread-local message (mpeghost_last_error).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "mpeg.hpp"

using namespace mpeg;

#include "mpeghost.h"
typedef mpeghost_frame mpeghost_frame_t;

namespace {
thread_local std::string g_err;
template <class F>
auto guard(F f, decltype(f()) fail) -> decltype(f())
{
    try {
        return f();
    } catch (const std::exception &e) {
        g_err = e.what();
        return fail;
    }
}

struct VideoHandle {
    std::unique_ptr<Buffer> buf;
    std::unique_ptr<Video> video;
    Frame *last = nullptr;
};
struct AudioHandle {
    std::unique_ptr<Buffer> buf;
    std::unique_ptr<Audio> audio;
};
struct MpegHandle {
    std::unique_ptr<MPEG> m;
    int video_calls = 0, audio_calls = 0; // callbacks installed by mpeghost_mpeg_count_callbacks
};
struct BatchHandle {
    std::vector<std::unique_ptr<Buffer>> bufs;
    std::unique_ptr<VideoBatch> batch;
    std::vector<Frame *> frames;
};
struct AudioBatchHandle {
    std::vector<std::unique_ptr<Buffer>> bufs;
    std::unique_ptr<AudioBatch> batch;
    std::vector<Samples *> samples;
};
struct ShardedHandle {
    std::vector<std::unique_ptr<Buffer>> bufs;
    std::unique_ptr<ShardedVideoBatch> batch;
    std::vector<Frame *> frames;
};
struct DemuxHandle {
    std::unique_ptr<Buffer> buf;
    std::unique_ptr<Demux> demux;
};
MPEG *M(void *h) { return static_cast<MpegHandle *>(h)->m.get(); }
void fill(mpeghost_frame_t *out, const Frame *f);
} // namespace

extern "C" {

const char *mpeghost_last_error(void) { return g_err.c_str(); }

void *mpeghost_device_create(int ordinal)
{
    return guard([&]() -> void * { return new Device(ordinal); }, (void *)nullptr);
}
void mpeghost_device_destroy(void *d) { delete static_cast<Device *>(d); }

// NewVideo over a complete elementary stream (the way TestVideoGolden builds it, mpeg_test.go:206-213)
void *mpeghost_video_open(void *device, const uint8_t *data, size_t len)
{
    return guard([&]() -> void * {
        std::unique_ptr<VideoHandle> h(new VideoHandle());
        h->buf = Buffer::FromMemory(data, len);
        h->video.reset(new Video(h->buf.get(), static_cast<Device *>(device)));
        return h.release();
    }, (void *)nullptr);
}
// same, with a caller-supplied backend (ownership passes to the decoder) — used by tests/host_emu
void *mpeghost_video_open_backend(void *backend, const uint8_t *data, size_t len)
{
    return guard([&]() -> void * {
        std::unique_ptr<VideoBackend> be(static_cast<VideoBackend *>(backend)); // ours from here on, whatever throws
        std::unique_ptr<VideoHandle> h(new VideoHandle());
        h->buf = Buffer::FromMemory(data, len);
        h->video.reset(new Video(h->buf.get(), std::move(be)));
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_video_close(void *h) { delete static_cast<VideoHandle *>(h); }
int mpeghost_video_width(void *h) { return static_cast<VideoHandle *>(h)->video->Width(); }
int mpeghost_video_height(void *h) { return static_cast<VideoHandle *>(h)->video->Height(); }
double mpeghost_video_framerate(void *h) { return static_cast<VideoHandle *>(h)->video->Framerate(); }
void mpeghost_video_set_no_delay(void *h, int v) { static_cast<VideoHandle *>(h)->video->SetNoDelay(v != 0); }
void mpeghost_video_set_sparse(void *h, int v) { static_cast<VideoHandle *>(h)->video->SetSparse(v != 0); }
void mpeghost_set_default_sparse(int v) { Video::SetDefaultSparse(v != 0); }
uint64_t mpeghost_debug_vlc_self_check(void) { return Video::VlcSelfCheck(); }
int mpeghost_debug_vlc_decode(int table, uint64_t window, int *value, int *len) { return Video::VlcDecode(table, window, value, len) ? 0 : -1; }
int mpeghost_video_decode(void *hv, mpeghost_frame *out)
{
    return guard([&]() -> int {
        auto *h = static_cast<VideoHandle *>(hv);
        Frame *f = h->video->Decode();
        h->last = f;
        if (!f)
            return 0;
        out->time = f->Time;
        out->width = f->Width;
        out->height = f->Height;
        out->luma_w = f->Y.Width;
        out->luma_h = f->Y.Height;
        out->chroma_w = f->Cb.Width;
        out->chroma_h = f->Cb.Height;
        out->y = f->Y.Data;
        out->cb = f->Cb.Data;
        out->cr = f->Cr.Data;
        out->luma_bytes = f->Y.Len;
        out->chroma_bytes = f->Cb.Len;
        return 1;
    }, -1);
}
const uint8_t *mpeghost_video_rgba(void *hv)
{
    return guard([&]() -> const uint8_t * {
        auto *h = static_cast<VideoHandle *>(hv);
        return h->last ? h->last->RGBA() : nullptr;
    }, (const uint8_t *)nullptr);
}
void mpeghost_video_stats(void *hv, uint64_t out[8])
{
    const VideoStats &s = static_cast<VideoHandle *>(hv)->video->Stats();
    out[0] = s.pictures;
    out[1] = s.submits;
    out[2] = s.macroblocks;
    out[3] = s.coded_blocks;
    out[4] = s.raw_macroblocks;
    out[5] = s.invalid_blocks;
    out[6] = s.duplicate_splits;
    out[7] = s.range_skips;
}

// video.go:178-201 on a lone decoder: Time / Rewind / HasEnded, and the look-ahead switch of Video::Decode
double mpeghost_video_time(void *h) { return static_cast<VideoHandle *>(h)->video->Time(); }
int mpeghost_video_has_ended(void *h) { return static_cast<VideoHandle *>(h)->video->HasEnded() ? 1 : 0; }
void mpeghost_video_rewind(void *hv)
{
    auto *h = static_cast<VideoHandle *>(hv);
    h->last = nullptr;
    h->video->Rewind();
}
void mpeghost_video_set_lookahead(void *h, int on) { static_cast<VideoHandle *>(h)->video->SetLookahead(on != 0); }
void mpeghost_video_set_host_mirror(void *h, int on) { static_cast<VideoHandle *>(h)->video->SetHostMirror(on != 0); }
void mpeghost_video_set_device_pack_from(void *h, uint32_t n_mbs) { static_cast<VideoHandle *>(h)->video->SetDevicePackFrom(n_mbs); }
// wall seconds of the decoder's host phases so far: parse, hand-over (submit), frames back (read)
void mpeghost_video_phase_seconds(void *hv, double out[3])
{
    const VideoStats &s = static_cast<VideoHandle *>(hv)->video->Stats();
    out[0] = s.seconds_parse;
    out[1] = s.seconds_submit;
    out[2] = s.seconds_read;
}

// NewAudio over a complete elementary stream (TestAudioGolden, mpeg_test.go:167-173)
void *mpeghost_audio_open(void *device, const uint8_t *data, size_t len, int fma_mode, int format)
{
    return guard([&]() -> void * {
        std::unique_ptr<AudioHandle> h(new AudioHandle());
        h->buf = Buffer::FromMemory(data, len);
        h->audio.reset(new Audio(h->buf.get(), static_cast<Device *>(device), fma_mode));
        h->audio->SetFormat((AudioFormat)format);
        return h.release();
    }, (void *)nullptr);
}
void *mpeghost_audio_open_backend(void *backend, const uint8_t *data, size_t len, int format)
{
    return guard([&]() -> void * {
        std::unique_ptr<AudioBackend> be(static_cast<AudioBackend *>(backend)); // ours from here on, whatever throws
        std::unique_ptr<AudioHandle> h(new AudioHandle());
        h->buf = Buffer::FromMemory(data, len);
        h->audio.reset(new Audio(h->buf.get(), std::move(be)));
        h->audio->SetFormat((AudioFormat)format);
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_audio_close(void *h) { delete static_cast<AudioHandle *>(h); }
int mpeghost_audio_samplerate(void *h) { return static_cast<AudioHandle *>(h)->audio->Samplerate(); }
int mpeghost_audio_channels(void *h) { return static_cast<AudioHandle *>(h)->audio->Channels(); }
// returns Samples.Interleaved (2304 floats) / S16 / F32 / Left+Right according to the format, or NULL at the end
const void *mpeghost_audio_decode(void *ha, double *time)
{
    return guard([&]() -> const void * {
        Samples *s = static_cast<AudioHandle *>(ha)->audio->Decode();
        if (!s)
            return nullptr;
        if (time)
            *time = s->Time;
        switch (s->format) {
        case AudioF32N: return s->Interleaved.data();
        case AudioF32: return s->F32.data();
        case AudioS16: return s->S16.data();
        default: return s->Left.data();
        }
    }, (const void *)nullptr);
}

double mpeghost_audio_time(void *h) { return static_cast<AudioHandle *>(h)->audio->Time(); }
int mpeghost_audio_has_ended(void *h) { return static_cast<AudioHandle *>(h)->audio->HasEnded() ? 1 : 0; }
void mpeghost_audio_rewind(void *h) { static_cast<AudioHandle *>(h)->audio->Rewind(); }
void mpeghost_audio_set_lookahead(void *h, int on) { static_cast<AudioHandle *>(h)->audio->SetLookahead(on != 0); }

// mpeg.New over a complete program stream
void *mpeghost_mpeg_open(void *device, const uint8_t *data, size_t len)
{
    return guard([&]() -> void * {
        std::unique_ptr<MpegHandle> h(new MpegHandle);
        h->m.reset(new MPEG(data, len, static_cast<Device *>(device)));
        return h.release();
    }, (void *)nullptr);
}
// same over injected backends (tests): the factories return `new`ed VideoBackend / AudioBackend objects
void *mpeghost_mpeg_open_backends(void *(*make_video)(void), void *(*make_audio)(int), const uint8_t *data, size_t len)
{
    return guard([&]() -> void * {
        MPEG::Backends b;
        b.video = [make_video]() { return std::unique_ptr<VideoBackend>(static_cast<VideoBackend *>(make_video())); };
        b.audio = [make_audio](int fma) { return std::unique_ptr<AudioBackend>(static_cast<AudioBackend *>(make_audio(fma))); };
        std::unique_ptr<MpegHandle> h(new MpegHandle);
        h->m.reset(new MPEG(data, len, std::move(b)));
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_mpeg_close(void *m) { delete static_cast<MpegHandle *>(m); }
void mpeghost_mpeg_info(void *mv, int out[6])
{
    MPEG *m = M(mv);
    out[0] = m->NumVideoStreams();
    out[1] = m->NumAudioStreams();
    out[2] = m->Width();
    out[3] = m->Height();
    out[4] = m->Samplerate();
    out[5] = m->Channels();
}
double mpeghost_mpeg_framerate(void *m) { return M(m)->Framerate(); }
void mpeghost_mpeg_set_enabled(void *m, int video, int audio)
{
    M(m)->SetVideoEnabled(video != 0);
    M(m)->SetAudioEnabled(audio != 0);
}
void mpeghost_mpeg_get_enabled(void *m, int out[2])
{
    out[0] = M(m)->VideoEnabled() ? 1 : 0;
    out[1] = M(m)->AudioEnabled() ? 1 : 0;
}
void mpeghost_mpeg_set_audio_stream(void *m, int stream_index)
{
    guard([&]() -> int { M(m)->SetAudioStream(stream_index); return 0; }, -1);
}
void mpeghost_mpeg_set_loop(void *m, int loop) { M(m)->SetLoop(loop != 0); }
int mpeghost_mpeg_loop(void *m) { return M(m)->Loop() ? 1 : 0; }
void mpeghost_mpeg_rewind(void *m)
{
    guard([&]() -> int { M(m)->Rewind(); return 0; }, -1);
}
int mpeghost_mpeg_decode_video(void *mv, mpeghost_frame *out)
{
    return guard([&]() -> int {
        Frame *f = M(mv)->DecodeVideo();
        if (!f)
            return 0;
        fill(out, f);
        return 1;
    }, -1);
}
const float *mpeghost_mpeg_decode_audio(void *mv, double *time)
{
    return guard([&]() -> const float * {
        Samples *s = M(mv)->DecodeAudio();
        if (!s)
            return nullptr;
        if (time)
            *time = s->Time;
        return s->Interleaved.data();
    }, (const float *)nullptr);
}
int mpeghost_mpeg_has_ended(void *m) { return M(m)->HasEnded() ? 1 : 0; }
int mpeghost_mpeg_take_done(void *m) { return M(m)->TakeDone() ? 1 : 0; }
int mpeghost_mpeg_audio_format(void *m) { return (int)M(m)->GetAudioFormat(); }
double mpeghost_mpeg_audio_lead_time(void *m) { return M(m)->AudioLeadTime(); }
void mpeghost_mpeg_set_audio_lead_time(void *m, double seconds) { M(m)->SetAudioLeadTime(seconds); }
void mpeghost_mpeg_set_audio_format(void *m, int format) { M(m)->SetAudioFormat((AudioFormat)format); }
int mpeghost_mpeg_probe(void *m, size_t probe_size)
{
    return guard([&]() -> int { return M(m)->Probe(probe_size) ? 1 : 0; }, -1);
}
int mpeghost_mpeg_has_headers(void *m)
{
    return guard([&]() -> int { return M(m)->HasHeaders() ? 1 : 0; }, -1);
}
double mpeghost_mpeg_duration(void *m) { return guard([&]() -> double { return M(m)->Duration(); }, -1.0); }
double mpeghost_mpeg_time(void *m) { return M(m)->Time(); }
double mpeghost_mpeg_audio_time(void *m) { return M(m)->GetAudio() ? M(m)->GetAudio()->Time() : -1.0; }
double mpeghost_mpeg_video_time(void *m) { return M(m)->GetVideo() ? M(m)->GetVideo()->Time() : -1.0; }
// install callbacks that only count (SetVideoCallback / SetAudioCallback), or remove them
void mpeghost_mpeg_count_callbacks(void *mv, int video, int audio)
{
    MpegHandle *h = static_cast<MpegHandle *>(mv);
    h->video_calls = h->audio_calls = 0;
    if (video)
        h->m->SetVideoCallback([h](MPEG *, Frame *) { h->video_calls++; });
    else
        h->m->SetVideoCallback(VideoFunc());
    if (audio)
        h->m->SetAudioCallback([h](MPEG *, Samples *) { h->audio_calls++; });
    else
        h->m->SetAudioCallback(AudioFunc());
}
void mpeghost_mpeg_callback_counts(void *mv, int out[2])
{
    MpegHandle *h = static_cast<MpegHandle *>(mv);
    out[0] = h->video_calls;
    out[1] = h->audio_calls;
}
void mpeghost_mpeg_decode(void *m, double tick)
{
    guard([&]() -> int { M(m)->Decode(tick); return 0; }, -1);
}
int mpeghost_mpeg_seek(void *m, double seconds, int exact)
{
    return guard([&]() -> int { return M(m)->Seek(seconds, exact != 0) ? 1 : 0; }, -1);
}
int mpeghost_mpeg_seek_frame(void *m, double seconds, int exact, mpeghost_frame *out)
{
    return guard([&]() -> int {
        Frame *f = M(m)->SeekFrame(seconds, exact != 0);
        if (!f)
            return 0;
        fill(out, f);
        return 1;
    }, -1);
}

// VideoBatch: many elementary streams of one picture size, one device call per tick
void *mpeghost_batch_open(void *device, uint32_t n_streams)
{
    return guard([&]() -> void * {
        std::unique_ptr<BatchHandle> h(new BatchHandle);
        h->batch.reset(new VideoBatch(static_cast<Device *>(device), n_streams));
        return h.release();
    }, (void *)nullptr);
}
// same over an injected BatchStore (tests); ownership of `store` passes to the batch
void *mpeghost_batch_open_store(void *store, uint32_t n_streams)
{
    return guard([&]() -> void * {
        std::unique_ptr<BatchHandle> h(new BatchHandle);
        h->batch.reset(new VideoBatch(std::unique_ptr<BatchStore>(static_cast<BatchStore *>(store)), n_streams));
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_batch_close(void *h) { delete static_cast<BatchHandle *>(h); }
int mpeghost_batch_add_stream(void *hv, const uint8_t *data, size_t len)
{
    return guard([&]() -> int {
        BatchHandle *h = static_cast<BatchHandle *>(hv);
        h->bufs.push_back(Buffer::FromMemory(data, len));
        h->batch->AddStream(h->bufs.back().get());
        return (int)h->batch->Streams() - 1;
    }, -1);
}
// advance every stream by one frame; returns the number of frames produced (-1 on error)
int mpeghost_batch_decode_all(void *hv, int fetch)
{
    return guard([&]() -> int {
        BatchHandle *h = static_cast<BatchHandle *>(hv);
        return (int)h->batch->DecodeAll(h->frames, fetch != 0);
    }, -1);
}
// CPU time the process gets, in cores (affinity mask capped by the cgroup quota): what pools are sized by
double mpeghost_effective_cores(void) { return mpeg::EffectiveCores(); }
// its cgroup part over a stand-in hierarchy (tests): root = what /sys/fs/cgroup would be, proc_file = /proc/self/cgroup's text
double mpeghost_cgroup_quota_cores(const char *root, const char *proc_file) { return mpeg::CgroupQuotaCores(root, proc_file); }
uint32_t mpeghost_batch_threads(void *hv) { return static_cast<BatchHandle *>(hv)->batch->Threads(); }
// host threads of the parse (VideoBatch::SetThreads)
void mpeghost_batch_set_threads(void *hv, uint32_t n)
{
    guard([&]() -> int {
        static_cast<BatchHandle *>(hv)->batch->SetThreads(n);
        return 0;
    }, -1);
}
// frame of stream i from the last decode_all: 1 and *out filled, or 0
int mpeghost_batch_frame(void *hv, uint32_t stream, mpeghost_frame *out)
{
    BatchHandle *h = static_cast<BatchHandle *>(hv);
    if (stream >= h->frames.size() || !h->frames[stream])
        return 0;
    fill(out, h->frames[stream]);
    return 1;
}
void mpeghost_batch_counters(void *hv, uint64_t out[2])
{
    BatchHandle *h = static_cast<BatchHandle *>(hv);
    out[0] = h->batch->DeviceSubmits();
    out[1] = h->batch->QueuedPictures();
}

void mpeghost_batch_phase_seconds(void *hv, double out[4]) { static_cast<BatchHandle *>(hv)->batch->PhaseSeconds(out); }
void mpeghost_batch_set_device_pack(void *hv, int on) { static_cast<BatchHandle *>(hv)->batch->SetDevicePack(on != 0); }
int mpeghost_batch_sync(void *hv)
{
    return guard([&]() -> int {
        static_cast<BatchHandle *>(hv)->batch->Sync();
        return 0;
    }, -1);
}
// the streams whose picture the refusal last reported (a -1 of decode_all / sync) named: up to cap of them; returns their number
uint32_t mpeghost_batch_refused_streams(void *hv, uint32_t *out, uint32_t cap)
{
    const std::vector<uint32_t> &r = static_cast<BatchHandle *>(hv)->batch->RefusedStreams();
    for (size_t i = 0; i < r.size() && i < cap; i++)
        out[i] = r[i];
    return (uint32_t)r.size();
}
int mpeghost_batch_device_pack(void *hv) { return static_cast<BatchHandle *>(hv)->batch->DevicePack() ? 1 : 0; }
// test hook (VideoBatch::DebugDamageNextPicture): the next picture of `stream` is damaged on its way to the device
void mpeghost_batch_debug_damage_next_picture(void *hv, uint32_t stream) { static_cast<BatchHandle *>(hv)->batch->DebugDamageNextPicture(stream); }
void mpeghost_batch_numa_pins(void *hv, uint32_t out[2])
{
    unsigned p[2];
    static_cast<BatchHandle *>(hv)->batch->NumaPins(p);
    out[0] = p[0];
    out[1] = p[1];
}

// ShardedVideoBatch: streams sharded over several devices (stream s -> device s mod G), one host thread per device
void *mpeghost_sharded_open(void *const *devices, uint32_t n_devices, uint32_t n_streams)
{
    return guard([&]() -> void * {
        std::vector<Device *> devs;
        for (uint32_t i = 0; i < n_devices; i++)
            devs.push_back(static_cast<Device *>(devices[i]));
        std::unique_ptr<ShardedHandle> h(new ShardedHandle);
        h->batch.reset(new ShardedVideoBatch(devs, n_streams));
        return h.release();
    }, (void *)nullptr);
}
void *mpeghost_sharded_open_stores(void *const *stores, uint32_t n_stores, uint32_t n_streams)
{
    return guard([&]() -> void * {
        std::vector<std::unique_ptr<BatchStore>> st; // ours from here on, whatever throws
        for (uint32_t i = 0; i < n_stores; i++)
            st.emplace_back(static_cast<BatchStore *>(stores[i]));
        std::unique_ptr<ShardedHandle> h(new ShardedHandle);
        h->batch.reset(new ShardedVideoBatch(std::move(st), n_streams));
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_sharded_close(void *h) { delete static_cast<ShardedHandle *>(h); }
int mpeghost_sharded_add_stream(void *hv, const uint8_t *data, size_t len)
{
    return guard([&]() -> int {
        ShardedHandle *h = static_cast<ShardedHandle *>(hv);
        h->bufs.push_back(Buffer::FromMemory(data, len));
        h->batch->AddStream(h->bufs.back().get());
        return (int)h->batch->Streams() - 1;
    }, -1);
}
void mpeghost_sharded_set_threads(void *hv, unsigned n)
{
    guard([&]() -> int {
        static_cast<ShardedHandle *>(hv)->batch->SetThreads(n);
        return 0;
    }, -1);
}
uint32_t mpeghost_sharded_threads(void *hv) { return static_cast<ShardedHandle *>(hv)->batch->Threads(); }
void mpeghost_sharded_set_device_pack(void *hv, int on)
{
    guard([&]() -> int {
        static_cast<ShardedHandle *>(hv)->batch->SetDevicePack(on != 0);
        return 0;
    }, -1);
}
int mpeghost_sharded_sync(void *hv)
{
    return guard([&]() -> int {
        static_cast<ShardedHandle *>(hv)->batch->Sync();
        return 0;
    }, -1);
}
int mpeghost_sharded_decode_all(void *hv, int fetch)
{
    return guard([&]() -> int {
        ShardedHandle *h = static_cast<ShardedHandle *>(hv);
        return (int)h->batch->DecodeAll(h->frames, fetch != 0);
    }, -1);
}
int mpeghost_sharded_frame(void *hv, uint32_t stream, mpeghost_frame *out)
{
    ShardedHandle *h = static_cast<ShardedHandle *>(hv);
    if (stream >= h->frames.size() || !h->frames[stream])
        return 0;
    fill(out, h->frames[stream]);
    return 1;
}
uint32_t mpeghost_sharded_device_of(void *hv, uint32_t stream) { return static_cast<ShardedHandle *>(hv)->batch->ShardOf(stream); }
void mpeghost_sharded_counters(void *hv, uint32_t shard, uint64_t out[2])
{
    out[0] = out[1] = 0;
    guard([&]() -> int { // (Shard() range-checks with at(): nothing throws across this boundary)
        VideoBatch &b = static_cast<ShardedHandle *>(hv)->batch->Shard(shard);
        out[0] = b.DeviceSubmits();
        out[1] = b.QueuedPictures();
        return 0;
    }, -1);
}

// AudioBatch: many MP2 streams, one synthesis call per tick (format: 0 F32N, 1 F32NLR, 2 F32, 3 S16)
void *mpeghost_audio_batch_open(void *device, uint32_t n_streams, int format, int fma_mode)
{
    return guard([&]() -> void * {
        std::unique_ptr<AudioBatchHandle> h(new AudioBatchHandle);
        h->batch.reset(new AudioBatch(static_cast<Device *>(device), n_streams, (AudioFormat)format, fma_mode));
        return h.release();
    }, (void *)nullptr);
}
void *mpeghost_audio_batch_open_store(void *store, uint32_t n_streams, int format, int fma_mode)
{
    return guard([&]() -> void * {
        std::unique_ptr<AudioBatchHandle> h(new AudioBatchHandle);
        h->batch.reset(new AudioBatch(std::unique_ptr<AudioBatchStore>(static_cast<AudioBatchStore *>(store)), n_streams,
                                      (AudioFormat)format, fma_mode));
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_audio_batch_close(void *h) { delete static_cast<AudioBatchHandle *>(h); }
int mpeghost_audio_batch_add_stream(void *hv, const uint8_t *data, size_t len)
{
    return guard([&]() -> int {
        AudioBatchHandle *h = static_cast<AudioBatchHandle *>(hv);
        h->bufs.push_back(Buffer::FromMemory(data, len));
        h->batch->AddStream(h->bufs.back().get());
        return (int)h->batch->Streams() - 1;
    }, -1);
}
int mpeghost_audio_batch_decode_all(void *hv)
{
    return guard([&]() -> int {
        AudioBatchHandle *h = static_cast<AudioBatchHandle *>(hv);
        return (int)h->batch->DecodeAll(h->samples);
    }, -1);
}
// AudioBatch::Stream(i)->Decode(): one frame of ONE stream, outside the batch's tick (its synthesis rides with the next device
// call: the samples are there after the next decode_all).  1 = a frame, 0 = the stream has ended, -1 error
int mpeghost_audio_batch_decode_stream(void *hv, uint32_t stream)
{
    return guard([&]() -> int {
        AudioBatchHandle *h = static_cast<AudioBatchHandle *>(hv);
        if (stream >= h->batch->Streams())
            throw std::runtime_error("mpeghost_audio_batch_decode_stream: no such stream");
        return h->batch->Stream(stream)->Decode() ? 1 : 0;
    }, -1);
}
// samples of stream i from the last decode_all (Interleaved / Left / F32 / S16 by format; right = Right for F32NLR)
const void *mpeghost_audio_batch_samples(void *hv, uint32_t stream, double *time, const void **right)
{
    AudioBatchHandle *h = static_cast<AudioBatchHandle *>(hv);
    if (stream >= h->samples.size() || !h->samples[stream])
        return nullptr;
    Samples *s = h->samples[stream];
    if (time)
        *time = s->Time;
    if (right)
        *right = s->Right.data();
    switch (s->format) {
    case AudioF32N: return s->Interleaved.data();
    case AudioF32: return s->F32.data();
    case AudioS16: return s->S16.data();
    default: return s->Left.data();
    }
}
uint64_t mpeghost_audio_batch_device_calls(void *hv) { return static_cast<AudioBatchHandle *>(hv)->batch->DeviceCalls(); }
// host threads of the parse (AudioBatch::SetThreads)
void mpeghost_audio_batch_set_threads(void *hv, uint32_t n)
{
    guard([&]() -> int {
        static_cast<AudioBatchHandle *>(hv)->batch->SetThreads(n);
        return 0;
    }, -1);
}

// NewDemux over a complete program stream (mpeg_test.go:88-100)
void *mpeghost_demux_open(const uint8_t *data, size_t len)
{
    return guard([&]() -> void * {
        std::unique_ptr<DemuxHandle> h(new DemuxHandle);
        h->buf = Buffer::FromMemory(data, len);
        h->demux.reset(new Demux(h->buf.get()));
        if (!h->demux->HasHeaders())
            throw std::runtime_error("invalid MPEG-PS header"); // ErrInvalidHeader
        return h.release();
    }, (void *)nullptr);
}
void mpeghost_demux_close(void *h) { delete static_cast<DemuxHandle *>(h); }
double mpeghost_demux_start_time(void *h, int type) { return static_cast<DemuxHandle *>(h)->demux->StartTime(type); }
double mpeghost_demux_duration(void *h, int type) { return static_cast<DemuxHandle *>(h)->demux->Duration(type); }
int mpeghost_demux_probe(void *h, size_t probe_size) { return static_cast<DemuxHandle *>(h)->demux->Probe(probe_size) ? 1 : 0; }
void mpeghost_demux_streams(void *h, int out[2])
{
    out[0] = static_cast<DemuxHandle *>(h)->demux->NumVideoStreams();
    out[1] = static_cast<DemuxHandle *>(h)->demux->NumAudioStreams();
}
void mpeghost_demux_rewind(void *h) { static_cast<DemuxHandle *>(h)->demux->Rewind(); }
// next packet: returns its type (0 at the end); pts / len / first payload bytes through the pointers
int mpeghost_demux_decode(void *h, double *pts, size_t *len, const uint8_t **data)
{
    Packet *p = static_cast<DemuxHandle *>(h)->demux->Decode();
    if (!p)
        return 0;
    *pts = p->Pts;
    *len = p->Len;
    *data = p->Data;
    return p->Type;
}
int mpeghost_demux_seek(void *h, double seconds, int type, int force_intra, double *pts, size_t *len, const uint8_t **data)
{
    Packet *p = static_cast<DemuxHandle *>(h)->demux->Seek(seconds, type, force_intra != 0);
    if (!p)
        return 0;
    *pts = p->Pts;
    *len = p->Len;
    *data = p->Data;
    return p->Type;
}

} // extern "C"

namespace {
void fill(mpeghost_frame_t *out, const Frame *f)
{
    out->time = f->Time;
    out->width = f->Width;
    out->height = f->Height;
    out->luma_w = f->Y.Width;
    out->luma_h = f->Y.Height;
    out->chroma_w = f->Cb.Width;
    out->chroma_h = f->Cb.Height;
    out->y = f->Y.Data;
    out->cb = f->Cb.Data;
    out->cr = f->Cr.Data;
    out->luma_bytes = f->Y.Len;
    out->chroma_bytes = f->Cb.Len;
}
} // namespace
