// hip_backend.cpp — the one and only reconstruction backend of the product: libmpeghip.
#include <stdexcept>
#include <string.h>
#include <string>

#include "mpeg.hpp"

namespace mpeg {

namespace {

void check(int rc, const char *what)
{
    if (rc != MPEGHIP_OK)
        throw std::runtime_error(std::string(what) + ": " + mpeghip_last_error());
}

// From how many macroblocks on a lone decoder's hand-over is packed by the device.  Measured on the GPU box with written streams
// (tools/single_stream_phases.py, profiles/round6_f_lone_decoder_phases.txt): the host packs at 50 ns per macroblock and nothing
// else (16 us of a SIF picture's hand-over, 0.42 ms of a 1080p picture's), a device-packed stage of one picture costs 90 us of
// calls, copy and waiting whatever its size plus 15 ns per macroblock (1080p: 0.12 ms; SIF: 50 us + 40 us more wait) — they cross
// near 2 500 macroblocks.
constexpr uint32_t kDevicePackFromDefault = 3000;

class HipVideoBackend : public VideoBackend {
public:
    explicit HipVideoBackend(mpeghip_ctx *ctx) : ctx_(ctx) {}
    ~HipVideoBackend() override
    {
        if (store_)
            mpeghip_video_close(store_);
    }
    void open(int width, int height) override
    {
        if (store_)
            mpeghip_video_close(store_);
        store_ = nullptr;
        check(mpeghip_video_open(ctx_, (uint32_t)width, (uint32_t)height, 1, &store_), "mpeghip_video_open");
        mirrored_ = false;
        setMirror(mirror_wanted_);
    }
    // a lone decoder's three frames once more in pinned host memory, written by the reconstruction launches (mpeghip_video_host_mirror);
    // without it (no pinned memory to be had) Decode reads back as before
    void setMirror(bool on) override
    {
        mirror_wanted_ = on;
        if (store_ && on != mirrored_ && mpeghip_video_host_mirror(store_, on ? 1 : 0) == MPEGHIP_OK)
            mirrored_ = on;
    }
    const uint8_t *mirrorAsync(uint32_t slot, uint64_t *ticket) override
    {
        if (!mirrored_)
            return nullptr;
        const uint8_t *planes = nullptr;
        check(mpeghip_video_mirror_async(store_, 0, slot, &planes, ticket), "mpeghip_video_mirror_async");
        return planes;
    }
    void setQuant(const uint8_t intra[64], const uint8_t non_intra[64]) override
    {
        check(mpeghip_video_set_quant(store_, 0, intra, non_intra), "mpeghip_video_set_quant");
    }
    void submit(const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
                size_t coef_bytes) override
    {
        // A large picture in the parser's sparse form goes through a DEVICE-PACKED stage of one picture: validating and packing it
        // costs this thread — the one that parses — 50 ns per macroblock (0.4 ms of a 1080p picture's 1.4), the device does it
        // beside the next picture's parse.  Its verdict is deferred (include/mpeghip.h) to the next call that waits for the device:
        // the frame's readWait — the parser hands over nothing the device refuses (video.cpp: emitPrediction drops what would).
        if (device_pack_from_ && n_mbs >= device_pack_from_ && (pic.flags & MPEGHIP_PIC_SPARSE)) {
            const size_t n_words = coef_bytes / 4;
            mpeghip_stage *stage = nullptr;
            check(mpeghip_video_stage_begin_device(store_, 1, &n_mbs, &n_words, &stage), "mpeghip_video_stage_begin_device");
            mpeghip_pic_desc p = pic;
            p.mb_first = 0;
            p.mb_count = n_mbs;
            const int put = mpeghip_video_stage_put_sparse(stage, 0, &p, mbs, reinterpret_cast<const uint32_t *>(coefs));
            const std::string why = put != MPEGHIP_OK ? mpeghip_last_error() : "";
            const int commit = mpeghip_video_stage_commit(stage); // (ends the stage whatever the put said)
            if (put != MPEGHIP_OK)
                throw std::runtime_error("mpeghip_video_stage_put_sparse: " + why);
            check(commit, "mpeghip_video_stage_commit");
            return;
        }
        check(mpeghip_video_submit(store_, &pic, 1, mbs, n_mbs, coefs, coef_bytes), "mpeghip_video_submit");
    }
    void setDevicePackFrom(uint32_t n_mbs) override { device_pack_from_ = n_mbs; }
    void readPlanes(uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        check(mpeghip_video_read_planes(store_, 0, slot, y, cb, cr), "mpeghip_video_read_planes");
    }
    void readRGBA(uint32_t slot, uint8_t *dst) override
    {
        check(mpeghip_video_rgba_convert(store_, slot, 0, 1), "mpeghip_video_rgba_convert");
        check(mpeghip_video_read_rgba(store_, 0, slot, dst), "mpeghip_video_read_rgba");
    }
    // Video::Decode's frames: pinned memory, filled by the device itself (mpeghip_video_read_planes_async), waited for by ticket
    uint8_t *allocPlanes(size_t bytes) override
    {
        uint8_t *p = static_cast<uint8_t *>(mpeghip_pinned_alloc(ctx_, bytes ? bytes : 1));
        if (p)
            memset(p, 0, bytes);
        return p;
    }
    void freePlanes(uint8_t *p) override { mpeghip_pinned_free(ctx_, p); }
    uint64_t readPlanesAsync(uint32_t slot, uint8_t *dst, size_t, size_t) override
    {
        uint64_t ticket = 0;
        check(mpeghip_video_read_planes_async(store_, 0, slot, dst, &ticket), "mpeghip_video_read_planes_async");
        return ticket;
    }
    void readWait(uint64_t ticket) override { check(mpeghip_video_read_wait(store_, ticket), "mpeghip_video_read_wait"); }

private:
    mpeghip_ctx *ctx_;
    mpeghip_video *store_ = nullptr;
    bool mirror_wanted_ = true, mirrored_ = false;
    uint32_t device_pack_from_ = kDevicePackFromDefault; // macroblocks per hand-over from which the device packs (0: never)
};

class HipBatchStore : public BatchStore {
public:
    explicit HipBatchStore(mpeghip_ctx *ctx) : ctx_(ctx) {}
    ~HipBatchStore() override
    {
        if (store_)
            mpeghip_video_close(store_);
    }
    void open(int width, int height, uint32_t n_streams) override
    {
        if (store_) { // (re-open: a device-packed commit still in flight on the old handle reports before the handle goes)
            const int verdict = mpeghip_video_sync(store_);
            const std::string why = verdict != MPEGHIP_OK ? mpeghip_last_error() : "";
            mpeghip_video_close(store_);
            store_ = nullptr;
            if (verdict != MPEGHIP_OK)
                throw std::runtime_error("mpeghip_video_sync (before the store was re-opened): " + why);
        }
        check(mpeghip_video_open(ctx_, (uint32_t)width, (uint32_t)height, n_streams, &store_), "mpeghip_video_open");
    }
    void setQuant(uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64]) override
    {
        check(mpeghip_video_set_quant(store_, stream, intra, non_intra), "mpeghip_video_set_quant");
    }
    void submit(const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
                size_t coef_bytes) override
    {
        check(mpeghip_video_submit(store_, pics, n_pics, mbs, n_mbs, coefs, coef_bytes), "mpeghip_video_submit");
    }
    void readPlanes(uint32_t stream, uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        check(mpeghip_video_read_planes(store_, stream, slot, y, cb, cr), "mpeghip_video_read_planes");
    }
    void readRGBA(uint32_t stream, uint32_t slot, uint8_t *dst) override
    {
        check(mpeghip_video_rgba_convert(store_, slot, stream, 1), "mpeghip_video_rgba_convert");
        check(mpeghip_video_read_rgba(store_, stream, slot, dst), "mpeghip_video_read_rgba");
    }
    bool canStage() const override { return true; }
    void stageBegin(const std::vector<uint32_t> &n_mbs, const std::vector<size_t> &coef_bytes, bool device_pack) override
    {
        if (device_pack) { // sparse pictures all: the arrays travel as they are, the device validates and packs them
            std::vector<size_t> n_words(coef_bytes.size());
            for (size_t i = 0; i < coef_bytes.size(); i++)
                n_words[i] = coef_bytes[i] / 4;
            check(mpeghip_video_stage_begin_device(store_, (uint32_t)n_mbs.size(), n_mbs.data(), n_words.data(), &stage_),
                  "mpeghip_video_stage_begin_device");
            return;
        }
        check(mpeghip_video_stage_begin(store_, (uint32_t)n_mbs.size(), n_mbs.data(), coef_bytes.data(), &stage_),
              "mpeghip_video_stage_begin");
    }
    void stagePut(uint32_t i, const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, const uint8_t *coefs) override
    {
        check(mpeghip_video_stage_put(stage_, i, &pic, mbs, coefs), "mpeghip_video_stage_put");
    }
    void stageCommit() override
    {
        mpeghip_stage *s = stage_;
        stage_ = nullptr; // the commit ends the stage whatever it returns
        check(mpeghip_video_stage_commit(s), "mpeghip_video_stage_commit");
    }
    void sync() override { check(mpeghip_video_sync(store_), "mpeghip_video_sync"); }
    void verdict() override { check(mpeghip_video_verdict(store_), "mpeghip_video_verdict"); }
    std::vector<uint32_t> refusedStreams() override
    {
        std::vector<uint32_t> streams(1024);
        const uint64_t n = mpeghip_video_refused(store_, nullptr, streams.data(), (uint32_t)streams.size());
        streams.resize(n < streams.size() ? (size_t)n : streams.size());
        return streams;
    }

private:
    mpeghip_ctx *ctx_;
    mpeghip_video *store_ = nullptr;
    mpeghip_stage *stage_ = nullptr;
};

class HipAudioBackend : public AudioBackend {
public:
    HipAudioBackend(mpeghip_ctx *ctx, int fma_mode) : ctx_(ctx)
    {
        check(mpeghip_audio_open(ctx, 1, fma_mode, &synth_), "mpeghip_audio_open");
        for (auto &o : out_) {
            o = static_cast<uint8_t *>(mpeghip_pinned_alloc(ctx_, 2304 * sizeof(float)));
            if (!o)
                throw std::runtime_error(std::string("mpeghip_pinned_alloc: ") + mpeghip_last_error());
        }
    }
    ~HipAudioBackend() override
    {
        mpeghip_audio_close(synth_);
        for (uint8_t *o : out_)
            mpeghip_pinned_free(ctx_, o);
    }
    // Audio::Decode's frames: sub-band samples in pinned memory, read by the kernel in place; its output lands in one of two
    // pinned buffers here and is copied to the caller's Samples at the wait (9 KB)
    int32_t *allocSamples() override
    {
        int32_t *p = static_cast<int32_t *>(mpeghip_pinned_alloc(ctx_, MPEGHIP_AUDIO_FRAME_INTS * sizeof(int32_t)));
        if (p)
            memset(p, 0, MPEGHIP_AUDIO_FRAME_INTS * sizeof(int32_t));
        return p;
    }
    void freeSamples(int32_t *p) override { mpeghip_pinned_free(ctx_, p); }
    uint64_t synthAsync(const int32_t *samples, int format) override
    {
        uint64_t ticket = 0;
        check(mpeghip_audio_synth_async(synth_, samples, 1, format, out_[queued_ & 1], &ticket), "mpeghip_audio_synth_async");
        format_[queued_ & 1] = format;
        ticket_[queued_ & 1] = ticket;
        return queued_++;
    }
    void synthWait(uint64_t n, void *out, void *out2) override
    {
        check(mpeghip_audio_synth_wait(synth_, ticket_[n & 1]), "mpeghip_audio_synth_wait");
        const uint8_t *o = out_[n & 1];
        if (format_[n & 1] == MPEGHIP_AUDIO_F32NLR) {
            memcpy(out, o, 1152 * sizeof(float));
            memcpy(out2, o + 1152 * sizeof(float), 1152 * sizeof(float));
        } else {
            memcpy(out, o, 2304 * (format_[n & 1] == MPEGHIP_AUDIO_S16 ? sizeof(int16_t) : sizeof(float)));
        }
    }
    void synth(const int32_t *samples, int format, void *out, void *out2) override
    {
        if (format == MPEGHIP_AUDIO_F32NLR) {
            float lr[2304];
            check(mpeghip_audio_synth(synth_, samples, 1, format, lr), "mpeghip_audio_synth");
            memcpy(out, lr, 1152 * sizeof(float));
            memcpy(out2, lr + 1152, 1152 * sizeof(float));
        } else {
            check(mpeghip_audio_synth(synth_, samples, 1, format, out), "mpeghip_audio_synth");
        }
    }

private:
    mpeghip_ctx *ctx_;
    mpeghip_audio *synth_ = nullptr;
    uint8_t *out_[2] = {nullptr, nullptr};
    int format_[2] = {0, 0};
    uint64_t ticket_[2] = {0, 0}, queued_ = 0;
};

class HipAudioBatchStore : public AudioBatchStore {
public:
    explicit HipAudioBatchStore(mpeghip_ctx *ctx) : ctx_(ctx) {}
    ~HipAudioBatchStore() override
    {
        if (synth_)
            mpeghip_audio_close(synth_);
    }
    void open(uint32_t n_streams, int fma_mode) override
    {
        check(mpeghip_audio_open(ctx_, n_streams, fma_mode, &synth_), "mpeghip_audio_open");
    }
    void synth(const int32_t *samples, const uint8_t *active, int format, void *out) override
    {
        check(mpeghip_audio_synth_masked(synth_, samples, 1, format, out, active), "mpeghip_audio_synth_masked");
    }
    void *allocHost(size_t bytes) override
    {
        void *p = mpeghip_pinned_alloc(ctx_, bytes ? bytes : 1);
        if (p)
            memset(p, 0, bytes);
        return p;
    }
    void freeHost(void *p) override { mpeghip_pinned_free(ctx_, p); }

private:
    mpeghip_ctx *ctx_;
    mpeghip_audio *synth_ = nullptr;
};

} // namespace

Device::Device(int ordinal)
{
    if (mpeghip_abi_version() != MPEGHIP_ABI_VERSION) // (the header this library was built against vs the libmpeghip.so it found)
        throw std::runtime_error("mpeg::Device: libmpeghip has ABI version " + std::to_string(mpeghip_abi_version()) + ", built for " +
                                 std::to_string(MPEGHIP_ABI_VERSION));
    if (mpeghip_ctx_create(ordinal, nullptr, &ctx_) != MPEGHIP_OK)
        throw std::runtime_error(std::string("mpeg::Device: ") + mpeghip_last_error());
}

Device::~Device() { mpeghip_ctx_destroy(ctx_); }
int Device::NumaNode() const { return mpeghip_ctx_numa_node(ctx_); }

std::unique_ptr<VideoBackend> Device::newVideoBackend() { return std::unique_ptr<VideoBackend>(new HipVideoBackend(ctx_)); }
std::unique_ptr<AudioBatchStore> Device::newAudioBatchStore()
{
    return std::unique_ptr<AudioBatchStore>(new HipAudioBatchStore(ctx_));
}
std::unique_ptr<BatchStore> Device::newBatchStore() { return std::unique_ptr<BatchStore>(new HipBatchStore(ctx_)); }
std::unique_ptr<AudioBackend> Device::newAudioBackend(int fma_mode)
{
    return std::unique_ptr<AudioBackend>(new HipAudioBackend(ctx_, fma_mode));
}

} // namespace mpeg
