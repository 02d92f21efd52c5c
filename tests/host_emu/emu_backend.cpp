// emu_backend.cpp — TEST INFRASTRUCTURE ONLY.  mpeg::VideoBackend / AudioBackend
// implemented with the lane emulator (tests/kernel_emu), so that the product's
// bitstream parser + descriptor emitter (libmpeghost) can be checked against the
// reference's golden hashes on a machine without a GPU.  Never part of the product:
// libmpeghost itself only knows the HIP backend.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <stdexcept>
#include <string>

#include <vector>

#include "mpeg.hpp"

extern "C" {
int emu_video_run(uint8_t *, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const mpeghip_pic_desc *, uint32_t,
                  const mpeghip_mb_desc *, uint32_t, const uint8_t *, const uint8_t *, uint8_t *, uint64_t);
void emu_set_device_pack(int);
void emu_set_wide(int);
int emu_get_wide(void);
void emu_set_mirror(uint8_t *, uint64_t);
void emu_make_qtable(uint8_t *, const uint8_t *, const uint8_t *);
void emu_rgba_convert(const uint8_t *, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t *);
void emu_relayout(uint8_t *, uint8_t *, uint32_t, uint32_t, int);
int emu_audio_run(const int32_t *, void *, float *, int32_t *, const float *, uint32_t, uint32_t, int32_t, int32_t, uint32_t);
int emu_audio_run_masked(const int32_t *, void *, float *, int32_t *, const float *, uint32_t, uint32_t, int32_t, int32_t, uint32_t,
                         const uint8_t *);
}

namespace {

class EmuVideoBackend : public mpeg::VideoBackend {
public:
    explicit EmuVideoBackend(int flavour) : flavour_(flavour) {}
    void open(int width, int height) override
    {
        w_ = width;
        h_ = height;
        lw_ = ((width + 15) >> 4) << 4;
        lh_ = ((height + 15) >> 4) << 4;
        luma_ = (size_t)lw_ * lh_;
        chroma_ = luma_ / 4;
        stride_ = (luma_ + 2 * chroma_ + (size_t)lw_ * 16 + 64 + 255) / 256 * 256;
        frames_.assign(stride_ * 3, 0);
        rgba_stride_ = ((size_t)width * height * 4 + 255) / 256 * 256;
        rgba_.assign(rgba_stride_ * 3, 0);
        dump_.assign(512, 0);
        mirror_stride_ = (luma_ + 2 * chroma_ + 255) / 256 * 256;
        mirror_.assign(mirror_stride_ * 3, 0); // (the frame store starts out zeroed: the copies are right from the start)
    }
    void setQuant(const uint8_t intra[64], const uint8_t non_intra[64]) override { emu_make_qtable(qt_, intra, non_intra); }
    void submit(const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs, size_t) override
    {
        mpeghip_pic_desc p = pic; // (the emulator packs each picture's own range)
        p.mb_first = 0;
        p.mb_count = n_mbs;
        const int wide_before = emu_get_wide();
        if (mirrored_) { // the host mirror: every chunk as recon_wide_kernel<false, true> runs it (a lone decoder's launches are small)
            emu_set_wide(1);
            emu_set_mirror(mirror_.data(), mirror_stride_);
        }
        emu_video_run(frames_.data(), stride_, lw_, lh_, w_, h_, &p, 1, mbs, n_mbs, coefs, qt_, rgba_.data(), rgba_stride_);
        if (mirrored_) {
            emu_set_wide(wide_before);
            emu_set_mirror(nullptr, 0);
        }
    }
    // (mpeghip_video_host_mirror's stand-in: the emulated launches keep the three linear copies; switching it on later untiles them)
    void setMirror(bool on) override
    {
        if (on && !mirrored_ && !frames_.empty())
            for (uint32_t slot = 0; slot < 3; slot++)
                emu_relayout(frames_.data() + slot * stride_, mirror_.data() + slot * mirror_stride_, lw_, lh_, 1);
        mirrored_ = on;
    }
    const uint8_t *mirrorAsync(uint32_t slot, uint64_t *ticket) override
    {
        *ticket = 0;
        return mirrored_ ? mirror_.data() + slot * mirror_stride_ : nullptr;
    }
    void readPlanes(uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        std::vector<uint8_t> lin(luma_ + 2 * chroma_); // the frame store is tiled: untile, as mpeghip_video_read_planes does
        emu_relayout(frames_.data() + slot * stride_, lin.data(), lw_, lh_, 1);
        memcpy(y, lin.data(), luma_);
        memcpy(cb, lin.data() + luma_, chroma_);
        memcpy(cr, lin.data() + luma_ + chroma_, chroma_);
    }
    void readRGBA(uint32_t slot, uint8_t *dst) override
    {
        emu_rgba_convert(frames_.data() + slot * stride_, lw_, lh_, w_, h_, dst);
    }

private:
    int flavour_;
    uint32_t w_ = 0, h_ = 0, lw_ = 0, lh_ = 0;
    size_t luma_ = 0, chroma_ = 0, stride_ = 0, rgba_stride_ = 0, mirror_stride_ = 0;
    std::vector<uint8_t> frames_, rgba_, dump_, mirror_;
    bool mirrored_ = false; // (off unless a test asks: Video::SetHostMirror(true) -> setMirror)
    alignas(16) uint8_t qt_[256 + 1024] = {}; // (+ the padding surplus lanes of the table load may read, as on the device)
};

// multi-stream store over the wave-chunk lane emulator (what HipBatchStore is over libmpeghip)
class EmuBatchStore : public mpeg::BatchStore {
public:
    void open(int width, int height, uint32_t n_streams) override
    {
        w_ = width;
        h_ = height;
        n_ = n_streams;
        lw_ = ((width + 15) >> 4) << 4;
        lh_ = ((height + 15) >> 4) << 4;
        luma_ = (size_t)lw_ * lh_;
        chroma_ = luma_ / 4;
        stride_ = (luma_ + 2 * chroma_ + (size_t)lw_ * 16 + 64 + 255) / 256 * 256;
        frames_.assign(stride_ * 3 * n_streams + 4096, 0);
        rgba_stride_ = ((size_t)w_ * h_ * 4 + 255) / 256 * 256;
        rgba_.assign(rgba_stride_ * 3 * n_streams, 0);
        qt_store_.assign((size_t)256 * n_streams + 16 + 1024, 0);
        qt_ = qt_store_.data() + (16 - reinterpret_cast<uintptr_t>(qt_store_.data()) % 16) % 16; // 16-byte aligned, as on the device
    }
    void setQuant(uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64]) override
    {
        emu_make_qtable(qt_ + (size_t)stream * 256, intra, non_intra);
    }
    void submit(const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
                size_t) override
    {
        emu_video_run(frames_.data(), stride_, lw_, lh_, w_, h_, pics, n_pics, mbs, n_mbs, coefs, qt_, rgba_.data(),
                         rgba_stride_);
    }
    // staged submit (the product's mpeghip_video_stage_*): pictures put from several threads into one merged submit
    bool canStage() const override { return true; }
    void stageBegin(const std::vector<uint32_t> &n_mbs, const std::vector<size_t> &coef_bytes, bool device_pack) override
    {
        device_pack_stages_ += device_pack ? 1 : 0;
        st_device_pack_ = device_pack;
        const size_t n = n_mbs.size();
        st_first_.assign(n, 0);
        st_unit_.assign(n, 0);
        st_count_ = n_mbs;
        st_bytes_ = coef_bytes;
        size_t mbs = 0, bytes = 0;
        for (size_t i = 0; i < n; i++) { // (every picture starts on a unit boundary: either form may follow)
            st_first_[i] = (uint32_t)mbs;
            st_unit_[i] = (uint32_t)(bytes / MPEGHIP_COEF_UNIT);
            mbs += n_mbs[i];
            bytes += (coef_bytes[i] + MPEGHIP_COEF_UNIT - 1) / MPEGHIP_COEF_UNIT * MPEGHIP_COEF_UNIT;
        }
        st_pics_.assign(n, mpeghip_pic_desc{});
        st_mbs_.assign(mbs, mpeghip_mb_desc{});
        st_coefs_.assign(bytes, 0);
        stage_puts_ = 0;
    }
    void stagePut(uint32_t i, const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, const uint8_t *coefs) override
    {
        mpeghip_pic_desc p = pic;
        p.mb_first = st_first_[i];
        p.mb_count = st_count_[i];
        st_pics_[i] = p;
        for (uint32_t k = 0; k < st_count_[i]; k++) {
            mpeghip_mb_desc m = mbs[k];
            m.pic = i;
            m.coef_off += (pic.flags & MPEGHIP_PIC_SPARSE) ? st_unit_[i] * (MPEGHIP_COEF_UNIT / 4) : st_unit_[i]; // dwords / units
            st_mbs_[st_first_[i] + k] = m;
        }
        if (st_bytes_[i])
            memcpy(st_coefs_.data() + (size_t)st_unit_[i] * MPEGHIP_COEF_UNIT, coefs, st_bytes_[i]);
        stage_puts_++;
    }
    void stageCommit() override
    {
        if (stage_puts_.load() != st_pics_.size())
            abort();
        // a device-packed stage: the pictures go through the DEVICE packer's lane functions (video_pack_lane.h), wave by wave —
        // picture by picture here, because a refusal is the picture's own (pack_gate_kernel): the others are reconstructed, the
        // refused one's stream is remembered for verdict() / sync(), which report it once
        if (st_device_pack_) {
            emu_set_device_pack(1);
            for (size_t i = 0; i < st_pics_.size(); i++) {
                const int rc = emu_video_run(frames_.data(), stride_, lw_, lh_, w_, h_, &st_pics_[i], 1, st_mbs_.data(), (uint32_t)st_mbs_.size(),
                                             st_coefs_.data(), qt_, rgba_.data(), rgba_stride_);
                if (rc != 0)
                    pending_refused_.push_back(st_pics_[i].stream);
            }
            emu_set_device_pack(0);
        } else {
            submit(st_pics_.data(), (uint32_t)st_pics_.size(), st_mbs_.data(), (uint32_t)st_mbs_.size(), st_coefs_.data(),
                   st_coefs_.size());
        }
        staged_commits_++;
    }
    void sync() override { verdict(); }
    void verdict() override
    {
        if (pending_refused_.empty())
            return;
        refused_ = pending_refused_;
        pending_refused_.clear();
        throw std::runtime_error("device-packed commit: " + std::to_string(refused_.size()) + " picture(s) refused and not reconstructed "
                                 "(the others were): stream " + std::to_string(refused_[0]));
    }
    std::vector<uint32_t> refusedStreams() override { return refused_; }
    void readPlanes(uint32_t stream, uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        std::vector<uint8_t> lin(luma_ + 2 * chroma_);
        emu_relayout(frames_.data() + ((size_t)stream * 3 + slot) * stride_, lin.data(), lw_, lh_, 1);
        memcpy(y, lin.data(), luma_);
        memcpy(cb, lin.data() + luma_, chroma_);
        memcpy(cr, lin.data() + luma_ + chroma_, chroma_);
    }
    void readRGBA(uint32_t stream, uint32_t slot, uint8_t *dst) override
    {
        emu_rgba_convert(frames_.data() + ((size_t)stream * 3 + slot) * stride_, lw_, lh_, w_, h_, dst);
    }

private:
    uint32_t w_ = 0, h_ = 0, lw_ = 0, lh_ = 0, n_ = 0;
    size_t luma_ = 0, chroma_ = 0, stride_ = 0, rgba_stride_ = 0;
    std::vector<uint8_t> frames_, rgba_, qt_store_;
    uint8_t *qt_ = nullptr;
    std::vector<uint32_t> st_first_, st_unit_, st_count_;
    std::vector<size_t> st_bytes_;
    std::vector<mpeghip_pic_desc> st_pics_;
    std::vector<mpeghip_mb_desc> st_mbs_;
    std::vector<uint8_t> st_coefs_;
    std::atomic<size_t> stage_puts_{0};

    bool st_device_pack_ = false;
    std::vector<uint32_t> pending_refused_, refused_;

public:
    uint64_t staged_commits_ = 0, device_pack_stages_ = 0;
};

class EmuAudioBackend : public mpeg::AudioBackend {
public:
    EmuAudioBackend(int fma, const float *window) : fma_(fma)
    {
        memset(ring_, 0, sizeof(ring_));
        memcpy(window_, window, sizeof(window_));
    }
    void synth(const int32_t *samples, int format, void *out, void *out2) override
    {
        if (format == MPEGHIP_AUDIO_F32NLR) {
            float lr[2304];
            emu_audio_run(samples, lr, &ring_[0][0], &vpos_, window_, 1, 1, format, fma_, 1);
            memcpy(out, lr, 1152 * sizeof(float));
            memcpy(out2, lr + 1152, 1152 * sizeof(float));
        } else {
            emu_audio_run(samples, out, &ring_[0][0], &vpos_, window_, 1, 1, format, fma_, 1);
        }
    }

private:
    int fma_;
    float ring_[2][1024];
    int32_t vpos_ = 0;
    float window_[512];
};

// multi-stream synthesis over the lane emulator (what HipAudioBatchStore is over libmpeghip)
class EmuAudioBatchStore : public mpeg::AudioBatchStore {
public:
    explicit EmuAudioBatchStore(const float *window) { memcpy(window_, window, sizeof(window_)); }
    void open(uint32_t n_streams, int fma_mode) override
    {
        n_ = n_streams;
        fma_ = fma_mode;
        ring_.assign((size_t)n_streams * 2048, 0.0f);
        vpos_.assign(n_streams, 0);
    }
    void synth(const int32_t *samples, const uint8_t *active, int format, void *out) override
    {
        emu_audio_run_masked(samples, out, ring_.data(), vpos_.data(), window_, n_, 1, format, fma_, 1, active);
    }

private:
    uint32_t n_ = 0;
    int fma_ = 0;
    std::vector<float> ring_;
    std::vector<int32_t> vpos_;
    float window_[512];
};

} // namespace

static int g_flavour = 0;
static float g_window[512];

extern "C" {
void *host_emu_video_backend(int flavour) { return new EmuVideoBackend(flavour); }
void *host_emu_audio_backend(int fma, const float *window512) { return new EmuAudioBackend(fma, window512); }
// factories with the signature mpeghost_mpeg_open_backends wants
void host_emu_configure(int flavour, const float *window512)
{
    g_flavour = flavour;
    memcpy(g_window, window512, sizeof(g_window));
}
void *host_emu_make_video(void) { return new EmuVideoBackend(g_flavour); }
// A store that swallows everything: what is left of a decode is the parser's own time (tools/bench_parse.py)
class NullBatchStore : public mpeg::BatchStore {
public:
    void open(int, int, uint32_t) override {}
    void setQuant(uint32_t, const uint8_t[64], const uint8_t[64]) override {}
    void submit(const mpeghip_pic_desc *, uint32_t, const mpeghip_mb_desc *, uint32_t, const uint8_t *, size_t) override {}
    void readPlanes(uint32_t, uint32_t, uint8_t *, uint8_t *, uint8_t *) override {}
    void readRGBA(uint32_t, uint32_t, uint8_t *) override {}
};
void *host_emu_null_batch_store(void) { return new NullBatchStore(); }
void *host_emu_batch_store(void) { return new EmuBatchStore(); }
// staged submits the store has seen (valid while the batch that owns the store is open)
uint64_t host_emu_batch_store_device_pack_stages(void *store) { return static_cast<EmuBatchStore *>(static_cast<mpeg::BatchStore *>(store))->device_pack_stages_; }
uint64_t host_emu_batch_store_staged_commits(void *store) { return static_cast<EmuBatchStore *>(static_cast<mpeg::BatchStore *>(store))->staged_commits_; }
void *host_emu_audio_batch_store(void) { return new EmuAudioBatchStore(g_window); }
void *host_emu_make_audio(int fma) { return new EmuAudioBackend(fma, g_window); }
}
