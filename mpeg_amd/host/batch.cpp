// batch.cpp — mpeg::VideoBatch: many streams, one reconstruction call per tick.
//
// Each stream keeps its own, unmodified parser (mpeg::Video); what it would submit to its own device
// store goes through a Port into the batch's descriptor arrays instead (stream index, macroblock and
// coefficient offsets rebased).  Pictures of ONE stream depend on each other, so a stream never has two
// pictures in the same device call: a second one (first reference picture of a stream, which yields
// no frame; the re-submit after a duplicated macroblock address in a damaged stream) flushes the batch
// first — launches are stream ordered, so "last writer in bitstream order" is kept.
#include <stdexcept>
#include <string.h>

#include "mpeg.hpp"

namespace mpeg {

class VideoBatch::Port : public VideoBackend {
public:
    Port(VideoBatch *b, uint32_t stream) : b_(b), stream_(stream) {}
    void open(int width, int height) override
    {
        if (b_->width_ == 0) {
            b_->width_ = width;
            b_->height_ = height;
            b_->store_->open(width, height, b_->capacity_);
        } else if (b_->width_ != width || b_->height_ != height) {
            throw std::runtime_error("VideoBatch: all streams must have the same picture size");
        }
    }
    void setQuant(const uint8_t intra[64], const uint8_t non_intra[64]) override
    {
        b_->Flush(); // the table belongs to pictures not yet queued
        b_->store_->setQuant(stream_, intra, non_intra);
    }
    void submit(const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
                size_t coef_bytes) override
    {
        b_->queue(stream_, pic, mbs, n_mbs, coefs, coef_bytes);
    }
    void readPlanes(uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) override
    {
        b_->Flush();
        b_->store_->readPlanes(stream_, slot, y, cb, cr);
    }
    void readRGBA(uint32_t slot, uint8_t *dst) override
    {
        b_->Flush();
        b_->store_->readRGBA(stream_, slot, dst);
    }

private:
    VideoBatch *b_;
    uint32_t stream_;
};

VideoBatch::VideoBatch(Device *dev, uint32_t n_streams) : VideoBatch(dev->newBatchStore(), n_streams) {}

VideoBatch::VideoBatch(std::unique_ptr<BatchStore> store, uint32_t n_streams) : store_(std::move(store)), capacity_(n_streams)
{
    if (n_streams == 0)
        throw std::runtime_error("VideoBatch: n_streams is 0");
    pending_.assign(n_streams, 0);
}

VideoBatch::~VideoBatch() {}

Video *VideoBatch::AddStream(Buffer *buf)
{
    if (videos_.size() >= capacity_)
        throw std::runtime_error("VideoBatch: more streams than the batch was opened for");
    const uint32_t idx = (uint32_t)videos_.size();
    videos_.emplace_back(new Video(buf, std::unique_ptr<VideoBackend>(new Port(this, idx))));
    return videos_.back().get();
}

void VideoBatch::queue(uint32_t stream, const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                       const uint8_t *coefs, size_t coef_bytes)
{
    if (pending_[stream])
        Flush(); // two pictures of one stream never share a device call
    const uint32_t pic_index = (uint32_t)pics_.size(), mb_first = (uint32_t)mbs_.size();
    const uint32_t unit0 = (uint32_t)(coefs_.size() / MPEGHIP_COEF_UNIT);
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
    store_->submit(pics_.data(), (uint32_t)pics_.size(), mbs_.data(), (uint32_t)mbs_.size(), coefs_.data(), coefs_.size());
    device_submits_++;
    pics_.clear();
    mbs_.clear();
    coefs_.clear();
    std::fill(pending_.begin(), pending_.end(), 0);
}

size_t VideoBatch::DecodeAll(std::vector<Frame *> &frames, bool fetch)
{
    const size_t n = videos_.size();
    frames.assign(n, nullptr);
    std::vector<uint32_t> slot(n, 0);
    std::vector<double> time(n, 0.0);
    std::vector<uint8_t> got(n, 0);
    // Rounds of "every stream that still owes a frame parses ONE picture, then one device call": a tick
    // costs as many calls as the neediest stream has pictures in it (1; 2 at a stream's start), not one
    // per stream.
    std::vector<uint32_t> todo(n);
    for (size_t i = 0; i < n; i++)
        todo[i] = (uint32_t)i;
    while (!todo.empty()) {
        std::vector<uint32_t> again;
        for (uint32_t i : todo) {
            const int r = videos_[i]->DecodeStep(&slot[i], &time[i]);
            if (r == 1)
                got[i] = 1;
            else if (r == 2)
                again.push_back(i);
        }
        Flush();
        todo.swap(again);
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
        if (b_->active_[stream_])
            b_->Flush(); // two frames of one stream never share a device call
        memcpy(b_->in_.data() + (size_t)stream_ * MPEGHIP_AUDIO_FRAME_INTS, samples, MPEGHIP_AUDIO_FRAME_INTS * sizeof(int32_t));
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
    in_.assign((size_t)n_streams * MPEGHIP_AUDIO_FRAME_INTS, 0);
    out_.assign((size_t)n_streams * 2304 * elemSize(format), 0);
    active_.assign(n_streams, 0);
    dest_.assign(n_streams, Dest());
}

AudioBatch::~AudioBatch() {}

Audio *AudioBatch::AddStream(Buffer *buf)
{
    if (audios_.size() >= capacity_)
        throw std::runtime_error("AudioBatch: more streams than the batch was opened for");
    const uint32_t idx = (uint32_t)audios_.size();
    audios_.emplace_back(new Audio(buf, std::unique_ptr<AudioBackend>(new Port(this, idx))));
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
    store_->synth(in_.data(), active_.data(), abiFormat(format_), out_.data());
    device_calls_++;
    const size_t es = elemSize(format_);
    for (uint32_t i = 0; i < capacity_; i++) {
        if (!active_[i])
            continue;
        const uint8_t *src = out_.data() + (size_t)i * 2304 * es;
        if (format_ == AudioF32NLR) {
            memcpy(dest_[i].out, src, 1152 * es);
            memcpy(dest_[i].out2, src + 1152 * es, 1152 * es);
        } else {
            memcpy(dest_[i].out, src, 2304 * es);
        }
        active_[i] = 0;
    }
}

size_t AudioBatch::DecodeAll(std::vector<Samples *> &samples)
{
    const size_t n = audios_.size();
    samples.assign(n, nullptr);
    size_t produced = 0;
    for (size_t i = 0; i < n; i++) { // CPU: parse, record
        samples[i] = audios_[i]->Decode();
        produced += samples[i] ? 1 : 0;
    }
    Flush();                         // GPU: one call for all streams
    return produced;
}

} // namespace mpeg
