// audio.cpp — mpeg::Audio: MPEG-1 Audio Layer II frame parse on the CPU
// (audio.go:163-490), sub-band synthesis (audio.go:378-422, 492-772) on the GPU.
#include <string.h>

#include "mpeg.hpp"

namespace mpeg {

namespace {

// ISO 11172-3 header / Layer II tables (audio.go:798-973)
const uint16_t kSamplerate[4] = {44100, 48000, 32000, 0};
const int16_t kBitrate[14] = {32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384};
const int kScalefactorBase[3] = {0x02000000, 0x01965FEA, 0x01428A30};
// step 1: [mono / stereo][bitrate index] -> bitrate class
const uint8_t kQuantLutStep1[2][14] = {{0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2}, {0, 0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2}};
// step 2: [class][sample rate] -> sblimit | (high-rate table ? 64 : 0)   (tables 3-B.2a..d)
const uint8_t kQuantLutStep2[3][3] = {{8, 8, 12}, {27 | 64, 27 | 64, 27 | 64}, {30 | 64, 27 | 64, 30 | 64}};
// step 3: [table][subband] -> nbal << 4 | row
const uint8_t kQuantLutStep3[2][32] = {
    {0x44, 0x44, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34, 0x34},
    {0x43, 0x43, 0x43, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x42, 0x31, 0x31, 0x31, 0x31, 0x31,
     0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x31, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20}};
// step 4: [row][allocation] -> quantiser class index (0 = no bits)
const uint8_t kQuantLutStep4[6][16] = {{0, 1, 2, 17},
                                       {0, 1, 2, 3, 4, 5, 6, 17},
                                       {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 17},
                                       {0, 1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17},
                                       {0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16},
                                       {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}};

constexpr int kFrameSync = 0x7ff, kMpeg1 = 0x3, kLayerII = 0x2;
constexpr int kModeStereo = 0, kModeJointStereo = 1, kModeMono = 3;

} // namespace

const Audio::QuantizerSpec Audio::quant_tab_[17] = {
    {3, 1, 5},     {5, 1, 7},     {7, 0, 3},      {9, 1, 10},     {15, 0, 4},     {31, 0, 5},
    {63, 0, 6},    {127, 0, 7},   {255, 0, 8},    {511, 0, 9},    {1023, 0, 10},  {2047, 0, 11},
    {4095, 0, 12}, {8191, 0, 13}, {16383, 0, 14}, {32767, 0, 15}, {65535, 0, 16}};

Audio::Audio(Buffer *buf, Device *dev, int fma_mode) : buf_(buf), backend_(dev->newAudioBackend(fma_mode)) { init(); }
Audio::Audio(Buffer *buf, std::unique_ptr<AudioBackend> backend) : buf_(buf), backend_(std::move(backend)) { init(); }
Audio::~Audio()
{
    for (int32_t *p : in_)
        if (p)
            backend_->freeSamples(p);
}

void Audio::init()
{ // audio.go:83-104
    samplerate_index_ = 3;
    for (int i = 0; i < 2; i++) {
        Samples &sm = samples_[i];
        sm.S16.assign(SamplesPerFrame * 2, 0);
        sm.F32.assign(SamplesPerFrame * 2, 0);
        sm.Left.assign(SamplesPerFrame, 0);
        sm.Right.assign(SamplesPerFrame, 0);
        sm.Interleaved.assign(SamplesPerFrame * 2, 0);
        in_[i] = backend_->allocSamples();
        if (!in_[i])
            throw std::bad_alloc();
    }
    next_frame_data_size_ = decodeHeader();
}

bool Audio::HasHeader()
{ // audio.go:112-120
    if (has_header_)
        return true;
    next_frame_data_size_ = decodeHeader();
    return has_header_;
}

int Audio::Samplerate() { return HasHeader() ? kSamplerate[samplerate_index_] : 0; }

void Audio::SetTime(double t)
{ // audio.go:143-146
    samples_decoded_ = (int)(t * (double)kSamplerate[samplerate_index_]);
    time_ = t;
    if (ahead_valid_) { // (a frame parsed ahead is the next one the reference would decode: it carries the new time)
        ahead_time_ = time_;
        samples_decoded_ += SamplesPerFrame;
        time_ = (double)samples_decoded_ / (double)kSamplerate[samplerate_index_];
    }
}

void Audio::Rewind()
{ // audio.go:149-154 — the V ring and vPos are NOT cleared, exactly like the reference
    ahead_tried_ = false;
    ahead_valid_ = ahead_failed_ = false; // a frame parsed ahead was never synthesised: the ring is what the frames RETURNED left behind
    buf_->Rewind();
    time_ = 0;
    samples_decoded_ = 0;
    next_frame_data_size_ = 0;
}

const uint8_t *Samples::Bytes(size_t *len) const
{ // audio.go:39-50
    switch (format) {
    case AudioF32N:
        *len = Interleaved.size() * 4;
        return reinterpret_cast<const uint8_t *>(Interleaved.data());
    case AudioF32:
        *len = F32.size() * 4;
        return reinterpret_cast<const uint8_t *>(F32.data());
    case AudioS16:
        *len = S16.size() * 2;
        return reinterpret_cast<const uint8_t *>(S16.data());
    default:
        *len = 0;
        return nullptr;
    }
}

bool Audio::parseNext(int *buf, double *time)
{ // audio.go:163-182, up to the synthesis
    if (next_frame_data_size_ == 0)
        next_frame_data_size_ = decodeHeader();
    if (next_frame_data_size_ == 0 || !buf_->has((size_t)next_frame_data_size_ << 3))
        return false;
    const int b = in_next_;
    in_next_ ^= 1;
    decodeFrame(reinterpret_cast<int32_t(*)[36][32]>(in_[b]));
    next_frame_data_size_ = 0;
    *buf = b;
    *time = time_;
    samples_decoded_ += SamplesPerFrame;
    time_ = (double)samples_decoded_ / (double)kSamplerate[samplerate_index_];
    return true;
}

Samples *Audio::Decode()
{ // audio.go:163-182, one frame ahead on the host (mpeg.hpp)
    int b;
    double t;
    ahead_tried_ = false;
    if (ahead_failed_) { // the attempt this call stands for has been made (and has consumed what it consumed): it found no frame
        ahead_failed_ = false;
        return nullptr;
    }
    if (ahead_valid_) { // parsed during the previous call
        b = ahead_buf_;
        t = ahead_time_;
        ahead_valid_ = false;
    } else if (!parseNext(&b, &t)) {
        return nullptr;
    }
    // synthesis of the whole frame on the device (audio.go:378-422): queued now, waited for after the next frame's parse
    static const int kFormat[] = {MPEGHIP_AUDIO_F32N, MPEGHIP_AUDIO_F32NLR, MPEGHIP_AUDIO_F32, MPEGHIP_AUDIO_S16};
    const int fmt = format_ == AudioF32N ? kFormat[0] : format_ == AudioF32NLR ? kFormat[1] : format_ == AudioF32 ? kFormat[2] : kFormat[3];
    const uint64_t ticket = backend_->synthAsync(in_[b], fmt);
    if (lookahead_) {
        // An attempt that fails for lack of data consumes nothing and is simply made again by the next call, which then sees what
        // the reference's call would see.  One that fails on a bad header HAS consumed bits (decodeHeader's hunt for a frame sync,
        // audio.go:184-272): it is the next call's attempt, made early — that call returns nil and does not hunt again.
        const size_t before = buf_->bitIndex();
        ended_before_ahead_ = buf_->HasEnded();
        ahead_tried_ = true;
        ahead_valid_ = parseNext(&ahead_buf_, &ahead_time_);
        ahead_failed_ = !ahead_valid_ && next_frame_data_size_ == 0 && buf_->bitIndex() != before;
    }
    Samples &sm = samples_[b];
    switch (format_) {
    case AudioF32N:
        backend_->synthWait(ticket, sm.Interleaved.data(), nullptr);
        break;
    case AudioF32NLR:
        backend_->synthWait(ticket, sm.Left.data(), sm.Right.data());
        break;
    case AudioS16:
        backend_->synthWait(ticket, sm.S16.data(), nullptr);
        break;
    case AudioF32:
        backend_->synthWait(ticket, sm.F32.data(), nullptr);
        break;
    }
    sm.Time = t;
    return &sm;
}

int Audio::decodeHeader()
{ // audio.go:184-272
    if (!buf_->has(48))
        return 0;
    buf_->skipBytes(0x00);
    const int sync = buf_->read(11);
    if (sync != kFrameSync && !buf_->findFrameSync())
        return 0;
    version_ = buf_->read(2);
    layer_ = buf_->read(2);
    const bool has_crc = buf_->read1() == 0;
    if (version_ != kMpeg1 || layer_ != kLayerII)
        return 0;
    const int bitrate_index = buf_->read(4) - 1;
    if (bitrate_index > 13 || bitrate_index < 0) // "free format" (-1) indexes out of range in the reference
        return 0;
    const int samplerate_index = buf_->read(2);
    if (samplerate_index == 3)
        return 0;
    const int padding = buf_->read1();
    buf_->skip(1);
    const int mode = buf_->read(2);
    if (has_header_ && (bitrate_index_ != bitrate_index || samplerate_index_ != samplerate_index || mode_ != mode))
        return 0;
    bitrate_index_ = bitrate_index;
    samplerate_index_ = samplerate_index;
    mode_ = mode;
    has_header_ = true;
    if (mode == kModeStereo || mode == kModeJointStereo)
        channels_ = 2;
    else if (mode == kModeMono)
        channels_ = 1;
    if (mode == kModeJointStereo) {
        bound_ = (buf_->read(2) + 1) << 2;
    } else {
        buf_->skip(2);
        bound_ = mode == kModeMono ? 0 : 32;
    }
    buf_->skip(4);
    if (has_crc)
        buf_->skip(16);
    const int frame_size = (144000 * (int)kBitrate[bitrate_index_] / (int)kSamplerate[samplerate_index_]) + padding;
    return frame_size - (has_crc ? 6 : 4);
}

const Audio::QuantizerSpec *Audio::readAllocation(int sb, int tab3)
{ // audio.go:429-438
    const int tab4 = kQuantLutStep3[tab3][sb];
    const int qtab = kQuantLutStep4[tab4 & 15][buf_->read(tab4 >> 4)];
    return qtab ? &quant_tab_[qtab - 1] : nullptr;
}

void Audio::readSamples(int ch, int sb, int part)
{ // audio.go:440-490
    const QuantizerSpec *q = allocation_[ch][sb];
    int sf = scale_factor_[ch][sb][part];
    int *s = sample_[ch][sb];
    if (!q) {
        s[0] = s[1] = s[2] = 0;
        return;
    }
    if (sf == 63) {
        sf = 0;
    } else {
        const int shift = sf / 3;
        sf = (kScalefactorBase[sf % 3] + ((1 << shift) >> 1)) >> shift;
    }
    int adj = q->Levels;
    if (q->Group) {
        int val = buf_->read(q->Bits);
        s[0] = val % adj;
        val /= adj;
        s[1] = val % adj;
        s[2] = val / adj;
    } else {
        s[0] = buf_->read(q->Bits);
        s[1] = buf_->read(q->Bits);
        s[2] = buf_->read(q->Bits);
    }
    const int scale = 65536 / (adj + 1);
    adj = ((adj + 1) >> 1) - 1;
    for (int k = 0; k < 3; k++) {
        const int64_t val = (int64_t)(adj - s[k]) * scale; // Go int is 64 bit
        s[k] = (int)((val * (sf >> 12) + ((val * (sf & 4095) + 2048) >> 12)) >> 12);
    }
}

void Audio::decodeFrame(int32_t (*frame_samples_)[36][32])
{ // audio.go:274-427
    const int tab1 = mode_ == kModeMono ? 0 : 1;
    const int tab2 = kQuantLutStep1[tab1][bitrate_index_];
    int tab3 = kQuantLutStep2[tab2][samplerate_index_];
    const int sblimit = tab3 & 63;
    tab3 >>= 6;
    if (bound_ > sblimit)
        bound_ = sblimit;

    for (int sb = 0; sb < bound_; sb++) {
        allocation_[0][sb] = readAllocation(sb, tab3);
        allocation_[1][sb] = readAllocation(sb, tab3);
    }
    for (int sb = bound_; sb < sblimit; sb++) {
        allocation_[0][sb] = readAllocation(sb, tab3);
        allocation_[1][sb] = allocation_[0][sb];
    }
    const int channels = mode_ == kModeMono ? 1 : 2;
    for (int sb = 0; sb < sblimit; sb++) {
        for (int ch = 0; ch < channels; ch++)
            if (allocation_[ch][sb])
                scale_factor_info_[ch][sb] = (uint8_t)buf_->read(2);
        if (mode_ == kModeMono)
            scale_factor_info_[1][sb] = scale_factor_info_[0][sb];
    }
    for (int sb = 0; sb < sblimit; sb++) {
        for (int ch = 0; ch < channels; ch++) {
            if (!allocation_[ch][sb])
                continue;
            int *sf = scale_factor_[ch][sb];
            switch (scale_factor_info_[ch][sb]) {
            case 0:
                sf[0] = buf_->read(6);
                sf[1] = buf_->read(6);
                sf[2] = buf_->read(6);
                break;
            case 1:
                sf[0] = sf[1] = buf_->read(6);
                sf[2] = buf_->read(6);
                break;
            case 2:
                sf[0] = sf[1] = sf[2] = buf_->read(6);
                break;
            case 3:
                sf[0] = buf_->read(6);
                sf[1] = sf[2] = buf_->read(6);
                break;
            }
        }
        if (mode_ == kModeMono)
            memcpy(scale_factor_[1][sb], scale_factor_[0][sb], sizeof(scale_factor_[0][sb]));
    }

    // Coefficient input: record the 36 sub-blocks instead of synthesising them inline
    int t = 0;
    for (int part = 0; part < 3; part++) {
        for (int granule = 0; granule < 4; granule++) {
            for (int sb = 0; sb < bound_; sb++) {
                readSamples(0, sb, part);
                readSamples(1, sb, part);
            }
            for (int sb = bound_; sb < sblimit; sb++) {
                readSamples(0, sb, part);
                memcpy(sample_[1][sb], sample_[0][sb], sizeof(sample_[0][sb]));
            }
            for (int sb = sblimit; sb < 32; sb++) {
                memset(sample_[0][sb], 0, sizeof(sample_[0][sb]));
                memset(sample_[1][sb], 0, sizeof(sample_[1][sb]));
            }
            for (int p = 0; p < 3; p++, t++)
                for (int ch = 0; ch < 2; ch++) // both channels, also for mono (audio.go:382)
                    for (int sb = 0; sb < 32; sb++)
                        frame_samples_[ch][t][sb] = sample_[ch][sb][p];
        }
    }
    buf_->align();

}

} // namespace mpeg
