// mpeg.cpp — mpeg::MPEG, the high-level facade (mirrors mpeg.go).
#include <stdexcept>
#include <string.h>

#include "mpeg.hpp"

namespace mpeg {

MPEG::MPEG(const uint8_t *data, size_t len, Device *dev, int audio_fma_mode) : audio_fma_mode_(audio_fma_mode)
{
    backends_.video = [dev]() { return dev->newVideoBackend(); };
    backends_.audio = [dev](int fma) { return dev->newAudioBackend(fma); };
    open(data, len);
}

MPEG::MPEG(const uint8_t *data, size_t len, Backends backends, int audio_fma_mode)
    : backends_(std::move(backends)), audio_fma_mode_(audio_fma_mode)
{
    open(data, len);
}

void MPEG::open(const uint8_t *data, size_t len)
{ // mpeg.go:85-117
    buf_ = Buffer::FromMemory(data, len);
    static const uint8_t magic[4] = {0x00, 0x00, 0x01, 0xBA};
    if (!buf_->has(32) || memcmp(magic, buf_->Bytes(), 4) != 0)
        throw std::runtime_error("invalid MPEG-PS"); // ErrInvalidMPEG (mpeg.go:55)
    buf_->Rewind();
    demux_.reset(new Demux(buf_.get()));
    if (!demux_->HasHeaders())
        throw std::runtime_error("invalid MPEG-PS header"); // ErrInvalidHeader (demux.go:32)
    initDecoders();
}

MPEG::~MPEG() {}

bool MPEG::HasHeaders()
{ // mpeg.go:121-141
    if (!demux_->HasHeaders() || !initDecoders())
        return false;
    if (video_ && !video_->HasHeader())
        return false;
    if (audio_ && !audio_->HasHeader())
        return false;
    return true;
}

void MPEG::SetVideoEnabled(bool e)
{ // mpeg.go:168-180
    video_enabled_ = e;
    if (!e) {
        video_packet_type_ = 0;
        return;
    }
    video_packet_type_ = (initDecoders() && video_) ? PacketVideo1 : 0;
}

void MPEG::SetAudioEnabled(bool e)
{ // mpeg.go:198-210
    audio_enabled_ = e;
    if (!e) {
        audio_packet_type_ = 0;
        return;
    }
    audio_packet_type_ = (initDecoders() && audio_) ? PacketAudio1 + audio_stream_index_ : 0;
}

void MPEG::SetAudioFormat(AudioFormat f)
{ // mpeg.go:234-240
    audio_format_ = f;
    if (audio_)
        audio_->SetFormat(f);
}

void MPEG::Rewind()
{ // mpeg.go:323-337
    if (video_)
        video_->Rewind();
    if (audio_)
        audio_->Rewind();
    demux_->Rewind();
    time_ = 0;
    has_ended_ = false; // mpeg.go:336
}

void MPEG::SetAudioStream(int stream_index)
{ // mpeg.go:270-279
    if (stream_index < 0 || stream_index > 3)
        return;
    audio_stream_index_ = stream_index;
    SetAudioEnabled(audio_enabled_); // sets the matching packet type
}

bool MPEG::Probe(size_t probe_size)
{ // mpeg.go:141-152
    if (!demux_->Probe(probe_size))
        return false;
    has_decoders_ = false;
    video_packet_type_ = 0;
    audio_packet_type_ = 0;
    return initDecoders();
}

Frame *MPEG::SeekFrame(double tm, bool seek_exact)
{ // mpeg.go:460-522
    if (!initDecoders() || video_packet_type_ == 0)
        return nullptr;
    const int type = video_packet_type_;
    const double start_time = demux_->StartTime(type);
    const double duration = demux_->Duration(type);
    if (tm < 0)
        tm = 0;
    else if (tm > duration)
        tm = duration;
    Packet *packet = demux_->Seek(tm, type, true);
    if (!packet)
        return nullptr;
    // no audio packets into the audio buffer while decoding video
    const int prev_audio_packet_type = audio_packet_type_;
    audio_packet_type_ = 0;
    video_->Rewind();
    video_->SetTime(packet->Pts - start_time);
    video_buf_->Write(packet->Data, packet->Len);
    // (nothing parsed ahead in here: a look-ahead would pull packets through the load callback while the audio packet type is
    // switched off, and audio packets the reference still delivers later would be dropped)
    const bool lookahead = video_->Lookahead();
    video_->SetLookahead(false);
    Frame *frame = video_->Decode();
    if (seek_exact)
        while (frame && frame->Time < tm)
            frame = video_->Decode();
    video_->SetLookahead(lookahead);
    audio_packet_type_ = prev_audio_packet_type;
    if (frame)
        time_ = frame->Time;
    has_ended_ = false;
    return frame;
}

bool MPEG::Seek(double tm, bool seek_exact)
{ // mpeg.go:524-576
    Frame *frame = SeekFrame(tm, seek_exact);
    if (!frame)
        return false;
    if (video_cb_)
        video_cb_(this, frame);
    if (audio_packet_type_ == 0)
        return true;
    // demux on until the first audio packet after the new time, then decode up to the lead time
    const double start_time = demux_->StartTime(video_packet_type_);
    audio_->Rewind();
    for (;;) {
        Packet *p = demux_->Decode();
        if (!p)
            break;
        if (p->Type == video_packet_type_) {
            video_buf_->Write(p->Data, p->Len);
        } else if (p->Type == audio_packet_type_ && p->Pts - start_time > time_) {
            audio_->SetTime(p->Pts - start_time);
            audio_buf_->Write(p->Data, p->Len);
            const int prev_audio_packet_type = audio_packet_type_;
            audio_packet_type_ = 0;
            Decode(0);
            audio_packet_type_ = prev_audio_packet_type;
            Decode(0);
            break;
        }
    }
    return true;
}

bool MPEG::initDecoders()
{ // mpeg.go:578-623
    if (has_decoders_)
        return true;
    if (!demux_->HasHeaders())
        return false;
    if (demux_->NumVideoStreams() > 0) {
        if (video_enabled_)
            video_packet_type_ = PacketVideo1;
        if (!video_) {
            video_buf_.reset(new Buffer());
            video_buf_->SetLoadCallback([this](Buffer *) { readPackets(video_packet_type_); });
            video_.reset(new Video(video_buf_.get(), backends_.video()));
        }
    }
    if (demux_->NumAudioStreams() > 0) {
        if (audio_enabled_)
            audio_packet_type_ = PacketAudio1 + audio_stream_index_;
        if (!audio_) {
            audio_buf_.reset(new Buffer());
            audio_buf_->SetLoadCallback([this](Buffer *) { readPackets(audio_packet_type_); });
            audio_.reset(new Audio(audio_buf_.get(), backends_.audio(audio_fma_mode_)));
            audio_->SetFormat(audio_format_);
        }
    }
    has_decoders_ = true;
    return true;
}

void MPEG::handleEnd()
{ // mpeg.go:625-632
    if (loop_) {
        Rewind();
    } else {
        has_ended_ = true;
        done_pending_ = true; // m.done <- true
        if (done_cb_)
            done_cb_();
    }
}

void MPEG::readPackets(int requested_type)
{ // mpeg.go:642-669
    for (;;) {
        Packet *p = demux_->Decode();
        if (!p)
            break;
        if (p->Type == video_packet_type_ && video_buf_)
            video_buf_->Write(p->Data, p->Len);
        else if (p->Type == audio_packet_type_ && audio_buf_)
            audio_buf_->Write(p->Data, p->Len);
        if (p->Type == requested_type)
            return;
    }
    if (demux_->HasEnded()) {
        if (video_buf_)
            video_buf_->SignalEnd();
        if (audio_buf_)
            audio_buf_->SignalEnd();
    }
}

void MPEG::Decode(double tick)
{ // mpeg.go:356-411
    if (!initDecoders())
        return;
    const bool decode_video = (bool)video_cb_ && video_packet_type_ != 0;
    const bool decode_audio = (bool)audio_cb_ && audio_packet_type_ != 0;
    if (!decode_video && !decode_audio)
        return;
    bool video_failed = false, audio_failed = false;
    const double video_target = time_ + tick, audio_target = time_ + tick + audio_lead_time_;
    for (;;) {
        bool did = false;
        if (decode_video && video_->Time() < video_target) {
            Frame *f = video_->Decode();
            if (f) {
                video_cb_(this, f);
                did = true;
            } else {
                video_failed = true;
            }
        }
        if (decode_audio && audio_->Time() < audio_target) {
            Samples *s = audio_->Decode();
            if (s) {
                audio_cb_(this, s);
                did = true;
            } else {
                audio_failed = true;
            }
        }
        if (!did)
            break;
    }
    if ((!decode_video || video_failed) && (!decode_audio || audio_failed) && demux_->HasEnded()) {
        handleEnd();
        return;
    }
    time_ += tick;
}

Frame *MPEG::DecodeVideo()
{ // mpeg.go:416-433
    if (!initDecoders() || video_packet_type_ == 0)
        return nullptr;
    Frame *f = video_->Decode();
    if (f)
        time_ = f->Time;
    else if (demux_->HasEnded())
        handleEnd();
    return f;
}

Samples *MPEG::DecodeAudio()
{ // mpeg.go:438-455
    if (!initDecoders() || audio_packet_type_ == 0)
        return nullptr;
    Samples *s = audio_->Decode();
    if (s)
        time_ = s->Time;
    else if (demux_->HasEnded())
        handleEnd();
    return s;
}

} // namespace mpeg
