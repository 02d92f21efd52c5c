// demux.cpp — mpeg::Demux: MPEG program stream -> PES packets (mirrors demux.go's
// HasHeaders / Decode path; Seek / Duration / Probe are not ported yet).
#include "mpeg.hpp"

namespace mpeg {

namespace {
constexpr int kStartPack = 0xBA, kStartSystem = 0xBB;
}

Demux::Demux(Buffer *buf) : buf_(buf) { HasHeaders(); } // demux.go:61-76 (ErrInvalidHeader <=> !HasHeaders())

bool Demux::HasHeaders()
{ // demux.go:85-155
    if (has_headers_)
        return true;
    if (!has_pack_header_) {
        if (start_code_ != kStartPack && buf_->findStartCode(kStartPack) == -1)
            return false;
        start_code_ = kStartPack;
        if (!buf_->has(64))
            return false;
        start_code_ = -1;
        if (buf_->read(4) != 0x02)
            return false;
        sys_clock_ref_ = decodeTime();
        buf_->skip(1);
        buf_->skip(22);
        buf_->skip(1);
        has_pack_header_ = true;
    }
    if (!has_system_header_) {
        if (start_code_ != kStartSystem && buf_->findStartCode(kStartSystem) == -1)
            return false;
        start_code_ = kStartSystem;
        if (!buf_->has(56))
            return false;
        start_code_ = -1;
        buf_->skip(16); // header_length
        buf_->skip(24); // rate bound
        num_audio_streams_ = buf_->read(6);
        buf_->skip(5);  // misc flags
        num_video_streams_ = buf_->read(5);
        has_system_header_ = true;
    }
    has_headers_ = true;
    return true;
}

void Demux::Rewind()
{ // demux.go:200-206
    buf_->Rewind();
    current_.length = 0;
    next_.length = 0;
    start_code_ = -1;
}

double Demux::decodeTime()
{ // demux.go:518-529
    int64_t clock = (int64_t)buf_->read(3) << 30;
    buf_->skip(1);
    clock |= (int64_t)buf_->read(15) << 15;
    buf_->skip(1);
    clock |= (int64_t)buf_->read(15);
    buf_->skip(1);
    return (double)clock / 90000.0;
}

Packet *Demux::Decode()
{ // demux.go:473-516
    if (!HasHeaders())
        return nullptr;
    if (current_.length != 0) {
        const size_t bits = (size_t)current_.length << 3;
        if (!buf_->has(bits))
            return nullptr;
        buf_->skip(bits);
        current_.length = 0;
    }
    if (next_.length != 0)
        return packet();
    if (start_code_ != -1)
        return decodePacket(start_code_);
    for (;;) {
        start_code_ = buf_->nextStartCode();
        if (start_code_ == PacketVideo1 || start_code_ == PacketPrivate ||
            (start_code_ >= PacketAudio1 && start_code_ <= PacketAudio4))
            return decodePacket(start_code_);
        if (start_code_ == -1)
            break;
    }
    return nullptr;
}

Packet *Demux::decodePacket(int type)
{ // demux.go:531-568
    if (!buf_->has(16 << 3))
        return nullptr;
    start_code_ = -1;
    next_.Type = type;
    next_.length = buf_->read(16);
    next_.length -= buf_->skipBytes(0xff); // stuffing
    if (buf_->read(2) == 0x01) {           // P-STD
        buf_->skip(16);
        next_.length -= 2;
    }
    const int marker = buf_->read(2);
    if (marker == 0x03) {
        next_.Pts = decodeTime();
        buf_->skip(40); // DTS
        next_.length -= 10;
    } else if (marker == 0x02) {
        next_.Pts = decodeTime();
        next_.length -= 5;
    } else if (marker == 0x00) {
        next_.Pts = PacketInvalidTS;
        buf_->skip(4);
        next_.length -= 1;
    } else {
        return nullptr; // invalid
    }
    return packet();
}

Packet *Demux::packet()
{ // demux.go:570-584
    if (next_.length < 0 || !buf_->has((size_t)next_.length << 3))
        return nullptr;
    current_.Data = buf_->Bytes() + buf_->Index();
    current_.Len = (size_t)next_.length;
    current_.Type = next_.Type;
    current_.Pts = next_.Pts;
    current_.length = next_.length;
    next_.length = 0;
    return &current_;
}

} // namespace mpeg
