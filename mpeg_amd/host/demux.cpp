// demux.cpp — mpeg::Demux: MPEG program stream -> PES packets (mirrors demux.go).
#include <algorithm>
#include <vector>

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
        last_decoded_pts_ = next_.Pts;
        buf_->skip(40); // DTS
        next_.length -= 10;
    } else if (marker == 0x02) {
        next_.Pts = decodeTime();
        last_decoded_pts_ = next_.Pts;
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

void Demux::bufferSeek(size_t pos)
{ // demux.go:513-518
    buf_->seek(pos);
    current_.length = 0;
    next_.length = 0;
    start_code_ = -1;
}

bool Demux::Probe(size_t probe_size)
{ // demux.go:158-198
    const size_t prev_pos = buf_->tell();
    bool video = false, audio[4] = {false, false, false, false};
    for (;;) {
        start_code_ = buf_->nextStartCode();
        if (start_code_ == PacketVideo1)
            video = true;
        else if (start_code_ >= PacketAudio1 && start_code_ <= PacketAudio4)
            audio[start_code_ - PacketAudio1] = true;
        if (start_code_ == -1 || buf_->tell() - prev_pos > probe_size)
            break;
    }
    num_video_streams_ = video ? 1 : 0;
    num_audio_streams_ = 0;
    for (bool a : audio)
        num_audio_streams_ += a ? 1 : 0;
    buf_->seek(prev_pos);
    return num_video_streams_ > 0 || num_audio_streams_ > 0;
}

double Demux::StartTime(int type)
{ // demux.go:357-404: the lowest PTS within one second of the first one (B-frame reordering)
    auto it = start_time_.find(type);
    if (it != start_time_.end())
        return it->second;
    const size_t prev_pos = buf_->tell();
    const int prev_start_code = start_code_;
    double start = PacketInvalidTS, anchor = PacketInvalidTS;
    Rewind();
    for (;;) {
        Packet *p = Decode();
        if (!p)
            break;
        if (p->Type != type || p->Pts == PacketInvalidTS)
            continue;
        if (anchor == PacketInvalidTS) {
            anchor = start = p->Pts;
        } else {
            if (p->Pts < start)
                start = p->Pts;
            if (p->Pts >= anchor + 1.0) // reorderWindow
                break;
        }
    }
    bufferSeek(prev_pos);
    start_code_ = prev_start_code;
    if (start != PacketInvalidTS) {
        start_time_[type] = start;
        first_pts_[type] = anchor;
    }
    return start;
}

static double frameStep(const std::vector<double> &sorted)
{ // demux.go:459-473: the smallest positive gap
    double step = PacketInvalidTS;
    for (size_t i = 1; i < sorted.size(); i++) {
        const double gap = sorted[i] - sorted[i - 1];
        if (gap > 0 && (step == PacketInvalidTS || gap < step))
            step = gap;
    }
    return step == PacketInvalidTS ? 0.0 : step;
}

double Demux::Duration(int type)
{ // demux.go:406-457
    const size_t file_size = buf_->Size();
    auto it = duration_.find(type);
    if (it != duration_.end() && last_file_size_ == file_size)
        return it->second;
    const size_t prev_pos = buf_->tell();
    const int prev_start_code = start_code_;
    // the highest PTS: search the last 64 KiB, then further back
    const long max_range = 4096 * 1024;
    for (long r = 64 * 1024; r <= max_range; r *= 2) {
        long seek_pos = (long)file_size - r;
        if (seek_pos < 0) {
            seek_pos = 0;
            r = max_range; // last round
        }
        bufferSeek((size_t)seek_pos);
        current_.length = 0;
        std::vector<double> pts;
        for (;;) {
            Packet *p = Decode();
            if (!p)
                break;
            if (p->Pts != PacketInvalidTS && p->Type == type)
                pts.push_back(p->Pts);
        }
        if (!pts.empty()) {
            std::sort(pts.begin(), pts.end());
            last_pts_[type] = pts.back();
            duration_[type] = pts.back() - StartTime(type) + frameStep(pts);
            break;
        }
    }
    bufferSeek(prev_pos);
    start_code_ = prev_start_code;
    last_file_size_ = file_size;
    return duration_[type];
}

Packet *Demux::Seek(double seek_time, int type, bool force_intra)
{ // demux.go:208-352: jump by estimated byte rate, scan forward for the last (intra) packet before seek_time
    if (!has_headers_)
        return nullptr;
    Duration(type);
    const double start_pts = first_pts_[type];
    const double span = last_pts_[type] - start_pts;
    const long file_size = (long)buf_->Size();
    double byte_rate = (double)file_size / span;
    double cur_time = last_decoded_pts_;
    double scan_span = 1;
    if (seek_time > span)
        seek_time = span;
    else if (seek_time < 0)
        seek_time = 0;
    seek_time += start_pts;

    for (int retry = 0; retry < 32; retry++) {
        bool found_packet_with_pts = false, found_packet_in_range = false;
        long last_valid_packet_start = -1;
        double first_packet_time = PacketInvalidTS;
        const long cur_pos = (long)buf_->tell();
        const double offset = (seek_time - cur_time - scan_span) * byte_rate;
        long seek_pos = cur_pos + (long)offset; // Go's int(float64): toward zero
        if (seek_pos < 0)
            seek_pos = 0;
        else if (seek_pos > file_size - 256)
            seek_pos = file_size - 256;
        bufferSeek((size_t)seek_pos);

        while (buf_->findStartCode(type) != -1) {
            const long packet_start = (long)buf_->tell();
            Packet *p = decodePacket(type);
            if (!p || p->Pts == PacketInvalidTS)
                continue;
            if (p->Pts > seek_time || p->Pts < seek_time - scan_span) { // outside: refine the estimate and jump again
                found_packet_with_pts = true;
                byte_rate = (double)(seek_pos - cur_pos) / (p->Pts - cur_time);
                cur_time = p->Pts;
                break;
            }
            if (!found_packet_in_range) {
                found_packet_in_range = true;
                first_packet_time = p->Pts;
            }
            if (force_intra) {
                for (long i = 0; i < (long)p->Len - 6; i++) {
                    if (p->Data[i] == 0x00 && p->Data[i + 1] == 0x00 && p->Data[i + 2] == 0x01 && p->Data[i + 3] == 0x00) {
                        if ((p->Data[i + 5] & 0x38) == 8) // picture_coding_type 1 = intra
                            last_valid_packet_start = packet_start;
                        break;
                    }
                }
            } else {
                last_valid_packet_start = packet_start;
            }
        }
        if (last_valid_packet_start != -1) {
            bufferSeek((size_t)last_valid_packet_start);
            return decodePacket(type);
        }
        if (found_packet_in_range) { // right range, no intra frame: widen
            scan_span *= 2;
            seek_time = first_packet_time;
        } else if (!found_packet_with_pts) { // probably ran off the end
            byte_rate = (double)(seek_pos - cur_pos) / (span - cur_time);
            cur_time = span;
        }
    }
    return nullptr;
}

} // namespace mpeg
