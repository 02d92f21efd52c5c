// buffer.cpp — mpeg::Buffer, the data source of all decoders (mirrors buffer.go).
#include <string.h>

#include "mpeg.hpp"

namespace mpeg {

size_t Buffer::BufferSize = 128 * 1024; // buffer.go:10

Buffer::Buffer(Reader reader) : reader_(std::move(reader))
{
    // buffer.go:32-61
    has_reader_ = (bool)reader_.read;
    if (has_reader_ && reader_.seek)
        total_size_ = reader_.size;
    bytes_.reserve(BufferSize);
    available_.resize(BufferSize);
    discard_read_ = true;
}

std::unique_ptr<Buffer> Buffer::FromMemory(const uint8_t *data, size_t len)
{
    // bytes.NewReader(data) + SetLoadCallback(LoadReaderCallback), the way the
    // reference's tests build their buffers (mpeg_test.go:25-30)
    auto pos = std::make_shared<size_t>(0);
    Reader r;
    r.read = [data, len, pos](uint8_t *dst, size_t n) {
        size_t k = len - *pos < n ? len - *pos : n;
        memcpy(dst, data + *pos, k);
        *pos += k;
        return k;
    };
    r.seek = [len, pos](size_t p) {
        *pos = p > len ? len : p;
        return true;
    };
    r.tell = [pos]() { return *pos; };
    r.size = len;
    std::unique_ptr<Buffer> b(new Buffer(std::move(r)));
    Buffer *raw = b.get();
    b->SetLoadCallback([raw](Buffer *x) { raw->LoadReaderCallback(x); });
    return b;
}

size_t Buffer::Write(const uint8_t *p, size_t n)
{ // buffer.go:79-89
    if (discard_read_)
        discardReadBytes();
    bytes_.insert(bytes_.end(), p, p + n);
    has_ended_ = false;
    return n;
}

void Buffer::LoadReaderCallback(Buffer *)
{ // buffer.go:131-156
    if (has_ended_)
        return;
    size_t n = 0;
    while (n < available_.size()) { // io.ReadFull
        size_t k = reader_.read(available_.data() + n, available_.size() - n);
        if (k == 0)
            break;
        n += k;
    }
    if (n == 0) {
        has_ended_ = true;
        return;
    }
    Write(available_.data(), n);
}

void Buffer::seek(size_t pos)
{ // buffer.go:158-176
    has_ended_ = false;
    if (has_reader_ && total_size_ > 0) {
        reader_.seek(pos);
        bytes_.clear();
        bit_index_ = 0;
    } else if (!has_reader_) {
        if (pos != 0)
            return;
        bytes_.clear();
        bit_index_ = 0;
    }
}

size_t Buffer::tell()
{ // buffer.go:178-187
    if (has_reader_ && total_size_ > 0 && reader_.tell)
        return reader_.tell() + (bit_index_ >> 3) - bytes_.size();
    return bit_index_ >> 3;
}

void Buffer::discardReadBytes()
{ // buffer.go:189-201
    size_t byte_pos = bit_index_ >> 3;
    if (byte_pos == bytes_.size()) {
        bytes_.clear();
        bit_index_ = 0;
    } else if (byte_pos > 0) {
        bytes_.erase(bytes_.begin(), bytes_.begin() + (ptrdiff_t)byte_pos);
        bit_index_ -= byte_pos << 3;
    }
}

bool Buffer::has(size_t count)
{ // buffer.go:203-221
    if ((bytes_.size() << 3) >= bit_index_ && (bytes_.size() << 3) - bit_index_ >= count)
        return true;
    if (load_) {
        load_(this);
        if ((bytes_.size() << 3) >= bit_index_ && (bytes_.size() << 3) - bit_index_ >= count)
            return true;
    }
    if (total_size_ != 0 && bytes_.size() == total_size_)
        has_ended_ = true;
    return false;
}

uint32_t Buffer::peek(int count)
{
    // up to 24 bits starting at the cursor, zero-padded past the end (the reference
    // would index out of range and panic there)
    const size_t byte = bit_index_ >> 3, n = bytes_.size();
    uint32_t w = 0;
    for (size_t k = 0; k < 4; k++)
        w = (w << 8) | (byte + k < n ? bytes_[byte + k] : 0u);
    w <<= (bit_index_ & 7);
    return count ? w >> (32 - count) : 0;
}

int Buffer::read(int count)
{ // buffer.go:223-244
    int value = 0;
    while (count > 0) {
        int take = count > 16 ? 16 : count;
        value = (value << take) | (int)peek(take);
        bit_index_ += (size_t)take;
        count -= take;
    }
    return value;
}

int Buffer::read1()
{ // buffer.go:246-255
    int v = (int)peek(1);
    bit_index_++;
    return v;
}

void Buffer::skip(size_t count)
{ // buffer.go:261-265
    if (has(count))
        bit_index_ += count;
}

int Buffer::skipBytes(uint8_t v)
{ // buffer.go:267-277
    align();
    int skipped = 0;
    while (has(8) && bytes_[bit_index_ >> 3] == v) {
        bit_index_ += 8;
        skipped++;
    }
    return skipped;
}

int Buffer::nextStartCode()
{ // buffer.go:279-302
    align();
    for (;;) {
        while ((bytes_.size() << 3) >= bit_index_ + (5 << 3)) {
            const size_t i = bit_index_ >> 3;
            const uint8_t *d = bytes_.data();
            if (d[i] == 0 && d[i + 1] == 0 && d[i + 2] == 1) {
                bit_index_ = (i + 4) << 3;
                return d[i + 3];
            }
            bit_index_ += 8;
        }
        if (!has(5 << 3))
            return -1;
    }
}

int Buffer::findStartCode(int code)
{ // buffer.go:304-311
    for (;;) {
        int cur = nextStartCode();
        if (cur == code || cur == -1)
            return cur;
    }
}

int Buffer::hasStartCode(int code)
{ // buffer.go:313-324
    const size_t prev = bit_index_;
    const bool prev_discard = discard_read_;
    discard_read_ = false;
    int cur = findStartCode(code);
    bit_index_ = prev;
    discard_read_ = prev_discard;
    return cur;
}

bool Buffer::findFrameSync()
{ // buffer.go:326-339
    size_t i;
    for (i = bit_index_ >> 3; i + 1 < bytes_.size(); i++) {
        if (bytes_[i] == 0xFF && (bytes_[i + 1] & 0xFE) == 0xFC) {
            bit_index_ = ((i + 1) << 3) + 3;
            return true;
        }
    }
    bit_index_ = (i + 1) << 3;
    return false;
}

bool Buffer::peekNonZero(int bitCount)
{ // buffer.go:341-350
    if (!has((size_t)bitCount))
        return false;
    return peek(bitCount) != 0; // bitCount <= 24
}

} // namespace mpeg
