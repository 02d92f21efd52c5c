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
    if (byte_pos >= bytes_.size()) { // (beyond the end only after a read past it, where the reference has panicked: buffer.go:246-255)
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
{ // buffer.go:279-302: the first byte position i at or after the cursor with 00 00 01 xx inside the
  // buffered bytes (i + 5 <= size); same result and same final cursor as the reference's byte-by-byte
  // walk, found by looking for the 01 bytes (memchr) — this scan runs over every picture twice
  // (hasStartCode's look-ahead, then the slices), a third of the parser's time when done bytewise.
    align();
    for (;;) {
        const uint8_t *d = bytes_.data();
        const size_t n = bytes_.size();
        size_t i = bit_index_ >> 3;
        while (i + 5 <= n) {
            const uint8_t *p = static_cast<const uint8_t *>(memchr(d + i + 2, 1, (n - 2) - (i + 2)));
            if (!p) {
                i = n - 4; // where the bytewise walk stops: fewer than 5 bytes left
                break;
            }
            const size_t k = (size_t)(p - d);
            if (d[k - 1] == 0 && d[k - 2] == 0) {
                bit_index_ = (k + 2) << 3;
                return d[k + 1];
            }
            i = k - 1; // the next candidate has its 01 byte behind this one
        }
        bit_index_ = i << 3;
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
