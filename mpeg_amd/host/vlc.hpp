// vlc.hpp — table-driven VLC decoding for the host parser.
//
// The reference walks a binary code tree one bit per iteration
// (buffer.go:352-376).  Here every ISO 11172-2 table is expanded once into a flat
// lookup table indexed by the next L bits (L = longest code of the table), giving
// {value, length} in one probe.  A prefix that no code starts with ("dead end" of
// the reference's tree) yields value 0 and consumes exactly the bits the tree walk
// would have consumed, which the damaged golden stream relies on.
#pragma once

#include <stdint.h>
#include <string.h>

#include <vector>

#include "iso11172_vlc_codes.h"
#include "mpeg.hpp"

namespace mpeg {

class VlcTable {
public:
    explicit VlcTable(const mpg_vlc_code *codes)
    {
        bits_ = 0;
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            int L = (int)strlen(c->bits);
            if (L > bits_)
                bits_ = L;
        }
        lut_.assign((size_t)1 << bits_, Entry{0, 0});
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            const int L = (int)strlen(c->bits);
            uint32_t code = 0;
            for (int k = 0; k < L; k++)
                code = (code << 1) | (uint32_t)(c->bits[k] - '0');
            const uint32_t lo = code << (bits_ - L), n = 1u << (bits_ - L);
            for (uint32_t k = 0; k < n; k++)
                lut_[lo + k] = Entry{(int32_t)(c->dead ? 0 : c->value), (int32_t)L};
        }
    }

    // decode one symbol at the buffer's cursor
    int read(Buffer *b) const
    {
        uint32_t w = bits_ <= 24 ? b->peek(bits_) : 0;
        const Entry &e = lut_[w];
        b->drop(e.len);
        return e.value;
    }
    int bits() const { return bits_; }
    // for callers that decode several fields from one 64-bit look at the stream (Buffer::window): the symbol
    // whose code starts at the window's top bit
    struct Symbol { int32_t value, len; };
    Symbol at(uint64_t window) const
    {
        const Entry &e = lut_[(size_t)(window >> (64 - bits_))];
        return Symbol{e.value, e.len};
    }

private:
    struct Entry { int32_t value, len; };
    int bits_;
    std::vector<Entry> lut_;
};

} // namespace mpeg
