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
        // Two levels: the next kFirst bits index a table that fits the L1 cache (the coefficient table's longest code has 17
        // bits: flat, that is 1 MB of entries, and a probe lands anywhere in it — most of the parser's cache misses); codes
        // longer than kFirst bits continue in a second table per distinct kFirst-bit prefix, indexed by the rest.
        first_bits_ = bits_ < kFirst ? bits_ : kFirst;
        rest_bits_ = bits_ - first_bits_;
        first_.assign((size_t)1 << first_bits_, Entry{0, 0});
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            const int L = (int)strlen(c->bits);
            uint32_t code = 0;
            for (int k = 0; k < L; k++)
                code = (code << 1) | (uint32_t)(c->bits[k] - '0');
            const Entry e{(int32_t)(c->dead ? 0 : c->value), (int32_t)L};
            if (L <= first_bits_) {
                const uint32_t lo = code << (first_bits_ - L), n = 1u << (first_bits_ - L);
                for (uint32_t k = 0; k < n; k++)
                    first_[lo + k] = e;
                continue;
            }
            Entry &link = first_[code >> (L - first_bits_)];
            if (link.len != kLink) { // this prefix's first long code: its second table
                link = Entry{(int32_t)rest_.size(), kLink};
                rest_.resize(rest_.size() + ((size_t)1 << rest_bits_), Entry{0, 0});
            }
            const uint32_t tail = code & ((1u << (L - first_bits_)) - 1);
            const uint32_t lo = tail << (bits_ - L), n = 1u << (bits_ - L);
            for (uint32_t k = 0; k < n; k++)
                rest_[(size_t)link.value + lo + k] = e;
        }
    }

    // decode one symbol at the buffer's cursor
    int read(Buffer *b) const
    {
        const Symbol s = at(b->window());
        b->drop(s.len);
        return s.value;
    }
    int bits() const { return bits_; }
    // for callers that decode several fields from one 64-bit look at the stream (Buffer::window): the symbol
    // whose code starts at the window's top bit
    struct Symbol { int32_t value, len; };
    Symbol at(uint64_t window) const
    {
        const Entry &e = first_[(size_t)(window >> (64 - first_bits_))];
        if (__builtin_expect(e.len != kLink, 1))
            return Symbol{e.value, e.len};
        const Entry &r = rest_[(size_t)e.value + (size_t)((window << first_bits_) >> (64 - rest_bits_))];
        return Symbol{r.value, r.len};
    }

private:
    struct Entry { int32_t value, len; };
    static constexpr int kFirst = 9;
    static constexpr int32_t kLink = -1; // Entry::len of a first-level entry that points into rest_ (value = offset)
    int bits_, first_bits_, rest_bits_;
    std::vector<Entry> first_, rest_;
};

} // namespace mpeg
