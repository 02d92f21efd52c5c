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

// The coefficient table with what FOLLOWS each code folded into the lookup (video.go:685-707 reads them one after the other:
// the code; after a '1' that is not the block's first coefficient one bit that says end_of_block; then the sign bit — or, after
// the escape code, run and level fields).  One probe of the next kFirst bits yields run, SIGNED level and the total length, or
// says which of the rare cases it is; same fields, same order, same consumption as the reference, just not one bit at a time.
// `first`: the table for a block's very first coefficient (n == 0: a non-intra block), where '1' is run 0 / level 1 and never
// end_of_block.
class CoeffTable {
public:
    enum Kind : uint8_t {
        kCoef = 0,   // run, level != 0, len = code + sign bit (or '11s')
        kEnd = 1,    // '10' after the first coefficient: end_of_block, len = 2
        kEscape = 2, // len = the escape code alone; run (6 bits) and level (8 or 16 bits) follow
        kZero = 3,   // a dead end of the reference's tree: it returns value 0 = run 0, level 0, and the sign bit is still read
        kLink = 4,   // first level only: `level` is the offset of this prefix's second table
        kCoefEnd = 5, // a kCoef whose next two bits are '10' = end_of_block, both inside the first-level probe: len covers all of it
    };
    struct Entry {
        int32_t level;
        uint8_t run, len, kind, pad;
    };
    static constexpr int kFirst = 10;

    CoeffTable(const mpg_vlc_code *codes, bool first)
    {
        int longest = 0;
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            const int L = (int)strlen(c->bits) + 2; // ('1' + end-of-block bit + sign is the only + 2, and it is short; + 1 elsewhere)
            if (L - 1 > longest)
                longest = L - 1;
        }
        bits_ = longest;
        rest_bits_ = bits_ > kFirst ? bits_ - kFirst : 0;
        first_.assign((size_t)1 << kFirst, Entry{0, 0, 0, kZero, 0});
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            const int L = (int)strlen(c->bits);
            uint32_t code = 0;
            for (int k = 0; k < L; k++)
                code = (code << 1) | (uint32_t)(c->bits[k] - '0');
            const int value = c->dead ? 0 : c->value;
            if (value == 0xffff) {
                put(code, L, Entry{0, 0, (uint8_t)L, kEscape, 0});
            } else if (value == 0x0001 && !first) {
                put(code << 1, L + 1, Entry{0, 0, (uint8_t)(L + 1), kEnd, 0});
                put((code << 2) | 2, L + 2, Entry{1, 0, (uint8_t)(L + 2), kCoef, 0});
                put((code << 2) | 3, L + 2, Entry{-1, 0, (uint8_t)(L + 2), kCoef, 0});
            } else {
                const int level = value & 0xff, run = value >> 8;
                const uint8_t kind = level == 0 ? kZero : kCoef;
                put(code << 1, L + 1, Entry{level, (uint8_t)run, (uint8_t)(L + 1), kind, 0});
                put((code << 1) | 1, L + 1, Entry{-level, (uint8_t)run, (uint8_t)(L + 1), kind, 0});
            }
        }
    }
    int bits() const { return bits_; }
    // the symbol whose code starts at the window's top bit
    const Entry &at(uint64_t window) const { return at(first_.data(), rest_.data(), rest_bits_, window); }
    // ... for a loop that keeps the tables' addresses in registers (its stores could alias the vectors' own pointers)
    const Entry *firstLevel() const { return first_.data(); }
    const Entry *secondLevel() const { return rest_.data(); }
    int restBits() const { return rest_bits_; }
    static const Entry &at(const Entry *first, const Entry *rest, int rest_bits, uint64_t window)
    {
        const Entry &e = first[(size_t)(window >> (64 - kFirst))];
        if (__builtin_expect(e.kind != kLink, 1))
            return e;
        return rest[(size_t)e.level + (size_t)((window << kFirst) >> (64 - rest_bits))];
    }

private:
    void put(uint32_t code, int L, const Entry &e)
    {
        if (L <= kFirst) {
            const uint32_t lo = code << (kFirst - L), n = 1u << (kFirst - L);
            for (uint32_t k = 0; k < n; k++)
                first_[lo + k] = e;
            if (e.kind == kCoef && L + 2 <= kFirst) { // ... followed by '10': the block's last coefficient and its end in one probe
                Entry last = e;
                last.kind = kCoefEnd;
                last.len = (uint8_t)(L + 2);
                const uint32_t lo2 = ((code << 2) | 2u) << (kFirst - L - 2), n2 = 1u << (kFirst - L - 2);
                for (uint32_t k = 0; k < n2; k++)
                    first_[lo2 + k] = last;
            }
            return;
        }
        Entry &link = first_[code >> (L - kFirst)];
        if (link.kind != kLink) {
            link = Entry{(int32_t)rest_.size(), 0, 0, kLink, 0};
            rest_.resize(rest_.size() + ((size_t)1 << rest_bits_), Entry{0, 0, 0, kZero, 0});
        }
        const uint32_t tail = code & ((1u << (L - kFirst)) - 1);
        const uint32_t lo = tail << (bits_ - L), n = 1u << (bits_ - L);
        for (uint32_t k = 0; k < n; k++)
            rest_[(size_t)link.level + lo + k] = e;
    }
    int bits_, rest_bits_;
    std::vector<Entry> first_, rest_;
};

// Two symbols per probe.  The next kBits bits of the stream index a table that answers: the coefficient that starts there and,
// where it fits in the same bits, the coefficient (or the end_of_block, or both) behind it — for the symbols AFTER a block's first
// (CoeffTable's `next` context; a block's first symbol, escapes, dead ends and the codes that do not fit go the one-symbol way).
// Built from the one-symbol table by decoding every kBits-bit prefix twice; checked against it for every prefix with filled-in
// continuations (Video::VlcSelfCheck).  The loop that reads it stores the second pair word unconditionally and counts it or not:
// whether a second symbol fits is a coin flip, and must not be a branch.
class CoeffPairTable {
public:
    static constexpr int kBits = 11;
    enum : uint8_t { kSecond = 1, kEndOfBlock = 2 };
    struct Entry {
        int8_t level1;
        uint8_t run1;
        int8_t level2;  // 0 / run2 0 without a second coefficient
        uint8_t run2;
        uint8_t len1, len2; // bits of each symbol with its sign (len2 0 without a second one)
        uint8_t flags;      // kSecond, kEndOfBlock (2 more bits, behind the last coefficient)
        uint8_t total;      // all of it; 0: not answered here — one symbol at a time
    };
    explicit CoeffPairTable(const CoeffTable &next)
    {
        tab_.assign((size_t)1 << kBits, Entry{0, 0, 0, 0, 0, 0, 0, 0});
        for (uint32_t p = 0; p < (1u << kBits); p++) {
            const uint64_t w = (uint64_t)p << (64 - kBits);
            const CoeffTable::Entry &a = next.at(w);
            if ((a.kind != CoeffTable::kCoef && a.kind != CoeffTable::kCoefEnd) || a.len > kBits || a.level < -128 || a.level > 127)
                continue;
            Entry e{(int8_t)a.level, a.run, 0, 0, a.len, 0, 0, a.len};
            if (a.kind == CoeffTable::kCoefEnd) { // (its len covers the '10')
                e.len1 = (uint8_t)(a.len - 2);
                e.flags = kEndOfBlock;
                tab_[p] = e;
                continue;
            }
            const int left = kBits - a.len; // bits of the prefix behind the first symbol: a second one counts if it lies within them
            const CoeffTable::Entry &b = next.at(w << a.len);
            if (b.len <= left && b.level >= -128 && b.level <= 127) {
                if (b.kind == CoeffTable::kEnd) {
                    e.flags = kEndOfBlock;
                    e.total = (uint8_t)(a.len + 2);
                } else if (b.kind == CoeffTable::kCoef || b.kind == CoeffTable::kCoefEnd) {
                    const bool end = b.kind == CoeffTable::kCoefEnd;
                    e.level2 = (int8_t)b.level;
                    e.run2 = b.run;
                    e.len2 = (uint8_t)(end ? b.len - 2 : b.len);
                    e.flags = (uint8_t)(kSecond | (end ? kEndOfBlock : 0));
                    e.total = (uint8_t)(a.len + b.len);
                }
            }
            tab_[p] = e;
        }
    }
    const Entry *data() const { return tab_.data(); }
    const Entry &at(uint64_t window) const { return tab_[(size_t)(window >> (64 - kBits))]; }

private:
    std::vector<Entry> tab_;
};

} // namespace mpeg
