// video.cpp — mpeg::Video: MPEG-1 video bitstream parse on the CPU, reconstruction
// on the GPU.  Mirrors the parse half of video.go (:209-745) and replaces its
// reconstruction half (:608-637, :747-1016) by descriptor recording +
// mpeghip_video_submit.
//
// What is recorded per macroblock is exactly what the reference would have
// executed inline:
//   predictMacroblock  -> one (reference slot, half-pel vector) pair — the LAST copy
//                         wins, so a bidirectional B macroblock records only the
//                         backward prediction (video.go:626-630)
//   decodeBlock        -> the block's quantised levels (int16, dequantised on the GPU)
//                         or, where the reference's persistent blockData carries stale
//                         coefficients from an earlier invalid block, a snapshot of
//                         blockData itself (int32, MPEGHIP_MB_COEF_RAW)
#include <atomic>
#include <stdexcept>
#include <string.h>

#include <chrono>

#include "mpeg.hpp"
#include "vlc.hpp"

namespace mpeg {

static std::atomic<bool> g_default_sparse{true};
void Video::SetDefaultSparse(bool v) { g_default_sparse.store(v); }


namespace {

// The tables are objects of this translation unit, built when the library is loaded (a millisecond), not function-local statics:
// a macroblock reads five of them and a block three, and each accessor of the old form was a call with a guard check in the
// parser's inner loops (four calls per block: 8 % of the parse).
const VlcTable kTabMba(mpg_vlc_mba_increment), kTabTypeI(mpg_vlc_mb_type_i), kTabTypeP(mpg_vlc_mb_type_p), kTabTypeB(mpg_vlc_mb_type_b),
    kTabCbp(mpg_vlc_coded_block_pattern), kTabMotion(mpg_vlc_motion_code), kTabDcLuma(mpg_vlc_dct_dc_size_luma),
    kTabDcChroma(mpg_vlc_dct_dc_size_chroma);
const CoeffTable kTabCoeffFirst(mpg_vlc_dct_coeff, true), kTabCoeffNext(mpg_vlc_dct_coeff, false);
const CoeffPairTable kTabCoeffPairs(kTabCoeffNext); // (after the table it is built from: objects of one unit are built in order)
inline const VlcTable &tabMba() { return kTabMba; }
inline const VlcTable &tabType(int picture_type) { return picture_type == 1 ? kTabTypeI : (picture_type == 2 ? kTabTypeP : kTabTypeB); }
inline const VlcTable &tabCbp() { return kTabCbp; }
inline const VlcTable &tabMotion() { return kTabMotion; }
inline const VlcTable &tabDcSize(int plane) { return plane == 0 ? kTabDcLuma : kTabDcChroma; }
const VlcTable &tabCoeff() { static const VlcTable t(mpg_vlc_dct_coeff); return t; }   // (the plain table: only the self-check reads it)
inline const CoeffTable &tabCoeffFirst() { return kTabCoeffFirst; }
inline const CoeffTable &tabCoeffNext() { return kTabCoeffNext; }
inline const CoeffPairTable &tabCoeffPairs() { return kTabCoeffPairs; }

// the code a prefix starts with, found the slow way: the first code of the list that the prefix's leading bits spell out
// (the lists are prefix-free: the reference's tree walk, buffer.go:352-376, ends at exactly that code)
uint64_t vlcMismatches(const mpg_vlc_code *codes, const VlcTable &table)
{
    const int L = table.bits();
    uint64_t bad = 0;
    for (uint64_t prefix = 0; prefix < (1ull << L); prefix++) {
        int value = 0, len = 0;
        for (const mpg_vlc_code *c = codes; c->bits; c++) {
            const int n = (int)strlen(c->bits);
            uint64_t code = 0;
            for (int k = 0; k < n; k++)
                code = (code << 1) | (uint64_t)(c->bits[k] - '0');
            if ((prefix >> (L - n)) == code) {
                value = c->dead ? 0 : c->value;
                len = n;
                break;
            }
        }
        const VlcTable::Symbol got = table.at(prefix << (64 - L));
        bad += (got.value != value || got.len != len) ? 1 : 0;
    }
    return bad;
}

// CoeffTable against the plain table followed by the reads video.go:685-707 makes after the code, for every prefix
uint64_t coeffMismatches(const CoeffTable &table, bool first)
{
    const int L = table.bits();
    uint64_t bad = 0;
    for (uint64_t prefix = 0; prefix < (1ull << L); prefix++) {
        const uint64_t w = prefix << (64 - L);
        const VlcTable::Symbol sym = tabCoeff().at(w);
        const uint64_t rest = w << sym.len;
        int kind, run = 0, level = 0, len = sym.len;
        if (sym.value == 0x0001 && !first && (rest >> 63) == 0) {
            kind = CoeffTable::kEnd;
            len += 1;
        } else if (sym.value == 0xffff) {
            kind = CoeffTable::kEscape;
        } else {
            uint64_t r = rest;
            if (sym.value == 0x0001 && !first) {
                r <<= 1;
                len += 1;
            }
            run = sym.value >> 8;
            level = sym.value & 0xff;
            if (r >> 63)
                level = -level;
            len += 1;
            kind = level == 0 ? CoeffTable::kZero : CoeffTable::kCoef;
        }
        if (len > L)
            continue; // (the prefix is shorter than this symbol with its sign: longer prefixes cover it)
        if (kind == CoeffTable::kCoef && len + 2 <= CoeffTable::kFirst && ((w << len) >> 62) == 2) { // ... then '10' inside the probe
            kind = CoeffTable::kCoefEnd;
            len += 2;
        }
        const CoeffTable::Entry &e = table.at(w);
        bad += (e.kind != kind || e.len != len || ((kind == CoeffTable::kCoef || kind == CoeffTable::kCoefEnd || kind == CoeffTable::kZero) && (e.run != run || e.level != level))) ? 1 : 0;
    }
    return bad;
}

// CoeffPairTable against the one-symbol table: for every prefix, with the bits behind it filled in several ways, what the pair
// entry says (one or two coefficients, an end_of_block, the bits of each) is what reading one symbol at a time finds there
uint64_t pairMismatches(const CoeffPairTable &pairs, const CoeffTable &next)
{
    uint64_t bad = 0, fill = 0x9e3779b97f4a7c15ull;
    for (uint32_t p = 0; p < (1u << CoeffPairTable::kBits); p++)
        for (int f = 0; f < 12; f++) {
            fill = fill * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t tail = f == 0 ? 0 : (f == 1 ? ~0ull : fill);
            const uint64_t w = ((uint64_t)p << (64 - CoeffPairTable::kBits)) | (tail >> CoeffPairTable::kBits);
            const CoeffPairTable::Entry &e = pairs.at(w);
            if (e.total == 0)
                continue;
            struct Ev { int kind, run, level, len; } want[3], got[4];
            int nw = 0, ng = 0;
            want[nw++] = Ev{0, e.run1, e.level1, e.len1};
            if (e.flags & CoeffPairTable::kSecond)
                want[nw++] = Ev{0, e.run2, e.level2, e.len2};
            if (e.flags & CoeffPairTable::kEndOfBlock)
                want[nw++] = Ev{1, 0, 0, 2};
            int pos = 0;
            bool ok = true;
            while (ng < nw) { // the same events, read one symbol at a time ("coefficient + end_of_block" probes are two events)
                const CoeffTable::Entry &s1 = next.at(w << pos);
                if (s1.kind == CoeffTable::kCoef) {
                    got[ng++] = Ev{0, s1.run, s1.level, s1.len};
                } else if (s1.kind == CoeffTable::kCoefEnd) {
                    got[ng++] = Ev{0, s1.run, s1.level, s1.len - 2};
                    got[ng++] = Ev{1, 0, 0, 2};
                } else if (s1.kind == CoeffTable::kEnd) {
                    got[ng++] = Ev{1, 0, 0, 2};
                } else {
                    ok = false;
                    break;
                }
                pos += s1.len;
            }
            int sum = 0;
            for (int k = 0; ok && k < nw; k++) {
                ok = want[k].kind == got[k].kind && want[k].run == got[k].run && want[k].level == got[k].level && want[k].len == got[k].len;
                sum += want[k].len;
            }
            ok = ok && sum == e.total;
            bad += ok ? 0 : 1;
        }
    return bad;
}

constexpr int kPictureTypeIntra = 1, kPictureTypePredictive = 2, kPictureTypeB = 3;
constexpr int kStartPicture = 0x00, kStartSliceFirst = 0x01, kStartSliceLast = 0xAF, kStartUserData = 0xB2,
              kStartSequence = 0xB3, kStartExtension = 0xB5;

const double kPictureRate[16] = {0.000, 23.976, 24.000, 25.000, 29.970, 30.000, 50.000, 59.940,
                                 60.000, 0, 0, 0, 0, 0, 0, 0}; // ISO 11172-2 table 2-D.4 (video.go:1034-1037)

constexpr uint8_t kZigZag[64] = { // ISO 11172-2 zig-zag scan (video.go:1044-1053)
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// scan index -> the level's place in a pair word: position column * 8 + row of natural index row * 8 + column, << 2
struct ScanPos {
    uint8_t v[64];
    constexpr ScanPos() : v()
    {
        for (int n = 0; n < 64; n++)
            v[n] = (uint8_t)((((kZigZag[n] & 7) * 8) + (kZigZag[n] >> 3)) << 2);
    }
};
constexpr ScanPos kScanPos;
inline int naturalIndexOfPair(uint32_t pair) { const int pos = (int)((pair >> 2) & 63); return (pos & 7) * 8 + (pos >> 3); }
inline int levelOfPair(uint32_t pair) { return (int16_t)(pair >> 16); }

const uint8_t kDefaultIntraQuant[64] = { // ISO default intra matrix (video.go:1055-1064)
    8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34,
    34, 38, 22, 22, 26, 27, 29, 34, 37, 40, 22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32,
    35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};

const uint8_t kPremultiplier[64] = { // video.go:1077-1086
    32, 44, 42, 38, 32, 25, 17, 9,  44, 62, 58, 52, 44, 35, 24, 12, 42, 58, 55, 49, 42, 33,
    23, 12, 38, 52, 49, 44, 38, 30, 20, 10, 32, 44, 42, 38, 32, 25, 17, 9,  25, 35, 33, 30,
    25, 20, 14, 7,  17, 24, 23, 20, 17, 14, 9,  5,  9,  12, 12, 10, 9,  7,  5,  2};

// dequantise + premultiply one level on the host — only needed on the rare paths
// that must reproduce the reference's blockData bit for bit (video.go:719-744)
int32_t dequantPremult(int level, bool intra, int qscale, int qm, int idx)
{
    level *= 2; // (the reference shifts; a negative level shifted is undefined before C++20)
    if (!intra)
        level += level < 0 ? -1 : 1;
    level = (level * qscale * qm) >> 4;
    if ((level & 1) == 0)
        level -= level > 0 ? 1 : -1;
    if (level > 2047)
        level = 2047;
    else if (level < -2048)
        level = -2048;
    return level * (int)kPremultiplier[idx];
}

} // namespace

namespace {
template <class C>
inline int vlcAt(const VlcTable &t, C &c)
{
    const VlcTable::Symbol s = t.at(c.window());
    c.bit += (size_t)s.len;
    return s.value;
}
} // namespace

uint64_t Video::VlcSelfCheck()
{
    return vlcMismatches(mpg_vlc_mba_increment, tabMba()) + vlcMismatches(mpg_vlc_mb_type_i, tabType(1)) +
           vlcMismatches(mpg_vlc_mb_type_p, tabType(2)) + vlcMismatches(mpg_vlc_mb_type_b, tabType(3)) +
           vlcMismatches(mpg_vlc_coded_block_pattern, tabCbp()) + vlcMismatches(mpg_vlc_motion_code, tabMotion()) +
           vlcMismatches(mpg_vlc_dct_dc_size_luma, tabDcSize(0)) + vlcMismatches(mpg_vlc_dct_dc_size_chroma, tabDcSize(1)) +
           vlcMismatches(mpg_vlc_dct_coeff, tabCoeff()) + coeffMismatches(tabCoeffFirst(), true) + coeffMismatches(tabCoeffNext(), false) +
           pairMismatches(tabCoeffPairs(), tabCoeffNext());
}

// test hook: ONE look at the stream through one of the parser's tables (tests/test_vlc_known_answers.py walks the reference's code
// trees through it).  Table order: address increment, macroblock type I / P / B, coded block pattern, motion code, DC size
// luma / chroma, coefficients (as the plain code table: what follows a code — sign, end_of_block — is coeffMismatches' matter).
bool Video::VlcDecode(int table, uint64_t window, int *value, int *len)
{
    const VlcTable *t = nullptr;
    switch (table) {
    case 0: t = &tabMba(); break;
    case 1: t = &tabType(1); break;
    case 2: t = &tabType(2); break;
    case 3: t = &tabType(3); break;
    case 4: t = &tabCbp(); break;
    case 5: t = &tabMotion(); break;
    case 6: t = &tabDcSize(0); break;
    case 7: t = &tabDcSize(1); break;
    case 8: t = &tabCoeff(); break;
    default: return false;
    }
    const VlcTable::Symbol s = t->at(window);
    *value = s.value;
    *len = s.len;
    return true;
}

Video::Video(Buffer *buf, Device *dev) : buf_(buf), backend_(dev->newVideoBackend()) { init(); }
Video::Video(Buffer *buf, std::unique_ptr<VideoBackend> backend) : buf_(buf), backend_(std::move(backend)) { init(); }

void Video::init()
{ // video.go:110-121
    sparse_ = sparse_wanted_ = g_default_sparse.load();
    memset(block_data_, 0, sizeof(block_data_));
    memset(intra_quant_, 0, sizeof(intra_quant_));
    memset(non_intra_quant_, 0, sizeof(non_intra_quant_));
    start_code_ = buf_->findStartCode(kStartSequence);
    if (start_code_ != -1)
        decodeSequenceHeader();
}

Video::~Video()
{
    for (uint8_t *p : out_planes_)
        if (p)
            backend_->freePlanes(p);
}

bool Video::HasHeader()
{ // video.go:130-147
    if (has_sequence_header_)
        return true;
    if (start_code_ != kStartSequence)
        start_code_ = buf_->findStartCode(kStartSequence);
    if (start_code_ == -1)
        return false;
    return decodeSequenceHeader();
}

void Video::SetTime(double t)
{ // video.go:189-192
    frames_decoded_ = (int)(frame_rate_ * t);
    time_ = t;
    if (ahead_.valid) { // (a picture parsed ahead completes the next frame the reference would return: it carries the new time)
        ahead_.time = time_;
        frames_decoded_++;
        time_ = (double)frames_decoded_ / frame_rate_;
    }
}

// A picture parsed ahead (Decode) has not reached the device: forgetting it takes the recorded hand-overs and the parser state
// that outlives a picture — the frame store, and so every later prediction, is as the reference has it.
void Video::dropLookahead()
{
    ahead_.valid = false;
    ahead_tried_ = false;
    n_deferred_ = 0;
    if (!undo_valid_)
        return;
    undo_valid_ = false;
    slot_cur_ = undo_.cur;
    slot_fwd_ = undo_.fwd;
    slot_bwd_ = undo_.bwd;
    picture_type_ = undo_.picture_type;
    has_reference_frame_ = undo_.has_reference_frame;
    block_dirty_ = undo_.block_dirty;
    memcpy(block_data_, undo_.block_data, sizeof(block_data_));
    motion_forward_ = undo_.motion_forward;
    motion_backward_ = undo_.motion_backward;
    stats_ = undo_.stats;
}

void Video::Rewind()
{ // video.go:195-201
    dropLookahead();
    buf_->Rewind();
    time_ = 0;
    frames_decoded_ = 0;
    has_reference_frame_ = false;
    start_code_ = -1;
}

bool Video::decodeSequenceHeader()
{ // video.go:270-331
    const size_t max_header_size = 64 + 2 * 64 * 8;
    if (!buf_->has(max_header_size))
        return false;
    width_ = buf_->read(12);
    height_ = buf_->read(12);
    if (width_ <= 0 || height_ <= 0)
        return false;
    buf_->read(4); // aspect ratio: not used by the decode path
    frame_rate_ = kPictureRate[buf_->read(4)];
    buf_->read(18); // bit rate
    buf_->skip(1 + 10 + 1);
    if (buf_->read1()) {
        for (int i = 0; i < 64; i++)
            intra_quant_[kZigZag[i]] = (uint8_t)buf_->read(8);
    } else {
        memcpy(intra_quant_, kDefaultIntraQuant, 64);
    }
    if (buf_->read1()) {
        for (int i = 0; i < 64; i++)
            non_intra_quant_[kZigZag[i]] = (uint8_t)buf_->read(8);
    } else {
        memset(non_intra_quant_, 16, 64); // video.go:1066-1075
    }
    mb_width_ = (width_ + 15) >> 4;
    mb_height_ = (height_ + 15) >> 4;
    mb_size_ = mb_width_ * mb_height_;
    luma_width_ = mb_width_ << 4;
    luma_height_ = mb_height_ << 4;
    chroma_width_ = mb_width_ << 3;
    chroma_height_ = mb_height_ << 3;
    {   // emitPrediction's bounds (video_noasm.go:48-50), once per sequence instead of once per macroblock
        const int64_t luma = (int64_t)luma_width_ * luma_height_, chroma = (int64_t)chroma_width_ * chroma_height_;
        range_total_ = luma + 2 * chroma + (int64_t)luma_width_ * 16;
        range_chroma_ = range_total_ - luma - chroma;
    }

    // initFrame x3 (video.go:324-326): the three slots live in the backend's frame store
    backend_->open(width_, height_);
    backend_->setQuant(intra_quant_, non_intra_quant_);
    luma_bytes_ = (size_t)luma_width_ * (size_t)luma_height_;
    chroma_bytes_ = (size_t)chroma_width_ * (size_t)chroma_height_;
    for (int s = 0; s < 3; s++) {
        host_planes_[s].assign(luma_bytes_ + 2 * chroma_bytes_, 0);
        Frame &f = frames_[s];
        f.owner = this;
        f.slot = (uint32_t)s;
        f.Width = width_;
        f.Height = height_;
        f.Y = Plane{luma_width_, luma_height_, host_planes_[s].data(), luma_bytes_};
        f.Cb = Plane{chroma_width_, chroma_height_, host_planes_[s].data() + luma_bytes_, chroma_bytes_};
        f.Cr = Plane{chroma_width_, chroma_height_, host_planes_[s].data() + luma_bytes_ + chroma_bytes_, chroma_bytes_};
    }
    for (int i = 0; i < 2; i++) { // the two frames Decode alternates between (video.go:209-268: "valid until the next call")
        if (out_planes_[i])
            backend_->freePlanes(out_planes_[i]);
        out_planes_[i] = backend_->allocPlanes(luma_bytes_ + 2 * chroma_bytes_);
        if (!out_planes_[i])
            throw std::bad_alloc();
        Frame &f = out_frames_[i];
        f.owner = this;
        f.Width = width_;
        f.Height = height_;
        f.Y = Plane{luma_width_, luma_height_, out_planes_[i], luma_bytes_};
        f.Cb = Plane{chroma_width_, chroma_height_, out_planes_[i] + luma_bytes_, chroma_bytes_};
        f.Cr = Plane{chroma_width_, chroma_height_, out_planes_[i] + luma_bytes_ + chroma_bytes_, chroma_bytes_};
    }
    host_rgba_.assign((size_t)width_ * (size_t)height_ * 4, 0);
    written_.assign((size_t)mb_size_, 0);
    slot_cur_ = 0;
    slot_fwd_ = 1;
    slot_bwd_ = 2;
    has_sequence_header_ = true;
    return true;
}

Frame *Video::frameForSlot(uint32_t slot)
{
    // Frame.Y/Cb/Cr.Data are host-visible: fetch the slot's planes (synchronises with the device)
    uint8_t *base = host_planes_[slot].data();
    const auto t0 = std::chrono::steady_clock::now();
    backend_->readPlanes(slot, base, base + luma_bytes_, base + luma_bytes_ + chroma_bytes_);
    stats_.seconds_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return &frames_[slot];
}

const uint8_t *Video::fetchRGBA(uint32_t slot)
{ // Frame.RGBA, video.go:31-36
    backend_->readRGBA(slot, host_rgba_.data());
    return host_rgba_.data();
}

const uint8_t *Frame::RGBA() { return owner->fetchRGBA(slot); }

Frame *Video::Decode()
{ // video.go:209-268, one picture ahead on the host (mpeg.hpp)
    uint32_t slot;
    double t;
    ahead_tried_ = false;
    replayDeferred(); // what the previous call parsed ahead goes to the device now
    if (ahead_.valid) {
        slot = ahead_.slot;
        t = ahead_.time;
        ahead_.valid = false;
    } else if (!DecodeDeferred(&slot, &t)) {
        return nullptr;
    }
    // the frame: the slot's copy in the backend's host mirror, which the reconstruction launches write themselves, or a read-back
    // queued behind the picture that completes it ...
    const int b = out_next_;
    out_next_ ^= 1;
    auto t0 = std::chrono::steady_clock::now();
    uint64_t ticket = 0;
    const uint8_t *planes = host_mirror_ ? backend_->mirrorAsync(slot, &ticket) : nullptr;
    if (!planes) {
        planes = out_planes_[b];
        ticket = backend_->readPlanesAsync(slot, out_planes_[b], luma_bytes_, chroma_bytes_);
    }
    stats_.seconds_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // ... the next picture's parse while the device works (its hand-over waits for the next call) ...
    if (lookahead_)
        parseAhead();
    // ... and only then the wait
    t0 = std::chrono::steady_clock::now();
    backend_->readWait(ticket);
    stats_.seconds_read += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    Frame *f = &out_frames_[b];
    f->slot = slot;
    f->Time = t;
    f->Y.Data = planes;
    f->Cb.Data = planes + luma_bytes_;
    f->Cr.Data = planes + luma_bytes_ + chroma_bytes_;
    return f;
}

void Video::parseAhead()
{
    undo_ = Undo{slot_cur_, slot_fwd_, slot_bwd_, picture_type_, has_reference_frame_, block_dirty_, {}, motion_forward_, motion_backward_, stats_};
    memcpy(undo_.block_data, block_data_, sizeof(block_data_));
    undo_valid_ = true;
    ended_before_ahead_ = buf_->HasEnded();
    ahead_tried_ = true;
    defer_submits_ = true;
    try {
        ahead_.valid = DecodeDeferred(&ahead_.slot, &ahead_.time);
    } catch (...) {
        defer_submits_ = false;
        throw;
    }
    defer_submits_ = false;
    if (!ahead_.valid && n_deferred_ == 0)
        undo_valid_ = false; // nothing was consumed that a Rewind would have to give back
}

void Video::replayDeferred()
{
    undo_valid_ = false;
    for (size_t i = 0; i < n_deferred_; i++) {
        Deferred &d = deferred_[i];
        const auto t0 = std::chrono::steady_clock::now();
        backend_->submitOwned(d.pic, d.mbs, d.coefs);
        stats_.seconds_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        d.mbs.clear();
        d.coefs.clear();
    }
    n_deferred_ = 0;
}

bool Video::DecodeDeferred(uint32_t *slot, double *time)
{
    int r;
    while ((r = DecodeStep(slot, time)) == 2) {
    }
    return r == 1;
}

int Video::DecodeStep(uint32_t *slot, double *time)
{ // one iteration of the loop of video.go:209-268
    if (!HasHeader())
        return 0;
    int out_slot = -1;
    if (start_code_ != kStartPicture) {
        start_code_ = buf_->findStartCode(kStartPicture);
        if (start_code_ == -1) {
            if (has_reference_frame_ && !assume_no_b_frames_ && buf_->HasEnded() &&
                (picture_type_ == kPictureTypeIntra || picture_type_ == kPictureTypePredictive)) {
                has_reference_frame_ = false;
                out_slot = (int)slot_bwd_;
            } else {
                return 0;
            }
        }
    }
    if (out_slot < 0) {
        if (buf_->hasStartCode(kStartPicture) == -1 && !buf_->HasEnded())
            return 0;
        buf_->discardReadBytes();

        {   // (the picture's parse = this call minus the hand-overs it makes: flushSubmit keeps their time apart)
            const auto t0 = std::chrono::steady_clock::now();
            const double submits_before = stats_.seconds_submit;
            decodePicture();
            stats_.seconds_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() -
                                    (stats_.seconds_submit - submits_before);
        }

        if (assume_no_b_frames_)
            out_slot = (int)slot_bwd_;
        else if (picture_type_ == kPictureTypeB)
            out_slot = (int)slot_cur_;
        else if (has_reference_frame_)
            out_slot = (int)slot_fwd_;
        else
            has_reference_frame_ = true;
        if (out_slot < 0)
            return 2;
    }
    *slot = (uint32_t)out_slot;
    *time = time_;
    frames_decoded_++;
    time_ = (double)frames_decoded_ / frame_rate_;
    return 1;
}

Frame *Video::Fetch(uint32_t slot, double time, bool read_back)
{
    Frame *frame = read_back ? frameForSlot(slot) : &frames_[slot];
    frame->Time = time;
    return frame;
}

void Video::decodePicture()
{ // video.go:374-434
    buf_->skip(10);
    picture_type_ = buf_->read(3);
    buf_->skip(16);
    if (picture_type_ <= 0 || picture_type_ > kPictureTypeB)
        return;
    if (picture_type_ == kPictureTypePredictive || picture_type_ == kPictureTypeB) {
        motion_forward_.FullPx = buf_->read1();
        int f_code = buf_->read(3);
        if (f_code == 0)
            return;
        motion_forward_.RSize = f_code - 1;
    }
    if (picture_type_ == kPictureTypeB) {
        motion_backward_.FullPx = buf_->read1();
        int f_code = buf_->read(3);
        if (f_code == 0)
            return;
        motion_backward_.RSize = f_code - 1;
    }
    stats_.pictures++;

    const uint32_t slot_temp = slot_fwd_;
    if (picture_type_ == kPictureTypeIntra || picture_type_ == kPictureTypePredictive)
        slot_fwd_ = slot_bwd_;

    mbs_.clear();
    coefs_.clear();
    coef_len_ = mb_pending_ = 0;
    std::fill(written_.begin(), written_.end(), 0);
    sparse_ = sparse_wanted_; // the hand-over form is latched per picture: its offsets count either units or dwords

    do {
        start_code_ = buf_->nextStartCode();
    } while (start_code_ == kStartExtension || start_code_ == kStartUserData);

    while (start_code_ >= kStartSliceFirst && start_code_ <= kStartSliceLast) {
        decodeSlice(start_code_ & 0xFF);
        if (macroblock_address_ >= mb_size_ - 2)
            break;
        start_code_ = buf_->nextStartCode();
    }
    flushSubmit();

    if (picture_type_ == kPictureTypeIntra || picture_type_ == kPictureTypePredictive) {
        slot_bwd_ = slot_cur_;
        slot_cur_ = slot_temp;
    }
}

void Video::flushSubmit()
{
    if (mbs_.empty())
        return;
    mpeghip_pic_desc pic;
    memset(&pic, 0, sizeof(pic));
    pic.stream = 0;
    pic.cur = (uint8_t)slot_cur_;
    pic.fwd = (uint8_t)slot_fwd_;
    pic.bwd = (uint8_t)slot_bwd_;
    pic.mb_first = 0;
    pic.mb_count = (uint32_t)mbs_.size();
    pic.flags = sparse_ ? MPEGHIP_PIC_SPARSE : 0;
    const size_t n_mbs = mbs_.size();
    coefs_.resize(coef_len_); // (it ran ahead of the recording: coefRoom)
    if (defer_submits_) { // a picture parsed ahead: its hand-over is kept (the arrays change places with the kept slot's empty ones)
        if (n_deferred_ == deferred_.size())
            deferred_.emplace_back();
        Deferred &d = deferred_[n_deferred_++];
        d.pic = pic;
        d.mbs.swap(mbs_);
        d.coefs.swap(coefs_);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        backend_->submitOwned(pic, mbs_, coefs_); // (may swap the arrays for others)
        stats_.seconds_submit += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    stats_.submits++;
    stats_.macroblocks += n_mbs;
    mbs_.clear();
    coefs_.clear();
    coef_len_ = mb_pending_ = 0;
    std::fill(written_.begin(), written_.end(), 0);
}

void Video::decodeSlice(int slice)
{ // video.go:436-460
    slice_begin_ = true;
    macroblock_address_ = (slice - 1) * mb_width_ - 1;
    motion_backward_.H = motion_forward_.H = 0;
    motion_backward_.V = motion_forward_.V = 0;
    dc_predictor_[0] = dc_predictor_[1] = dc_predictor_[2] = 128;
    quantizer_scale_ = buf_->read(5);
    while (buf_->read1())
        buf_->skip(8);
    // The bit cursor of the slice, in locals (Cursor): nothing inside a macroblock refills the buffer, and between macroblocks
    // only peekNonZero's has() may — when fewer than 23 bits are left of what is loaded; then the cursor goes back to the buffer,
    // the buffer does what the reference does (buffer.go:341-350, 203-221: load more, or notice the end), and the cursor is read anew.
    Cursor c{buf_->Bytes(), buf_->Len(), buf_->bitIndex()};
    for (;;) {
        decodeMacroblock(c);
        if (macroblock_address_ >= mb_size_ - 1)
            break;
        if (__builtin_expect((c.len << 3) >= c.bit + 23, 1)) {
            if ((c.window() >> (64 - 23)) == 0)
                break;
            continue;
        }
        buf_->setBitIndex(c.bit);
        const bool more = buf_->peekNonZero(23);
        c = Cursor{buf_->Bytes(), buf_->Len(), buf_->bitIndex()};
        if (!more)
            break;
    }
    buf_->setBitIndex(c.bit);
}

uint8_t *Video::coefRoom(size_t bytes)
{
    const size_t at = coef_len_ + mb_pending_;
    if (coefs_.size() < at + bytes)
        coefs_.resize(at + bytes + (256u << 10)); // a picture grows its array a few times, not once per block
    return coefs_.data() + at;
}

uint8_t *Video::coefAppend(size_t bytes)
{
    mb_pending_ = 0;
    uint8_t *p = coefRoom(bytes);
    memset(p, 0, bytes);
    coef_len_ += bytes;
    return p;
}

void Video::beginMacroblockRecord(bool intra)
{
    // A damaged stream can address a macroblock twice in one picture; the reference
    // simply executes both in bitstream order.  Macroblocks of one submit run
    // concurrently on the device, so the earlier ones are flushed first.
    const size_t addr = (size_t)macroblock_address_; // (= mb_row_ * mb_width_ + mb_col_: decodeMacroblock keeps them in step)
    if (written_[addr]) {
        flushSubmit();
        stats_.duplicate_splits++;
    }
    written_[addr] = 1;
    // (not `rec_ = MbRec()`: that clears 2.3 KB of block storage per macroblock; a block is reset by its own
    // decodeBlock, and endMacroblockRecord only looks at the blocks of this macroblock's pattern)
    rec_.has_pred = rec_.backward = rec_.any_raw = rec_.out_of_range = false;
    rec_.valid_cbp = 0;
    rec_.mv_x = rec_.mv_y = 0;
    rec_.cbp = 0;
    rec_.active = true;
    rec_.intra = intra;
    rec_.mb_x = mb_col_;
    rec_.mb_y = mb_row_;
    rec_.qscale = quantizer_scale_;
    mb_pending_ = 0;
    coefRoom(6 * 66 * 4); // six blocks of a count word and 65 pairs at most: decodeBlock's pointers stay valid for the macroblock
}

void Video::emitPrediction(int mh, int mv, bool backward)
{
    // copyMacroblock(mh, mv, mbRow, mbCol, ..., src, &frameCurrent): a later call overwrites an earlier one
    rec_.has_pred = true;
    rec_.backward = backward;
    rec_.mv_x = mh;
    rec_.mv_y = mv;
    // legal read range of copyMacroblock (video_noasm.go:48-50): [plane start, end of base).  Outside it the reference
    // panics; here the whole macroblock is dropped (endMacroblockRecord) — also when the call that would panic is one a
    // later call overwrites (a B macroblock's forward copy)
    const int64_t lw = luma_width_, cw = chroma_width_;
    const int64_t total = range_total_, chroma_total = range_chroma_; // (luma + 2 chroma + 16 rows of padding; chroma + the padding)
    const int64_t lsi = ((int64_t)(rec_.mb_y << 4) + (mv >> 1)) * lw + (rec_.mb_x << 4) + (mh >> 1);
    const int64_t llast = lsi + (15 + (mv & 1)) * lw + 15 + (mh & 1);
    const int cmh = mh / 2, cmv = mv / 2;
    const int64_t csi = ((int64_t)(rec_.mb_y << 3) + (cmv >> 1)) * cw + (rec_.mb_x << 3) + (cmh >> 1);
    const int64_t clast = csi + (7 + (cmv & 1)) * cw + 7 + (cmh & 1);
    if (lsi < 0 || llast >= total || csi < 0 || clast >= chroma_total)
        rec_.out_of_range = true;
}

void Video::endMacroblockRecord()
{
    if (!rec_.active)
        return;
    rec_.active = false;
    if (!rec_.intra && !rec_.has_pred)
        return; // cannot happen: every non-intra macroblock is predicted (video.go:543-544)

    if (!rec_.intra && rec_.out_of_range) {
        stats_.range_skips++; // the reference panics here; the macroblock is dropped instead
        return;
    }

    bool raw = rec_.any_raw;              // (kept by decodeBlock as the blocks end)
    const int cbp = rec_.valid_cbp & rec_.cbp;
    mpeghip_mb_desc d;
    memset(&d, 0, sizeof(d));
    d.pic = 0;
    d.mb_x = (uint16_t)rec_.mb_x;
    d.mb_y = (uint16_t)rec_.mb_y;
    d.mv_x = (int16_t)rec_.mv_x;
    d.mv_y = (int16_t)rec_.mv_y;
    d.flags = (uint8_t)(rec_.intra ? MPEGHIP_MB_INTRA : (rec_.backward ? MPEGHIP_MB_REF_BWD : MPEGHIP_MB_REF_FWD));
    if (raw)
        d.flags |= MPEGHIP_MB_COEF_RAW;
    d.cbp = (uint8_t)cbp;
    d.qscale = (uint8_t)(rec_.qscale < 1 ? 1 : (rec_.qscale > 31 ? 31 : rec_.qscale));
    d.coef_off = (uint32_t)(coef_len_ / (sparse_ ? 4 : MPEGHIP_COEF_UNIT)); // sparse pictures count dwords
    if (rec_.qscale < 1 && cbp)
        raw = true, d.flags |= MPEGHIP_MB_COEF_RAW; // quantiser_scale 0 (forbidden value): keep the reference's arithmetic

    stats_.coded_blocks += (uint64_t)__builtin_popcount((unsigned)cbp);
    if (!raw && sparse_) {
        // The blocks are where the VLC loop wrote them: a count, then one pair per coded level (an intra block's DC first) — a
        // device entry short of the bits the library's packer adds; a coded zero level stays a pair.  Every block that
        // advanced mb_pending_ is valid and clean, and those are exactly the blocks of `cbp` here.
        coef_len_ += mb_pending_;
        mb_pending_ = 0;
        mbs_.push_back(d);
        return;
    }
    // The rare forms.  The clean blocks' pair words may sit in coefs_ right where this macroblock's bytes are about to go
    // (a sparse picture's raw macroblock): they are read from a copy.
    uint32_t kept[6][66];
    const uint32_t *pairs_of[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int b = 0; b < 6; b++)
        if ((cbp & (0x20 >> b)) && !rec_.blocks[b].needs_raw) {
            memcpy(kept[b], rec_.blocks[b].pairs, 4 * (size_t)rec_.blocks[b].n_pairs);
            pairs_of[b] = kept[b];
        }
    for (int b = 0; b < 6; b++) {
        if (!(cbp & (0x20 >> b)))
            continue;
        const BlockRec &br = rec_.blocks[b];
        if (raw) {
            int32_t snap[64];
            if (br.needs_raw) {
                memcpy(snap, br.raw, sizeof(snap));
            } else { // this block was clean: dequantise it here so the whole macroblock shares one format
                const uint8_t *qm = rec_.intra ? intra_quant_ : non_intra_quant_;
                memset(snap, 0, sizeof(snap));
                for (int k = 0; k < br.n_pairs; k++) { // the coded levels — a coded ZERO among them dequantises to +-1
                    const int i = naturalIndexOfPair(pairs_of[b][k]);
                    snap[i] = dequantPremult(levelOfPair(pairs_of[b][k]), rec_.intra, rec_.qscale, qm[i], i);
                }
                if (rec_.intra)
                    snap[0] = (int32_t)br.dc * 256;
            }
            if (sparse_) { // every block of the sparse form begins with its count word: 64 for a snapshot
                const uint32_t n = 64;
                memcpy(coefAppend(4), &n, 4);
            }
            int32_t *dst = reinterpret_cast<int32_t *>(coefAppend(2 * MPEGHIP_COEF_UNIT));
            for (int r = 0; r < 8; r++)
                for (int c = 0; c < 8; c++)
                    dst[c * 8 + r] = snap[r * 8 + c]; // column-major
        } else {
            // column-major (coefAppend zero-filled the unit): only the positions the block's levels went to
            int16_t *dst = reinterpret_cast<int16_t *>(coefAppend(MPEGHIP_COEF_UNIT));
            if (rec_.intra)
                dst[0] = br.dc;
            for (int k = 0; k < br.n_pairs; k++)
                dst[(pairs_of[b][k] >> 2) & 63] = (int16_t)levelOfPair(pairs_of[b][k]);
        }
    }
    mb_pending_ = 0;
    if (raw)
        stats_.raw_macroblocks++;
    mbs_.push_back(d);
}

void Video::decodeMacroblock(Cursor &c)
{ // video.go:462-562
    int increment = 0;
    int t = vlcAt(tabMba(), c);
    while (t == 34)
        t = vlcAt(tabMba(), c); // macroblock_stuffing
    while (t == 35) {
        increment += 33; // macroblock_escape
        t = vlcAt(tabMba(), c);
    }
    increment += t;

    if (slice_begin_) {
        slice_begin_ = false;
        macroblock_address_ += increment;
        mb_row_ = macroblock_address_ / mb_width_;
        mb_col_ = macroblock_address_ % mb_width_;
    } else {
        if (macroblock_address_ + increment >= mb_size_)
            return; // invalid
        if (increment > 1) {
            dc_predictor_[0] = dc_predictor_[1] = dc_predictor_[2] = 128;
            if (picture_type_ == kPictureTypePredictive) {
                motion_forward_.H = 0;
                motion_forward_.V = 0;
            }
        }
        // (row and column follow the address step by step here: they were derived from it when the slice began, and a
        // division per macroblock is a tenth of a no-coefficient macroblock's parse)
        while (increment > 1) { // skipped macroblocks are predicted
            macroblock_address_++;
            if (++mb_col_ == mb_width_)
                mb_col_ = 0, mb_row_++;
            beginMacroblockRecord(false);
            predictMacroblock();
            endMacroblockRecord();
            increment--;
        }
        macroblock_address_++;
        if (++mb_col_ == mb_width_)
            mb_col_ = 0, mb_row_++;
    }
    if (mb_col_ >= mb_width_ || mb_row_ >= mb_height_ || macroblock_address_ < 0)
        return; // corrupt stream

    macroblock_type_ = vlcAt(tabType(picture_type_), c);
    macroblock_intra_ = (macroblock_type_ & 0x01) != 0;
    motion_forward_.IsSet = (macroblock_type_ & 0x08) != 0;
    motion_backward_.IsSet = (macroblock_type_ & 0x04) != 0;
    if (macroblock_type_ & 0x10)
        quantizer_scale_ = c.bits(5);

    beginMacroblockRecord(macroblock_intra_);
    if (macroblock_intra_) {
        motion_backward_.H = motion_forward_.H = 0;
        motion_backward_.V = motion_forward_.V = 0;
    } else {
        dc_predictor_[0] = dc_predictor_[1] = dc_predictor_[2] = 128;
        decodeMotionVectors(c);
        predictMacroblock();
    }

    int cbp = 0;
    if (macroblock_type_ & 0x02)
        cbp = vlcAt(tabCbp(), c);
    else if (macroblock_intra_)
        cbp = 0x3f;
    rec_.cbp = cbp;
    for (unsigned todo = (unsigned)cbp & 0x3f; todo;) { // the coded blocks, first to last (0x20 = block 0): one loop exit to predict
        const int top = 31 - __builtin_clz(todo);
        todo &= ~(1u << top);
        decodeBlock(c, 5 - top);
    }
    endMacroblockRecord();
}

void Video::decodeMotionVectors(Cursor &c)
{ // video.go:564-581
    if (motion_forward_.IsSet) {
        const int r = motion_forward_.RSize;
        motion_forward_.H = decodeMotionVector(c, r, motion_forward_.H);
        motion_forward_.V = decodeMotionVector(c, r, motion_forward_.V);
    } else if (picture_type_ == kPictureTypePredictive) {
        motion_forward_.H = 0;
        motion_forward_.V = 0;
    }
    if (motion_backward_.IsSet) {
        const int r = motion_backward_.RSize;
        motion_backward_.H = decodeMotionVector(c, r, motion_backward_.H);
        motion_backward_.V = decodeMotionVector(c, r, motion_backward_.V);
    }
}

int Video::decodeMotionVector(Cursor &c, int rSize, int motion)
{ // video.go:583-606
    const int fscale = 1 << rSize;
    // motion_code, then motion_r (rSize bits) if the code is not zero, from one look at the stream (<= 11 + 6 bits), and the
    // reference's two cases (code 0 or f_code 1: d = code; else d = sign * (((|code| - 1) << rSize) + r + 1)) as ONE expression:
    // with rSize 0 the second form is |code| itself, and a zero code is masked out — its sign and its zero-ness are coin flips
    const uint64_t w = c.window();
    const VlcTable::Symbol mc = tabMotion().at(w);
    const int m_code = mc.value;
    const int nonzero = m_code != 0;
    const int r_bits = rSize & -nonzero;
    const int r = (int)(((w << mc.len) >> 1) >> (63 - r_bits)); // (0 when r_bits is 0)
    c.bit += (size_t)(mc.len + r_bits);
    const int sign = m_code >> 31;                   // -1 / 0
    const int mag = (m_code ^ sign) - sign;          // |code|
    int d = ((mag - 1) * fscale + r + 1) & -nonzero; // ((mag - 1) is -1 for a zero code: a product, not a shift)
    d = (d ^ sign) - sign;
    motion += d;
    if (motion > (fscale << 4) - 1)
        motion -= fscale << 5;
    else if (motion < -(fscale << 4))
        motion += fscale << 5;
    return motion;
}

void Video::predictMacroblock()
{ // video.go:608-637
    int fw_h = motion_forward_.H, fw_v = motion_forward_.V;
    if (motion_forward_.FullPx) {
        fw_h *= 2; // (full-pel vectors; may be negative)
        fw_v *= 2;
    }
    if (picture_type_ == kPictureTypeB) {
        int bw_h = motion_backward_.H, bw_v = motion_backward_.V;
        if (motion_backward_.FullPx) {
            bw_h *= 2;
            bw_v *= 2;
        }
        if (motion_forward_.IsSet) {
            emitPrediction(fw_h, fw_v, false);
            if (motion_backward_.IsSet)
                emitPrediction(bw_h, bw_v, true); // overwrites the forward copy: no averaging in the reference
        } else {
            emitPrediction(bw_h, bw_v, true);
        }
    } else {
        emitPrediction(fw_h, fw_v, false);
    }
}

void Video::decodeBlock(Cursor &c, int block)
{ // video.go:639-745 (parse) + the bookkeeping that replaces :747-798
    BlockRec &br = rec_.blocks[block];
    br.valid = false;
    br.needs_raw = false;
    int n = 0;
    const uint8_t *quant_matrix;
    // block_data_ mirrors the reference's persistent blockData.  It is all zero except
    // after an invalid block (video.go:711-714 returns before the clears); only then —
    // or when this block itself ends invalid — do its exact contents matter.
    const bool dirty_at_start = block_dirty_;
    bool explicit_zero = false;
    int32_t dc256 = 0;

    if (macroblock_intra_) {
        const int plane_index = block > 3 ? block - 3 : 0;
        const int predictor = dc_predictor_[plane_index];
        // dct_dc_size and the differential from one look at the stream (<= 9 + 11 bits); a differential below half its range
        // is negative: d - (2^size - 1) = the reference's (-(1 << size)) | (d + 1), without a branch on a coin flip
        const uint64_t w0 = c.window();
        const VlcTable::Symbol ds = tabDcSize(plane_index).at(w0);
        const int dct_size = ds.value;
        int dc = predictor;
        if (dct_size > 0) {
            const int differential = (int)((w0 << ds.len) >> (64 - dct_size));
            const int negative = ((differential >> (dct_size - 1)) & 1) ^ 1;
            dc += differential - (((1 << dct_size) - 1) & -negative);
        }
        c.bit += (size_t)(ds.len + dct_size);
        dc_predictor_[plane_index] = dc;
        if (dc < -32768 || dc > 32767)
            br.needs_raw = true; // not expressible as int16: goes through the snapshot path
        br.dc = (int16_t)(dc < -32768 ? -32768 : (dc > 32767 ? 32767 : dc));
        // blockData[0] = dc << 8.  Beyond +-2^30 the pixel saturates whatever the AC terms add
        // (their sum is below 0.6 * 2^30), so clamping there is exact and keeps int32.
        int64_t v = (int64_t)dc * 256;
        if (v > (1 << 30))
            v = 1 << 30;
        if (v < -(1 << 30))
            v = -(1 << 30);
        dc256 = (int32_t)v;
        if (dirty_at_start)
            block_data_[0] = dc256;
        quant_matrix = intra_quant_;
        n = 1;
    } else {
        quant_matrix = non_intra_quant_;
    }

    // The block's words of the sparse hand-over go straight to their place in the picture's array (beginMacroblockRecord made
    // the room): count, an intra block's DC, then a pair per level from the loop below.  They only COUNT once the block turns
    // out valid and clean (mb_pending_ below); a block that does not is overwritten by the next one.
    uint32_t *const words = sparse_ ? reinterpret_cast<uint32_t *>(coefs_.data() + coef_len_ + mb_pending_) : pair_scratch_[block];
    uint32_t *const __restrict pairs = words + (macroblock_intra_ ? 2 : 1);
    int n_pairs = 0;
    br.pairs = pairs;
    br.n_pairs = 0;
    int level = 0;
    bool invalid = false;
    // Intra blocks arrive here with n == 1 (their DC is read above), non-intra blocks with n == 0: only those can begin with
    // the '1' that means run 0 / level 1 and not end_of_block (video.go:687).
    const CoeffTable &next_table = tabCoeffNext(), &first_table = n == 0 ? tabCoeffFirst() : next_table;
    const CoeffTable::Entry *const next_l1 = next_table.firstLevel(), *const next_l2 = next_table.secondLevel();
    const CoeffTable::Entry *l1 = first_table.firstLevel(), *l2 = first_table.secondLevel();
    const int rest_bits = next_table.restBits(); // (both tables are built from the same code list)
    // The cursor lives in locals for the length of the block: the stores below may alias anything.
    const uint8_t *const data = c.data;
    const size_t data_len = c.len;
    size_t bit = c.bit;
    auto window = [&]() -> uint64_t { // the next 57+ bits, left-aligned, zero-padded past the end (Cursor::window on the locals)
        const size_t byte = bit >> 3;
        uint64_t w;
        if (byte + 8 <= data_len) {
            memcpy(&w, data + byte, 8);
            w = __builtin_bswap64(w);
        } else {
            w = 0;
            for (size_t k = 0; k < 8; k++)
                w = (w << 8) | (byte + k < data_len ? data[byte + k] : 0u);
        }
        return w << (bit & 7);
    };
    // One 64-bit look at the stream serves as many symbols as fit: a table symbol is at most 18 bits with its sign, an escape
    // 6 + 6 + 16: the loop looks again below 18 valid bits, the escape below 28.  The table (CoeffTable) answers run,
    // signed level and length in one probe: the dependent chain per coefficient is shift -> table -> shift.
    uint64_t w = window();
    int valid = 64 - (int)(bit & 7);
    const CoeffPairTable::Entry *const pair_tab = tabCoeffPairs().data();
    bool two_at_once = !dirty_at_start && n != 0; // (n == 0: a non-intra block's first symbol has its own table)
    for (;;) {
        if (valid < 18) { // (a table symbol with its sign; the escape below looks again if it needs its 28)
            w = window();
            valid = 64 - (int)(bit & 7);
        }
        if (two_at_once) {
            // Two symbols from one probe where the table has them (CoeffPairTable); `total` 0 sends the probe the one-symbol way.
            const CoeffPairTable::Entry pe = pair_tab[(size_t)(w >> (64 - CoeffPairTable::kBits))];
            if (__builtin_expect(pe.total != 0, 1)) {
                n += pe.run1;
                if (__builtin_expect(n >= 64, 0)) { // video.go:711-714: the cursor stays behind the symbol that overflowed
                    bit += pe.len1;
                    invalid = true;
                    break;
                }
                pairs[n_pairs++] = ((uint32_t)(uint16_t)(int16_t)pe.level1 << 16) | kScanPos.v[n];
                n++;
                const int second = pe.flags & CoeffPairTable::kSecond; // 0 / 1: counted, not branched on
                const int n2 = n + pe.run2;
                if (__builtin_expect(second & (n2 >= 64), 0)) {
                    bit += (size_t)(pe.len1 + pe.len2);
                    invalid = true;
                    break;
                }
                pairs[n_pairs] = ((uint32_t)(uint16_t)(int16_t)pe.level2 << 16) | kScanPos.v[n2 & 63]; // (kept only if `second`)
                n_pairs += second;
                n = n2 + second; // (run2 is 0 without a second symbol)
                bit += pe.total;
                w <<= pe.total;
                valid -= pe.total;
                if (pe.flags & CoeffPairTable::kEndOfBlock)
                    break;
                continue;
            }
        }
        const CoeffTable::Entry &e = CoeffTable::at(l1, l2, rest_bits, w);
        l1 = next_l1;
        l2 = next_l2;
        two_at_once = !dirty_at_start; // (behind a block's first symbol; a dirty blockData wants every level as it comes)
        int run;
        if (__builtin_expect(e.kind == CoeffTable::kCoef, 1)) {
            run = e.run;
            level = e.level;
            bit += e.len;
            w <<= e.len;
            valid -= e.len;
        } else if (e.kind == CoeffTable::kEnd) {
            bit += e.len;
            break;
        } else if (e.kind == CoeffTable::kCoefEnd) { // the block's last coefficient with the end_of_block behind it
            bit += e.len;
            n += e.run;
            if (n >= 64) { // (the reference finds the coefficient out of range before it would read the end_of_block)
                bit -= 2;
                invalid = true;
                break;
            }
            pairs[n_pairs++] = ((uint32_t)(uint16_t)e.level << 16) | kScanPos.v[n];
            if (dirty_at_start) {
                const int dz = kZigZag[n];
                block_data_[dz] = dequantPremult(e.level, macroblock_intra_, quantizer_scale_, quant_matrix[dz], dz);
            }
            n++;
            break;
        } else if (e.kind == CoeffTable::kEscape) { // run (6 bits), level (8 bits, or 8 + 8): video.go:690-700
            if (valid < 28) {
                w = window();
                valid = 64 - (int)(bit & 7);
            }
            const uint32_t f = (uint32_t)((w << e.len) >> (64 - 22));
            run = (int)(f >> 16);
            const int b = (int)((f >> 8) & 0xff);
            int used = 14;
            if (b & 0x7f) {
                level = (int8_t)b; // -127 .. 127, without a branch on the sign (it is a coin flip)
            } else {               // 0x00 / 0x80: the level is in the next 8 bits (128 .. 255 / -255 .. -129)
                level = (int)(f & 0xff) - (b ? 256 : 0);
                used = 22;
            }
            used += e.len;
            bit += (size_t)used;
            w <<= used;
            valid -= used;
            if (level == 0)
                explicit_zero = true; // dequantises to +-1, which "0 = absent" cannot express
        } else { // kZero: a dead end of the code tree reads as run 0, level 0 (+ the sign bit)
            run = e.run;
            level = 0;
            bit += e.len;
            w <<= e.len;
            valid -= e.len;
            explicit_zero = true;
        }
        n += run;
        if (n < 0 || n >= 64) {
            invalid = true;
            break;
        }
        pairs[n_pairs++] = ((uint32_t)(uint16_t)level << 16) | kScanPos.v[n]; // = MPEGHIP_PAIR(level, position of scan index n)
        if (dirty_at_start) {
            const int dz = kZigZag[n];
            block_data_[dz] = dequantPremult(level, macroblock_intra_, quantizer_scale_, quant_matrix[dz], dz);
        }
        n++;
    }

    c.bit = bit;
    br.n_pairs = n_pairs;

    // bring block_data_ up to date when it was not maintained on the fly
    auto materialize = [&]() {
        if (dirty_at_start)
            return;
        if (macroblock_intra_)
            block_data_[0] = dc256;
        for (int k = 0; k < n_pairs; k++) {
            const int dz = naturalIndexOfPair(pairs[k]);
            block_data_[dz] = dequantPremult(levelOfPair(pairs[k]), macroblock_intra_, quantizer_scale_, quant_matrix[dz], dz);
        }
    };

    if (invalid) {
        // video.go:711-714: return without reconstructing and WITHOUT clearing blockData
        stats_.invalid_blocks++;
        materialize();
        block_dirty_ = true;
        return;
    }

    if (!dirty_at_start && (!explicit_zero || sparse_) && !br.needs_raw) { // (a coded zero level: a pair says it, a unit cannot)
        // the common case: blockData held nothing but this block, and the reference clears it
        // again after use (video.go:777, 781-783, 790, 794-796) — nothing to keep on the host
        br.valid = true;
        rec_.valid_cbp |= 0x20 >> block;
        if (sparse_) {
            const uint32_t count = (uint32_t)n_pairs + (macroblock_intra_ ? 1u : 0u);
            words[0] = count;
            if (macroblock_intra_)
                words[1] = MPEGHIP_PAIR(br.dc, 0);
            mb_pending_ += 4 * (size_t)(1 + count);
        }
        return;
    }

    // snapshot path: reproduce exactly what idct() / the DC fast path would consume
    materialize();
    br.needs_raw = true;
    if (n == 1) { // video.go:774-777 / 787-790: only blockData[0] is used, and only it is cleared
        memset(br.raw, 0, sizeof(br.raw));
        br.raw[0] = block_data_[0];
        block_data_[0] = 0;
    } else {
        if (n < 10) { // video.go:807-866: the reduced IDCT ignores rows >= 4 and columns >= 4
            for (int i = 0; i < 64; i++)
                br.raw[i] = ((i >> 3) < 4 && (i & 7) < 4) ? block_data_[i] : 0;
        } else {
            memcpy(br.raw, block_data_, sizeof(br.raw));
        }
        memset(block_data_, 0, sizeof(block_data_)); // video.go:781-783 / 794-796
    }
    block_dirty_ = false;
    for (int i = 0; i < 64; i++)
        if (block_data_[i] != 0) {
            block_dirty_ = true;
            break;
        }
    br.valid = true;
    rec_.valid_cbp |= 0x20 >> block;
    rec_.any_raw = true;
}

} // namespace mpeg
