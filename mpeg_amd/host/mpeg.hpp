// mpeg.hpp — host-side mirror of the gen2brain/mpeg API above the libmpeghip C ABI.
//
// The reference is a Go package; no Go toolchain exists in the build image, so
// the host layer is C++ with the reference's names, argument meaning and error
// behaviour (the Go package + cgo shim a maintainer would ship is in go/, see
// INTEGRATION.md).  The serial work stays on the CPU exactly as in the
// reference — program-stream demux (demux.go), bit buffer (buffer.go), MPEG-1
// video VLC parse (video.go:209-745) and MP2 frame parse (audio.go:163-490) —
// but instead of reconstructing pixels / samples inline, the parsers RECORD
// macroblock descriptors / sub-band samples and hand one whole picture / audio
// frame to the GPU through mpeghip_video_submit / mpeghip_audio_synth.
//
// There is no CPU reconstruction path here: without a gfx950 device the
// constructors fail (they need a mpeghip context).
#pragma once

#include <stddef.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <new>
#include <utility>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <exception>

#include "mpeghip.h"

namespace mpeg {

// A picture's coefficient bytes: a vector whose resize() does not zero what it adds (the parser writes every byte it hands over,
// and grows the array in large steps ahead of what it has recorded: zero-filling those steps cost it a memset of every picture).
template <class T>
struct NoInitAllocator : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAllocator<U>; };
    NoInitAllocator() = default;
    template <class U> NoInitAllocator(const NoInitAllocator<U> &) {}
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        if constexpr (sizeof...(Args) == 0)
            ::new (static_cast<void *>(p)) U; // default-initialised: nothing is written
        else
            ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};
using CoefBytes = std::vector<uint8_t, NoInitAllocator<uint8_t>>;


// ------------------------------------------------------------------ buffer.go
class Buffer;
using LoadFunc = std::function<void(Buffer *)>; // buffer.go:14

// minimal io.ReadSeeker for Buffer
struct BufferReader {
    std::function<size_t(uint8_t *, size_t)> read; // returns bytes read, 0 = EOF
    std::function<bool(size_t)> seek;              // absolute; may be empty (not seekable)
    std::function<size_t()> tell;                  // current offset (io.Seeker's Seek(0, SeekCurrent)); with seek
    size_t size = 0;                               // total size if seekable
};

// buffer.go:17-221.  Data source of all decoders: a growable byte buffer with a
// bit cursor, fed by Write() or on demand through the load callback.
class Buffer {
public:
    // NewBuffer (buffer.go:32-61).  `reader` may be empty (push mode: Write()).
    using Reader = BufferReader;
    explicit Buffer(Reader reader = Reader());
    static std::unique_ptr<Buffer> FromMemory(const uint8_t *data, size_t len); // bytes.NewReader + LoadReaderCallback

    static size_t BufferSize;            // buffer.go:10 (default 128 KiB)

    const uint8_t *Bytes() const { return bytes_.data(); }
    size_t Len() const { return bytes_.size(); }
    size_t Index() const { return bit_index_ >> 3; }            // buffer.go:69
    bool Seekable() const { return has_reader_ && total_size_ > 0; }
    size_t Write(const uint8_t *p, size_t n);                   // buffer.go:79-89
    void SignalEnd() { total_size_ = bytes_.size(); }           // buffer.go:94-96
    void SetLoadCallback(LoadFunc cb) { load_ = std::move(cb); }
    void Rewind() { seek(0); }                                  // buffer.go:105-107
    size_t Size() const { return total_size_ > 0 ? total_size_ : bytes_.size(); }
    size_t Remaining() const { return bytes_.size() - (bit_index_ >> 3); }
    bool HasEnded() const { return has_ended_; }
    void LoadReaderCallback(Buffer *);                           // buffer.go:131-156

    // package-private in the reference; used by the decoders
    void seek(size_t pos);
    size_t tell();
    void discardReadBytes();
    bool has(size_t count);
    // next `count` (<= 24) bits, zero-padded past the end (the reference would index out of range and panic
    // there).  Inline: the parser spends its time here.
    uint32_t peek(int count) const
    {
        const size_t byte = bit_index_ >> 3, n = bytes_.size();
        uint32_t w;
        if (byte + 4 <= n) {
            memcpy(&w, bytes_.data() + byte, 4);
            w = __builtin_bswap32(w);
        } else {
            w = 0;
            for (size_t k = 0; k < 4; k++)
                w = (w << 8) | (byte + k < n ? bytes_[byte + k] : 0u);
        }
        w <<= (bit_index_ & 7);
        return count ? w >> (32 - count) : 0;
    }
    // the next 57+ bits, left-aligned in 64, zero-padded past the end: several fields per look
    uint64_t window() const
    {
        const size_t byte = bit_index_ >> 3, n = bytes_.size();
        uint64_t w;
        if (byte + 8 <= n) {
            memcpy(&w, bytes_.data() + byte, 8);
            w = __builtin_bswap64(w);
        } else {
            w = 0;
            for (size_t k = 0; k < 8; k++)
                w = (w << 8) | (byte + k < n ? bytes_[byte + k] : 0u);
        }
        return w << (bit_index_ & 7);
    }
    int read(int count)
    { // buffer.go:223-244
        int value = 0;
        while (count > 0) {
            const int take = count > 16 ? 16 : count;
            value = (value << take) | (int)peek(take);
            bit_index_ += (size_t)take;
            count -= take;
        }
        return value;
    }
    int read1()
    { // buffer.go:246-255
        const int v = (int)peek(1);
        bit_index_++;
        return v;
    }
    void drop(int count) { bit_index_ += (size_t)count; }
    void align() { bit_index_ = ((bit_index_ + 7) >> 3) << 3; }
    void skip(size_t count);
    int skipBytes(uint8_t v);
    int nextStartCode();
    int findStartCode(int code);
    int hasStartCode(int code);
    bool findFrameSync();
    bool peekNonZero(int bitCount);
    size_t bitIndex() const { return bit_index_; }
    void setBitIndex(size_t b) { bit_index_ = b; }

private:
    Reader reader_;
    bool has_reader_ = false;
    std::vector<uint8_t> bytes_;
    size_t bit_index_ = 0;
    size_t total_size_ = 0;
    bool has_ended_ = false;
    bool discard_read_ = true;
    std::vector<uint8_t> available_;
    LoadFunc load_;
};

// -------------------------------------------------------------------- video.go
struct Plane {                // video.go:50-54
    int Width = 0, Height = 0;
    const uint8_t *Data = nullptr;   // host copy (pinned), valid until the next Decode
    size_t Len = 0;
};

class Video;
struct Frame {                // video.go:11-23
    double Time = 0;
    int Width = 0, Height = 0;
    Plane Y, Cb, Cr;
    // Frame.RGBA (video.go:31-36): width*height*4 bytes, stride 4*width, computed on
    // the device.  Valid until the next Decode.
    const uint8_t *RGBA();
    Video *owner = nullptr;
    uint32_t slot = 0;
};

struct VideoStats {
    uint64_t pictures = 0, submits = 0, macroblocks = 0, coded_blocks = 0, raw_macroblocks = 0;
    uint64_t invalid_blocks = 0, duplicate_splits = 0, range_skips = 0;
    // wall time of a lone Video's three host phases (bench.py's single-stream legs say where a picture's microseconds go): the
    // bitstream parse of its pictures, the hand-over of their work (VideoBackend::submit*), waiting for / copying frames back
    double seconds_parse = 0, seconds_submit = 0, seconds_read = 0;
};

// What the decoders need from the reconstruction device.  The product ships exactly
// one implementation, HipVideoBackend / HipAudioBackend over libmpeghip (hip_backend.cpp);
// there is no CPU implementation in this library.  The interface exists so that the
// parser can be unit-tested without a GPU by injecting the test-only lane emulator
// (tests/host_emu), never as a fallback.
class VideoBackend {
public:
    virtual ~VideoBackend() {}
    virtual void open(int width, int height) = 0;            // (re)create the 3-slot frame store, zeroed
    virtual void setQuant(const uint8_t intra[64], const uint8_t non_intra[64]) = 0;
    virtual void submit(const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                        const uint8_t *coefs, size_t coef_bytes) = 0;
    // The same submit, with the parser's own arrays handed over: a backend that keeps the picture for later may
    // swap them for arrays of its own instead of copying (the parser clears whatever it gets back).
    virtual void submitOwned(const mpeghip_pic_desc &pic, std::vector<mpeghip_mb_desc> &mbs, CoefBytes &coefs)
    {
        submit(pic, mbs.data(), (uint32_t)mbs.size(), coefs.data(), coefs.size());
    }
    virtual void readPlanes(uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) = 0;
    virtual void readRGBA(uint32_t slot, uint8_t *dst) = 0;  // Frame.RGBA of the slot
    // Video::Decode works one picture ahead (round 6): the frame it returns is read back ASYNCHRONOUSLY into one of two buffers
    // of allocPlanes() — luma | Cb | Cr, linear — while the next picture is parsed; readWait(ticket) blocks until it is there.
    // The defaults make any backend work (the read-back happens at once): the HIP backend hands out pinned memory and queues
    // mpeghip_video_read_planes_async.
    virtual uint8_t *allocPlanes(size_t bytes) { return static_cast<uint8_t *>(calloc(bytes ? bytes : 1, 1)); }
    virtual void freePlanes(uint8_t *p) { free(p); }
    virtual uint64_t readPlanesAsync(uint32_t slot, uint8_t *dst, size_t luma_bytes, size_t chroma_bytes)
    {
        readPlanes(slot, dst, dst + luma_bytes, dst + luma_bytes + chroma_bytes);
        return 0;
    }
    virtual void readWait(uint64_t ticket) { (void)ticket; }
    // The HOST MIRROR (mpeghip_video_host_mirror): a backend whose reconstruction writes every frame once more, linearly, into host
    // memory hands out THAT copy of the slot — no read-back is queued at all; readWait(*ticket) blocks until the copy holds the slot
    // as it is after everything submitted so far.  It stays as it is until the next picture is reconstructed into the slot.
    // nullptr: no such copy (the default; the HIP backend with the mirror switched off) — Decode reads back as above.
    virtual const uint8_t *mirrorAsync(uint32_t slot, uint64_t *ticket)
    {
        (void)slot;
        (void)ticket;
        return nullptr;
    }
    virtual void setMirror(bool on) { (void)on; }
    // From how many macroblocks on a hand-over of the sparse form is validated and packed by the DEVICE (a device-packed stage of one
    // picture; 0 = never).  Backends without such a stage ignore it.
    virtual void setDevicePackFrom(uint32_t n_mbs) { (void)n_mbs; }
};

// Frame stores of MANY streams of one picture size behind one reconstruction call (libmpeghip's
// mpeghip_video with n_streams > 1): what VideoBatch drives.  Same rule as VideoBackend: the product
// ships one implementation (hip_backend.cpp), tests inject the lane emulator.
class BatchStore {
public:
    virtual ~BatchStore() {}
    virtual void open(int width, int height, uint32_t n_streams) = 0;
    virtual void setQuant(uint32_t stream, const uint8_t intra[64], const uint8_t non_intra[64]) = 0;
    virtual void submit(const mpeghip_pic_desc *pics, uint32_t n_pics, const mpeghip_mb_desc *mbs, uint32_t n_mbs,
                        const uint8_t *coefs, size_t coef_bytes) = 0;
    virtual void readPlanes(uint32_t stream, uint32_t slot, uint8_t *y, uint8_t *cb, uint8_t *cr) = 0;
    virtual void readRGBA(uint32_t stream, uint32_t slot, uint8_t *dst) = 0;
    // One submit assembled picture by picture (mpeghip_video_stage_*): stagePut may be called from several
    // threads for distinct i; each picture brings its own arrays (coef_off relative to its coefs, pic.stream
    // set).  Stores without it return false from canStage() and get one merged submit() instead.
    // device_pack: every picture of the stage is in the sparse form and the store may have the DEVICE validate and pack them
    // (mpeghip_video_stage_begin_device: the puts only copy; errors are deferred to the next sync()).
    virtual bool canStage() const { return false; }
    virtual void stageBegin(const std::vector<uint32_t> &, const std::vector<size_t> &, bool device_pack = false) { (void)device_pack; }
    virtual void stagePut(uint32_t, const mpeghip_pic_desc &, const mpeghip_mb_desc *, const uint8_t *) {}
    virtual void stageCommit() {}
    virtual void sync() {}                          // wait for queued work; throws a device-packed commit's deferred error
    // wait until the device-packed commits so far are VALIDATED (not reconstructed); throws their deferred error — a refusal is per
    // picture: refusedStreams() then names the streams whose picture was refused (the commit's other pictures were reconstructed)
    virtual void verdict() { sync(); }
    virtual std::vector<uint32_t> refusedStreams() { return {}; }
};

class AudioBackend {
public:
    virtual ~AudioBackend() {}
    // one frame: samples int32 [2][36][32] -> 2304 elements of the format's type in `out`; for
    // MPEGHIP_AUDIO_F32NLR the two halves go to `out` (left, 1152 floats) and `out2` (right).  The backend
    // may complete the writes later, but before anything reads them (AudioBatch: at its Flush()).
    virtual void synth(const int32_t *samples, int format, void *out, void *out2) = 0;
    // Audio::Decode works one frame ahead (round 6): frame N's samples — in one of two buffers of allocSamples() — are handed
    // over with synthAsync, frame N + 1 is parsed, synthWait(ticket, out, out2) blocks until frame N's output is in `out` (/
    // `out2`, as for synth).  The defaults make any backend work (the synthesis runs inside synthWait): the HIP backend hands out
    // pinned memory and queues mpeghip_audio_synth_async.
    virtual int32_t *allocSamples() { return static_cast<int32_t *>(calloc(2 * 36 * 32, sizeof(int32_t))); }
    virtual void freeSamples(int32_t *p) { free(p); }
    virtual uint64_t synthAsync(const int32_t *samples, int format)
    {
        held_samples_ = samples;
        held_format_ = format;
        return 0;
    }
    virtual void synthWait(uint64_t ticket, void *out, void *out2)
    {
        (void)ticket;
        synth(held_samples_, held_format_, out, out2);
    }

private:
    const int32_t *held_samples_ = nullptr;
    int held_format_ = 0;
};

// Synthesis state of MANY streams behind one call (libmpeghip's mpeghip_audio with n_streams > 1):
// what AudioBatch drives.  active[i] == 0: stream i has no frame in this call and keeps its state.
class AudioBatchStore {
public:
    virtual ~AudioBatchStore() {}
    virtual void open(uint32_t n_streams, int fma_mode) = 0;
    virtual void synth(const int32_t *samples, const uint8_t *active, int format, void *out) = 0; // [n][2][36][32] -> [n][2304]
    // The batch's two host arrays (what the streams record into, what the samples come back in): a device store hands out
    // page-locked memory (mpeghip_pinned_alloc), so that the two copies of a tick run at the link's rate and not through the
    // runtime's bounce buffers.  Zeroed; freed by the batch before the store goes.
    virtual void *allocHost(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
    virtual void freeHost(void *p) { free(p); }
};

// Bind the calling thread to the cores of host NUMA node `node` (sysfs cpulist); false if the node is unknown or the call fails.
bool pinThisThreadToNode(int node);

// Shared device context for decoders (one per GPU).
class Device {
public:
    explicit Device(int ordinal = 0);   // throws std::runtime_error without a gfx950 GPU
    ~Device();
    mpeghip_ctx *ctx() const { return ctx_; }
    int NumaNode() const;               // host NUMA node the GPU is attached to, -1 if unknown (mpeghip_ctx_numa_node)
    std::unique_ptr<VideoBackend> newVideoBackend();
    std::unique_ptr<AudioBackend> newAudioBackend(int fma_mode);
    std::unique_ptr<BatchStore> newBatchStore();
    std::unique_ptr<AudioBatchStore> newAudioBatchStore();
private:
    mpeghip_ctx *ctx_ = nullptr;
};

class Video {
public:
    Video(Buffer *buf, Device *dev);                 // NewVideo (video.go:110-121)
    Video(Buffer *buf, std::unique_ptr<VideoBackend> backend); // same, with an injected backend (tests)
    ~Video();
    Buffer *GetBuffer() { return buf_; }
    bool HasHeader();                                // video.go:130-147
    double Framerate() { return HasHeader() ? frame_rate_ : 0; }
    int Width() { return HasHeader() ? width_ : 0; }
    int Height() { return HasHeader() ? height_ : 0; }
    void SetNoDelay(bool v) { assume_no_b_frames_ = v; }
    // The form the parser hands pictures over in (include/mpeghip.h): SPARSE — its own (position, level) pairs, as the
    // reference's VLC loop produces them (video.go:680-745), MPEGHIP_PIC_SPARSE — or 128-byte UNITS of int16 levels.
    // Sparse is the product's form; units remain for callers of the unit ABI and for tests that compare the two.
    // Takes effect at the next picture: the form is latched when a picture begins (its offsets count units or dwords).
    void SetSparse(bool v) { sparse_wanted_ = v; }
    bool Sparse() const { return sparse_wanted_; }
    static void SetDefaultSparse(bool v); // the form of decoders created from now on (process-wide; sparse unless told otherwise)
    // test hook: every VLC table of the parser against a walk over its code list, for every possible look at the stream
    // (all 2^L prefixes of the table's longest code).  Returns the number of prefixes that decode differently (0).
    static uint64_t VlcSelfCheck();
    // test hook: the symbol table `table` (0 .. 8: address increment, type I / P / B, coded block pattern, motion code, DC size
    // luma / chroma, coefficient codes) finds at the top of `window`; false: no such table
    static bool VlcDecode(int table, uint64_t window, int *value, int *len);
    // (a picture parsed ahead — Decode below — is the reference's NEXT picture: its time is the decoder's time, and the stream
    // has not ended while it waits to be returned)
    double Time() const { return ahead_.valid ? ahead_.time : time_; }
    void SetTime(double t);
    void Rewind();
    // (... nor when the attempt to parse ahead found nothing: the reference learns that in its NEXT call — until then the answer
    // is what it was before the attempt)
    bool HasEnded() const { return ahead_.valid ? false : ahead_tried_ ? ended_before_ahead_ : buf_->HasEnded(); }
    // video.go:209-268.  Works ONE PICTURE AHEAD on the host: a call hands the picture parsed during the previous call to the
    // device, queues the read-back of the frame it returns (asynchronous, into one of two pinned buffers), parses the NEXT
    // picture while the device works — that picture's work is held back until the next call, so the device's frame store is
    // never ahead of the frames returned: Rewind / Seek drop the parsed picture and everything is as the reference has it — and
    // only then waits.  The returned Frame (one of two, alternating) and its planes are valid until the next Decode call
    // (mpeg.go:413-415).  SetLookahead(false): parse, submit, read back, return — nothing parsed ahead (MPEG::SeekFrame needs
    // that: it must not pull packets while it keeps audio packets out).
    Frame *Decode();
    void SetLookahead(bool v) { lookahead_ = v; }
    bool Lookahead() const { return lookahead_; }
    // Decode's frames straight out of the backend's host mirror (VideoBackend::mirrorAsync; the default where the backend has one):
    // the reconstruction launch has written them, nothing is read back.  false: the asynchronous read-back into two pinned frames.
    // (Switching ends the life of the frame in hand: the mirror's memory goes with it.)
    void SetHostMirror(bool v)
    {
        host_mirror_ = v;
        backend_->setMirror(v);
    }
    bool HostMirror() const { return host_mirror_; }
    // a lone decoder's large pictures are validated and packed on the device (VideoBackend::setDevicePackFrom): from n_mbs
    // macroblocks per hand-over on; 0 = the host packs everything
    void SetDevicePackFrom(uint32_t n_mbs) { backend_->setDevicePackFrom(n_mbs); }
    // Decode() in two halves, for VideoBatch: DecodeDeferred parses up to and including the picture that
    // completes the next output frame and hands its work to the backend WITHOUT reading anything back;
    // Fetch copies that frame's planes to the host (what makes Frame.Y/Cb/Cr.Data valid).
    bool DecodeDeferred(uint32_t *slot, double *time);
    // one picture of DecodeDeferred: 1 = a frame is complete (*slot, *time), 2 = a picture was consumed but
    // no frame is due yet (the stream's first reference picture): call again, 0 = nothing to decode now
    int DecodeStep(uint32_t *slot, double *time);
    Frame *Fetch(uint32_t slot, double time, bool read_back = true);
    const VideoStats &Stats() const { return stats_; }

    // used by Frame
    const uint8_t *fetchRGBA(uint32_t slot);

private:
    struct Motion { int FullPx = 0, RSize = 0, H = 0, V = 0; bool IsSet = false; };
    bool decodeSequenceHeader();
    void decodePicture();
    void decodeSlice(int slice);
    // The bit cursor of a slice's macroblocks, held in locals: nothing inside a macroblock refills the buffer (fields past its end
    // read as zeros, as Buffer::peek has them), and the Buffer's own members would travel through memory at every field —
    // the parser's stores may alias them.  decodeSlice copies it out of *buf_ and writes the position back (and lets the buffer
    // refill between macroblocks when it has to).
    struct Cursor {
        const uint8_t *data;
        size_t len, bit;
        uint64_t window() const // the next 57+ bits, left-aligned in 64, zero-padded past the end
        {
            const size_t byte = bit >> 3;
            uint64_t w;
            if (__builtin_expect(byte + 8 <= len, 1)) {
                memcpy(&w, data + byte, 8);
                w = __builtin_bswap64(w);
            } else {
                w = 0;
                for (size_t k = 0; k < 8; k++)
                    w = (w << 8) | (byte + k < len ? data[byte + k] : 0u);
            }
            return w << (bit & 7);
        }
        int bits(int count) // Buffer::read for 0 <= count <= 32
        {
            if (count == 0)
                return 0;
            const int v = (int)(window() >> (64 - count));
            bit += (size_t)count;
            return v;
        }
    };
    void decodeMacroblock(Cursor &c);
    void decodeMotionVectors(Cursor &c);
    int decodeMotionVector(Cursor &c, int rSize, int motion);
    void predictMacroblock();
    void emitPrediction(int mh, int mv, bool backward);
    void decodeBlock(Cursor &c, int block);
    void beginMacroblockRecord(bool intra);
    void endMacroblockRecord();
    void flushSubmit();
    Frame *frameForSlot(uint32_t slot);

    void init();
    Buffer *buf_;
    std::unique_ptr<VideoBackend> backend_;
    size_t luma_bytes_ = 0, chroma_bytes_ = 0;

    double frame_rate_ = 0, time_ = 0;
    int frames_decoded_ = 0;
    int width_ = 0, height_ = 0, mb_width_ = 0, mb_height_ = 0, mb_size_ = 0;
    int luma_width_ = 0, luma_height_ = 0, chroma_width_ = 0, chroma_height_ = 0;
    int64_t range_total_ = 0, range_chroma_ = 0;   // ends of the byte ranges a prediction may read (emitPrediction)
    int start_code_ = -1, picture_type_ = 0;
    Motion motion_forward_, motion_backward_;
    bool has_sequence_header_ = false;
    int quantizer_scale_ = 0;
    bool slice_begin_ = false;
    int macroblock_address_ = 0, mb_row_ = 0, mb_col_ = 0, macroblock_type_ = 0;
    bool macroblock_intra_ = false;
    int dc_predictor_[3] = {128, 128, 128};
    uint8_t intra_quant_[64], non_intra_quant_[64];
    bool has_reference_frame_ = false, assume_no_b_frames_ = false;

    // frame slots (rotation of video.go:406-409 / 430-433)
    uint32_t slot_cur_ = 0, slot_fwd_ = 1, slot_bwd_ = 2;

    // persistent blockData (video.go:101): only ever non-zero after an invalid block
    int32_t block_data_[64];
    bool block_dirty_ = false;
    bool sparse_;          // the form of the picture being recorded
    bool sparse_wanted_;   // ... of the next one (SetSparse)

    // per-picture recording
    std::vector<mpeghip_mb_desc> mbs_;
    // The picture's coefficient bytes.  coefs_.size() runs AHEAD of what is recorded (it grows in large steps, and shrinks
    // to coef_len_ when the arrays are handed over): in the sparse form the VLC loop writes a block's words straight behind the
    // macroblock's earlier blocks — coef_len_ + mb_pending_ — and endMacroblockRecord only has to accept them.
    CoefBytes coefs_;
    size_t coef_len_ = 0;               // bytes recorded (whole macroblocks)
    size_t mb_pending_ = 0;             // bytes the current macroblock's clean blocks have written behind coef_len_
    uint8_t *coefRoom(size_t bytes);    // room for `bytes` behind coef_len_ + mb_pending_ (no recording; may move coefs_)
    uint8_t *coefAppend(size_t bytes);  // `bytes` zeroed bytes recorded at coef_len_
    std::vector<uint8_t> written_;      // macroblock address already emitted in this submit
    // What decodeBlock leaves of a block: an intra block's DC (clamped to int16; needs_raw says when that lost something),
    // and its coded levels as MPEGHIP_PAIR words in scan order — in place in coefs_ for a sparse picture, in pair_scratch_
    // otherwise; raw: the snapshot of blockData for the blocks that need the reference's own arithmetic.
    struct BlockRec { bool valid; bool needs_raw; int16_t dc; int n_pairs; const uint32_t *pairs; int32_t raw[64]; };
    uint32_t pair_scratch_[6][66];
    struct MbRec { bool active = false, intra = false; int mb_x = 0, mb_y = 0; bool has_pred = false, backward = false;
                   int mv_x = 0, mv_y = 0; int qscale = 0; int cbp = 0; BlockRec blocks[6]; bool any_raw = false;
                   int valid_cbp = 0; /* the blocks of cbp that ended valid (decodeBlock), any_raw: one of them needs the snapshot form */
                   bool out_of_range = false; /* a copyMacroblock call of this macroblock would panic in the reference */ } rec_;

    Frame frames_[3];
    std::vector<uint8_t> host_planes_[3];
    std::vector<uint8_t> host_rgba_;
    VideoStats stats_;

    // ---- Decode's look-ahead
    struct Deferred { mpeghip_pic_desc pic; std::vector<mpeghip_mb_desc> mbs; CoefBytes coefs; };
    std::vector<Deferred> deferred_;      // hand-overs recorded while defer_submits_ (storage is kept and reused)
    size_t n_deferred_ = 0;
    bool defer_submits_ = false, lookahead_ = true, host_mirror_ = true;
    bool ahead_tried_ = false, ended_before_ahead_ = false; // an attempt to parse ahead was made in the last Decode call / HasEnded() before it
    struct Ahead { bool valid = false; uint32_t slot = 0; double time = 0; } ahead_; // the frame the parsed-ahead picture completes
    // what a dropped look-ahead must give back (Rewind): the parser state that outlives a picture
    struct Undo { uint32_t cur, fwd, bwd; int picture_type; bool has_reference_frame, block_dirty; int32_t block_data[64];
                  Motion motion_forward, motion_backward; VideoStats stats; } undo_;
    bool undo_valid_ = false;
    uint8_t *out_planes_[2] = {nullptr, nullptr};  // VideoBackend::allocPlanes: the two frames Decode alternates between
    Frame out_frames_[2];
    int out_next_ = 0;
    void parseAhead();
    void replayDeferred();
    void dropLookahead();
};

// Many independent streams of one picture size on one GPU: every DecodeAll() advances each stream by
// one output frame — the streams are parsed one after the other on the CPU, their pictures are
// reconstructed by ONE device call (one mpeghip_pic_desc per stream), then the
// frames are fetched.  This is the shape of the 10 000-stream deployment; a lone Video pays one
// launch + one read-back per 330-macroblock picture.
// The host thread pool of the batches (batch.cpp): run(n, fn) spreads fn(0 .. n-1) over the calling thread and its workers.
class HostPool;
// The CPU time the process gets, in cores: its affinity mask capped by the cgroup's CPU-time quota.  VideoBatch / AudioBatch
// pools never start more threads than this (rounded up): SetThreads(n) is a request, Threads() says what it became;
// SetThreads(0) asks for "as many as fit".
double EffectiveCores();
// ... its cgroup part: the tightest CPU-time quota (in cores; 0: none) of the process's cgroup and all its ancestors, v2 and v1
// (root / proc_file: stand-ins for /sys/fs/cgroup and /proc/self/cgroup — tests; nullptr = the real ones)
double CgroupQuotaCores(const char *root, const char *proc_file);

class VideoBatch {
public:
    VideoBatch(Device *dev, uint32_t n_streams);
    VideoBatch(std::unique_ptr<BatchStore> store, uint32_t n_streams); // injected store (tests)
    ~VideoBatch();
    // NewVideo over `buf` as stream number Streams(); the Video is owned by the batch.  All streams must
    // have the same picture size (std::runtime_error from the first Decode otherwise).
    Video *AddStream(Buffer *buf);
    uint32_t Streams() const { return (uint32_t)videos_.size(); }
    Video *Stream(uint32_t i) { return videos_[i].get(); }
    // frames[i] = the next frame of stream i, or nullptr (ended / no frame yet).  With fetch = false the
    // planes stay on the device (frames[i]->Y.Data is stale): for consumers that read RGBA / planes there.
    size_t DecodeAll(std::vector<Frame *> &frames, bool fetch = true);
    // Parse with n host threads (default 1).  The bitstream parse is the serial part of a stream but streams
    // are independent: with n > 1 every round of DecodeAll parses its streams on a pool of n threads (each
    // stream's device requests are recorded privately), then replays the recorded requests in stream order on
    // the calling thread — the device sees an equivalent sequence of calls (same pictures, same order per stream).  Load callbacks of different
    // streams' Buffers may then run concurrently.
    void SetThreads(unsigned n);
    unsigned Threads() const { return threads_; }
    // Bind the parse threads (started by SetThreads) to the cores of a host NUMA node — the one the batch's GPU is attached
    // to (Device::NumaNode): on a two-socket node every device is fed from its own socket.  -1: leave them alone (default).
    // Call it while no DecodeAll is running (it restarts the pool).  The calling thread of DecodeAll parses too and is NOT
    // bound by this (it is the application's thread: ShardedVideoBatch binds its own shard threads).
    void SetNumaNode(int node);
    int NumaNode() const { return numa_node_; }
    // threads of this batch that asked to be bound to the node / whose binding failed (no such node, sched_setaffinity refused)
    void NumaPins(unsigned out[2]) const;
    // Staged submits of sparse pictures are validated and packed ON THE DEVICE (the default since round 6:
    // mpeghip_video_stage_begin_device — the host side of the hand-over shrinks to a copy; 16 % more pictures per second from
    // bitstreams than the other form) or, SetDevicePack(false), by the pool's threads on the host (a picture the validator refuses
    // then makes the DecodeAll that sent it throw, as a refused mpeghip_video_submit does).  The device reports DEFERRED, and per
    // PICTURE — the unit of failure of the reference (video.go:374-460): the refused picture is not reconstructed, the commit's
    // other pictures (other streams) are.  Every round asks for the verdict on the commit before it between its parse and its
    // own hand-over (the verdict has been there for a whole parse by then: no wait), so a picture refused in round k is reported —
    // DecodeAll throws, RefusedStreams() names the streams — BEFORE anything of round k + 1 reaches the device; that round is
    // parsed and HELD: the next DecodeAll commits it instead of parsing and returns the tick's frames.  No healthy stream loses a
    // picture or a frame.  (The product's parser does not emit pictures the validator refuses)
    void SetDevicePack(bool on) { device_pack_ = on; }
    bool DevicePack() const { return device_pack_; }
    void Sync();
    const std::vector<uint32_t> &RefusedStreams() const { return refused_streams_; } // of the refusal last thrown
    // test hook: the next picture stream `stream` hands over is damaged on its way (a quantiser scale of 0: what the parser cannot
    // produce and every validator refuses) — the error contract above, end to end
    void DebugDamageNextPicture(uint32_t stream) { debug_damage_.store((int64_t)stream); }
    void Flush();                                  // submit whatever is queued
    uint64_t DeviceSubmits() const { return device_submits_; }
    uint64_t QueuedPictures() const { return queued_pictures_; }
    // wall seconds spent so far in {parse rounds, stage begin, puts, commits} (pooled DecodeAll only)
    void PhaseSeconds(double out[4]) const { out[0] = t_parse_; out[1] = t_begin_; out[2] = t_put_; out[3] = t_commit_; }

private:
    class Port;
    friend class Port;
    void queue(uint32_t stream, const mpeghip_pic_desc &pic, const mpeghip_mb_desc *mbs, uint32_t n_mbs, const uint8_t *coefs,
               size_t coef_bytes);
    void openStore(int width, int height);
    std::unique_ptr<BatchStore> store_;
    std::vector<Port *> ports_;                    // (owned by the Videos)
    std::unique_ptr<HostPool> pool_;
    unsigned threads_ = 1;
    int numa_node_ = -1;
    uint32_t capacity_;
    int width_ = 0, height_ = 0;
    std::vector<std::unique_ptr<Video>> videos_;
    std::vector<mpeghip_pic_desc> pics_;
    std::vector<mpeghip_mb_desc> mbs_;
    std::vector<uint8_t> coefs_;
    bool any_sparse_queued_ = false;
    bool device_pack_ = true;
    bool verdict_owed_ = false;                    // a device-packed commit whose verdict has not been asked for
    // a tick's state: a DecodeAll that throws a refusal keeps its parsed round (held) for the next call
    struct Round { std::vector<uint32_t> slot, todo; std::vector<double> time; std::vector<uint8_t> got; std::vector<int> result;
                   bool held = false; } round_;
    std::exception_ptr held_refusal_;              // learnt while a tick's frames were fetched: thrown by the next call, before it parses
    std::vector<uint32_t> refused_streams_;
    std::atomic<int64_t> debug_damage_{-1};
    void reapVerdict();
    std::vector<uint8_t> pending_;                 // stream already has a picture in the open batch
    uint64_t device_submits_ = 0, queued_pictures_ = 0;
    double t_parse_ = 0, t_put_ = 0, t_commit_ = 0, t_begin_ = 0; // wall time per phase (PhaseSeconds)
};

// Streams sharded over SEVERAL devices ("stream s -> GPU s mod G ... host driver = G threads, one
// HIP stream each; collective: none").  Every shard is a VideoBatch of its own on its own Device (its own libmpeghip
// context, HIP stream and frame store), driven by its own host thread: a tick wakes the G threads, each advances its
// streams by one frame with one device call, and the caller gets the frames back in global stream order.  Shards
// share nothing — no collective, no peer access, no lock on the data path.
class ShardedVideoBatch {
public:
    ShardedVideoBatch(const std::vector<Device *> &devices, uint32_t n_streams);
    ShardedVideoBatch(std::vector<std::unique_ptr<BatchStore>> stores, uint32_t n_streams); // injected stores (tests)
    ~ShardedVideoBatch();
    Video *AddStream(Buffer *buf);                 // becomes stream number Streams(), on shard Streams() % Shards()
    uint32_t Streams() const { return n_added_; }
    uint32_t Shards() const { return (uint32_t)shards_.size(); }
    uint32_t ShardOf(uint32_t stream) const { return stream % Shards(); }
    VideoBatch &Shard(uint32_t g);
    // parse threads of ALL shards together (0: as many as the process has CPU time for — EffectiveCores — and never more): the
    // budget is divided among the shards, at least one thread each; Threads() = what the shards' pools add up to
    void SetThreads(unsigned n);
    unsigned Threads() const;
    void SetDevicePack(bool on);                   // VideoBatch::SetDevicePack of every shard
    void Sync();                                   // VideoBatch::Sync of every shard: waits; throws the first deferred error
    // one tick of every shard, concurrently; frames[s] = the next frame of global stream s (or nullptr)
    size_t DecodeAll(std::vector<Frame *> &frames, bool fetch = true);

private:
    struct ShardState;
    void start(uint32_t n_streams);
    void stopWorkers();
    std::vector<std::unique_ptr<ShardState>> shards_;
    uint32_t n_added_ = 0, capacity_ = 0;
};

// -------------------------------------------------------------------- audio.go
enum AudioFormat { AudioF32N = 0, AudioF32NLR = 1, AudioF32 = 2, AudioS16 = 3 }; // audio.go:12-23
constexpr int SamplesPerFrame = 1152;                                             // audio.go:9

struct Samples {              // audio.go:27-36
    double Time = 0;
    std::vector<int16_t> S16;
    std::vector<float> F32, Left, Right, Interleaved;
    AudioFormat format = AudioF32N;
    const uint8_t *Bytes(size_t *len) const;     // audio.go:39-50
};

class Audio {
public:
    Audio(Buffer *buf, Device *dev, int fma_mode = MPEGHIP_AUDIO_FMA_NONE); // NewAudio (audio.go:83-104)
    Audio(Buffer *buf, std::unique_ptr<AudioBackend> backend);
    ~Audio();
    Buffer *GetBuffer() { return buf_; }
    bool HasHeader();
    int Samplerate();
    int Channels() const { return channels_; }
    double Time() const { return ahead_valid_ ? ahead_time_ : time_; }   // (a frame parsed ahead is the reference's NEXT frame)
    void SetTime(double t);
    void Rewind();
    bool HasEnded() const { return ahead_valid_ ? false : ahead_tried_ ? ended_before_ahead_ : buf_->HasEnded(); }
    void SetFormat(AudioFormat f) { format_ = f; samples_[0].format = samples_[1].format = f; } // MPEG.SetAudioFormat (mpeg.go:234)
    // audio.go:163-182.  Works ONE FRAME AHEAD on the host, like Video::Decode: a call hands the frame parsed during the previous
    // call to the device (asynchronous synthesis from / into pinned memory), parses the NEXT frame while the device works — held
    // back until the next call: the V ring is never ahead of the frames returned (Rewind keeps it, audio.go:149-154) — then
    // waits.  The returned Samples (one of two, alternating) are valid until the next Decode call (mpeg.go:435-437).
    Samples *Decode();
    void SetLookahead(bool v) { lookahead_ = v; }

private:
    struct QuantizerSpec { uint16_t Levels; uint8_t Group, Bits; };
    int decodeHeader();
    void decodeFrame(int32_t (*frame_samples)[36][32]);   // the parse of one frame: its 2 x 36 x 32 sub-band samples
    bool parseNext(int *buf, double *time);                // audio.go:163-182 up to the synthesis: frame -> in_[*buf]
    const QuantizerSpec *readAllocation(int sb, int tab3);
    void readSamples(int ch, int sb, int part);

    void init();
    Buffer *buf_;
    std::unique_ptr<AudioBackend> backend_;
    double time_ = 0;
    int samples_decoded_ = 0, samplerate_index_ = 3, bitrate_index_ = 0, version_ = 0, layer_ = 0, mode_ = 0;
    int channels_ = 0, bound_ = 0, next_frame_data_size_ = 0;
    bool has_header_ = false;
    const QuantizerSpec *allocation_[2][32] = {};
    uint8_t scale_factor_info_[2][32] = {};
    int scale_factor_[2][32][3] = {};
    int sample_[2][32][3] = {};
    int32_t *in_[2] = {nullptr, nullptr};   // AudioBackend::allocSamples: [2][36][32] each
    int in_next_ = 0;
    Samples samples_[2];
    bool lookahead_ = true, ahead_valid_ = false, ahead_failed_ = false;
    bool ahead_tried_ = false, ended_before_ahead_ = false; // as Video's
    int ahead_buf_ = 0;
    double ahead_time_ = 0;
    AudioFormat format_ = AudioF32N;
    static const QuantizerSpec quant_tab_[17];
};

// Many MP2 streams on one GPU: every DecodeAll() advances each stream by one frame; the streams are
// parsed one after the other on the CPU, then ONE device call synthesises all of them (streams without a
// frame in this tick sit it out).  All streams share the output format.
class AudioBatch {
public:
    AudioBatch(Device *dev, uint32_t n_streams, AudioFormat format = AudioF32N, int fma_mode = MPEGHIP_AUDIO_FMA_NONE);
    AudioBatch(std::unique_ptr<AudioBatchStore> store, uint32_t n_streams, AudioFormat format, int fma_mode);
    ~AudioBatch();
    Audio *AddStream(Buffer *buf);                 // NewAudio over `buf`, owned by the batch
    uint32_t Streams() const { return (uint32_t)audios_.size(); }
    Audio *Stream(uint32_t i) { return audios_[i].get(); }
    // samples[i] = the next frame of stream i or nullptr; valid until the next DecodeAll
    size_t DecodeAll(std::vector<Samples *> &samples);
    void Flush();
    uint64_t DeviceCalls() const { return device_calls_; }
    // Host threads of the parse (default 1): the streams' frames of a DecodeAll are parsed side by side — every stream records
    // into its own slot of the batch's input — and synthesised by the one device call behind them.  Call while idle.
    void SetThreads(unsigned n);
    unsigned Threads() const { return threads_; }

private:
    class Port;
    friend class Port;
    std::unique_ptr<AudioBatchStore> store_;
    uint32_t capacity_;
    AudioFormat format_;
    std::vector<std::unique_ptr<Audio>> audios_;
    int32_t *in_ = nullptr;                        // [capacity][2][36][32]   (store_->allocHost: page-locked on a device store)
    uint8_t *out_ = nullptr;                       // [capacity][2304] elements of the format
    std::vector<uint8_t> active_;
    struct Dest { void *out = nullptr, *out2 = nullptr; };
    std::vector<Dest> dest_;
    uint64_t device_calls_ = 0;
    std::unique_ptr<HostPool> pool_;
    unsigned threads_ = 1;
    bool parallel_ = false;                        // a pooled parse is running: no stream may flush the batch from inside it
};

// -------------------------------------------------------------------- demux.go
struct Packet {               // demux.go:11-17
    int Type = 0;
    double Pts = 0;
    const uint8_t *Data = nullptr;
    size_t Len = 0;
    int length = 0;
};
constexpr int PacketInvalidTS = -1, PacketPrivate = 0xBD, PacketAudio1 = 0xC0, PacketAudio2 = 0xC1,
              PacketAudio3 = 0xC2, PacketAudio4 = 0xC3, PacketVideo1 = 0xE0; // demux.go:20-29

class Demux {
public:
    explicit Demux(Buffer *buf);                   // NewDemux (demux.go:61-76); check HasHeaders()
    Buffer *GetBuffer() { return buf_; }
    bool HasHeaders();                             // demux.go:85-155
    int NumVideoStreams() { return HasHeaders() ? num_video_streams_ : 0; }
    int NumAudioStreams() { return HasHeaders() ? num_audio_streams_ : 0; }
    void Rewind();
    bool HasEnded() const { return buf_->HasEnded(); }
    Packet *Decode();                              // demux.go:473-516
    bool Probe(size_t probe_size);                 // demux.go:158-198: count the streams that really occur
    // demux.go:208-352: packet of `type` with the PTS just before seek_time (0-based); with
    // force_intra only packets that start an intra picture.  nullptr if none.
    Packet *Seek(double seek_time, int type, bool force_intra);
    double StartTime(int type);                    // demux.go:357-404: lowest PTS of the type, or PacketInvalidTS
    double Duration(int type);                     // demux.go:406-457: highest - lowest PTS + one frame

private:
    double decodeTime();
    Packet *decodePacket(int type);
    Packet *packet();
    void bufferSeek(size_t pos);
    Buffer *buf_;
    double sys_clock_ref_ = 0, last_decoded_pts_ = 0;
    std::map<int, double> start_time_, duration_, first_pts_, last_pts_;
    size_t last_file_size_ = 0;
    int start_code_ = -1;
    bool has_pack_header_ = false, has_system_header_ = false, has_headers_ = false;
    int num_audio_streams_ = 0, num_video_streams_ = 0;
    Packet current_, next_;
};

// --------------------------------------------------------------------- mpeg.go
using VideoFunc = std::function<void(class MPEG *, Frame *)>;    // mpeg.go:48
using AudioFunc = std::function<void(class MPEG *, Samples *)>;  // mpeg.go:51

// High-level player facade (mpeg.go:57-669): demux -> per-stream buffers -> decoders,
// A/V clocking, seeking.
class MPEG {
public:
    // mpeg.New (mpeg.go:85-117).  Throws std::runtime_error("invalid MPEG-PS") like ErrInvalidMPEG /
    // ErrInvalidHeader.  `data` must outlive the object.
    MPEG(const uint8_t *data, size_t len, Device *dev, int audio_fma_mode = MPEGHIP_AUDIO_FMA_NONE);
    // same, with injected backends (tests): the factories are called when the decoders are created
    struct Backends {
        std::function<std::unique_ptr<VideoBackend>()> video;
        std::function<std::unique_ptr<AudioBackend>(int fma_mode)> audio;
    };
    MPEG(const uint8_t *data, size_t len, Backends backends, int audio_fma_mode = MPEGHIP_AUDIO_FMA_NONE);
    ~MPEG();
    bool HasHeaders();
    bool HasEnded() const { return has_ended_; }
    void SetVideoEnabled(bool e);
    void SetAudioEnabled(bool e);
    void SetLoop(bool l) { loop_ = l; }
    bool Loop() const { return loop_; }                   // mpeg.go:338
    bool VideoEnabled() const { return video_enabled_; }  // mpeg.go:170
    bool AudioEnabled() const { return audio_enabled_; }  // mpeg.go:245
    void SetAudioStream(int stream_index);                // 0..3, default 0 (mpeg.go:270-279)
    void SetAudioLeadTime(double s) { audio_lead_time_ = s; }
    double AudioLeadTime() const { return audio_lead_time_; }  // mpeg.go:301
    void SetAudioFormat(AudioFormat f);
    AudioFormat GetAudioFormat() const { return audio_format_; } // mpeg.go:229 (AudioFormat() there: the type's name here)
    // mpeg.go:155 Done(): the reference sends `true` on a channel of capacity 1 when the stream ends without looping
    // (handleEnd, :625-632).  Here: a callback run at that moment, and a flag that one TakeDone() call consumes — what
    // `select { case <-m.Done(): }` does to the channel.
    void SetDoneCallback(std::function<void()> f) { done_cb_ = std::move(f); }
    bool TakeDone()
    {
        const bool d = done_pending_;
        done_pending_ = false;
        return d;
    }
    void SetVideoCallback(VideoFunc f) { video_cb_ = std::move(f); }
    void SetAudioCallback(AudioFunc f) { audio_cb_ = std::move(f); }
    int NumVideoStreams() { return demux_->NumVideoStreams(); }
    int NumAudioStreams() { return demux_->NumAudioStreams(); }
    int Width() { return initDecoders() && video_ ? video_->Width() : 0; }
    int Height() { return initDecoders() && video_ ? video_->Height() : 0; }
    double Framerate() { return initDecoders() && video_ ? video_->Framerate() : 0; }
    int Samplerate() { return initDecoders() && audio_ ? audio_->Samplerate() : 0; }
    int Channels() { return initDecoders() && audio_ ? audio_->Channels() : 0; }
    double Time() const { return time_; }
    double Duration() { return demux_->Duration(PacketVideo1); }   // mpeg.go:318-320
    bool Probe(size_t probe_size);      // mpeg.go:141-152
    void Rewind();
    // mpeg.go:460-522: the intra frame just before `seconds` (clamped to 0..duration), or, with
    // seek_exact, the frame at that time (decodes forward from the intra frame).  No callbacks.
    Frame *SeekFrame(double seconds, bool seek_exact);
    // mpeg.go:524-576: SeekFrame + the video callback exactly once + audio re-synchronised
    bool Seek(double seconds, bool seek_exact);
    void Decode(double tick_seconds);   // mpeg.go:356-411
    Frame *DecodeVideo();               // mpeg.go:416-433
    Samples *DecodeAudio();             // mpeg.go:438-455
    Video *GetVideo() { return video_.get(); }
    Audio *GetAudio() { return audio_.get(); }

private:
    bool initDecoders();
    void handleEnd();
    std::function<void()> done_cb_;
    bool done_pending_ = false;
    void readPackets(int requested_type);
    void open(const uint8_t *data, size_t len);
    Backends backends_;
    int audio_fma_mode_;
    std::unique_ptr<Buffer> buf_, video_buf_, audio_buf_;
    std::unique_ptr<Demux> demux_;
    std::unique_ptr<Video> video_;
    std::unique_ptr<Audio> audio_;
    double time_ = 0, audio_lead_time_ = 0;
    bool loop_ = false, has_ended_ = false, has_decoders_ = false;
    bool video_enabled_ = true, audio_enabled_ = true;
    int video_packet_type_ = 0, audio_packet_type_ = 0, audio_stream_index_ = 0;
    AudioFormat audio_format_ = AudioF32N;
    VideoFunc video_cb_;
    AudioFunc audio_cb_;
};

} // namespace mpeg
