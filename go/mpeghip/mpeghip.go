//go:build hip

// Package mpeghip is the cgo binding of libmpeghip (include/mpeghip.h), the MI355X
// reconstruction core.  It is the thin shim the north star asks for: Go stays the host
// language, one cgo call per picture / audio frame.
//
// NOTE: no Go toolchain exists in the image this repository was built in, so this
// package has been written against the C header but never compiled; every C entry point
// it binds is exercised through ctypes by tests/ instead.  See INTEGRATION.md.
package mpeghip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../mpeg_amd -lmpeghip -Wl,-rpath,${SRCDIR}/../../mpeg_amd
#include <stdlib.h>
#include "mpeghip.h"
*/
import "C"

import (
	"errors"
	"unsafe"
)

// PicDesc mirrors mpeghip_pic_desc (16 bytes).
type PicDesc struct {
	Stream          uint32
	Cur, Fwd, Bwd   uint8
	Flags           uint8
	MbFirst, MbCount uint32
}

// MbDesc mirrors mpeghip_mb_desc (32 bytes).
type MbDesc struct {
	Pic        uint32
	MbX, MbY   uint16
	MvX, MvY   int16
	Flags      uint8
	Cbp        uint8
	Qscale     uint8
	_          uint8
	CoefOff    uint32
	_          [3]uint32
}

const (
	MbIntra   = C.MPEGHIP_MB_INTRA
	MbRefFwd  = C.MPEGHIP_MB_REF_FWD
	MbRefBwd  = C.MPEGHIP_MB_REF_BWD
	MbCoefRaw = C.MPEGHIP_MB_COEF_RAW
	PicRGBA   = C.MPEGHIP_PIC_RGBA
	PicSparse = C.MPEGHIP_PIC_SPARSE // the picture's coefficient data: (position, level) pairs, CoefOff in dwords

	FMANone   = C.MPEGHIP_AUDIO_FMA_NONE   // mul and add rounded separately (pure Go / SSE2)
	FMAWindow = C.MPEGHIP_AUDIO_FMA_WINDOW // fused multiply-add in the window (amd64 AVX2)

	AudioF32N   = C.MPEGHIP_AUDIO_F32N // = mpeg.AudioF32N ... AudioS16, same order
	AudioF32NLR = C.MPEGHIP_AUDIO_F32NLR
	AudioF32    = C.MPEGHIP_AUDIO_F32
	AudioS16    = C.MPEGHIP_AUDIO_S16
)

// lastError: mpeghip_last_error is per OS thread; goroutines that call into the library keep theirs with
// runtime.LockOSThread (the one that owns a Context does so anyway: a HIP context is bound to its thread's
// device selection), otherwise the text may belong to another call — the code is always right.
func lastError(rc C.int) error {
	if rc == C.MPEGHIP_OK {
		return nil
	}
	return errors.New(C.GoString(C.mpeghip_last_error()))
}

// Context owns one GPU + stream.
type Context struct{ h *C.mpeghip_ctx }

// ABIVersion is the MPEGHIP_ABI_VERSION this binding was written against; NewContext refuses a libmpeghip of another version
// (version 2: snapshot blocks of the sparse form carry a count word, macroblocks name their words in order, chroma as
// Cb|Cr pairs in device memory, device-packed stages with deferred errors; version 3: asynchronous read-back and synthesis for a
// decoder that works one picture / frame ahead, a device-packed commit refuses single PICTURES — Verdict / Refused —,
// mpeghip_ctx_pci_bus_id).
const ABIVersion = 3

func NewContext(device int) (*Context, error) {
	if int(C.mpeghip_abi_version()) != ABIVersion || C.MPEGHIP_ABI_VERSION != ABIVersion {
		return nil, errors.New("mpeghip: libmpeghip has another ABI version than this binding")
	}
	c := &Context{}
	if err := lastError(C.mpeghip_ctx_create(C.int(device), nil, &c.h)); err != nil {
		return nil, err // no GPU: there is no CPU fallback
	}
	return c, nil
}
func (c *Context) Close() { C.mpeghip_ctx_destroy(c.h) }

// NumaNode is the host NUMA node the context's GPU is attached to (-1: unknown): where the goroutines that
// parse for this device should run (runtime.LockOSThread + sched_setaffinity) on a two-socket node.
func (c *Context) NumaNode() int { return int(C.mpeghip_ctx_numa_node(c.h)) }

// PCIBusID is the PCI address of the context's GPU ("0000:c1:00.0"): what identifies the physical device when several
// processes each see their own ordinal 0.
func (c *Context) PCIBusID() (string, error) {
	var buf [64]C.char
	if err := lastError(C.mpeghip_ctx_pci_bus_id(c.h, &buf[0], C.size_t(len(buf)))); err != nil {
		return "", err
	}
	return C.GoString(&buf[0]), nil
}

// PinnedAlloc hands out n bytes of pinned host memory of the library (C-allocated: Go's collector neither moves nor frees it) as a
// Go slice: what the asynchronous entries below read and write while the caller goes on; PinnedFree gives it back.
func (c *Context) PinnedAlloc(n int) []byte {
	p := C.mpeghip_pinned_alloc(c.h, C.size_t(n))
	if p == nil {
		return nil
	}
	return unsafe.Slice((*byte)(p), n)
}
func (c *Context) PinnedFree(b []byte) {
	if len(b) > 0 {
		C.mpeghip_pinned_free(c.h, unsafe.Pointer(&b[0]))
	}
}

// Video is the 3-slot frame store + reconstruction of one stream.
type Video struct {
	h    *C.mpeghip_video
	Info C.mpeghip_video_info
}

func (c *Context) OpenVideo(width, height int) (*Video, error) {
	v := &Video{}
	if err := lastError(C.mpeghip_video_open(c.h, C.uint32_t(width), C.uint32_t(height), 1, &v.h)); err != nil {
		return nil, err
	}
	C.mpeghip_video_info_get(v.h, &v.Info)
	return v, nil
}
func (v *Video) Close() { C.mpeghip_video_close(v.h) }

func (v *Video) SetQuant(intra, nonIntra *[64]byte) error {
	return lastError(C.mpeghip_video_set_quant(v.h, 0, (*C.uint8_t)(&intra[0]), (*C.uint8_t)(&nonIntra[0])))
}

// SetTilePolicy overrides the library's per-batch choice of the reconstruction kernel instance
// (0 = automatic, 1 = int16 coefficient tile, 2 = int32 tile; include/mpeghip.h).  Results are identical either way.
func (v *Video) SetTilePolicy(policy int) error {
	return lastError(C.mpeghip_video_set_tile_policy(v.h, C.int(policy)))
}

// Submit hands one picture to the GPU.  The slices are only read during the call (cgo rule).
func (v *Video) Submit(pic *PicDesc, mbs []MbDesc, coefs []byte) error {
	var mp unsafe.Pointer
	var cp unsafe.Pointer
	if len(mbs) > 0 {
		mp = unsafe.Pointer(&mbs[0])
	}
	if len(coefs) > 0 {
		cp = unsafe.Pointer(&coefs[0])
	}
	return lastError(C.mpeghip_video_submit(v.h, (*C.mpeghip_pic_desc)(unsafe.Pointer(pic)), 1,
		(*C.mpeghip_mb_desc)(mp), C.uint32_t(len(mbs)), cp, C.size_t(len(coefs))))
}

// Pair is one coded coefficient of the sparse hand-over (MPEGHIP_PAIR): what the VLC loop produces per symbol
// (video.go:680-745), position = column*8 + row.  A block = a count word, then its pairs (an intra block's DC first).
func Pair(level int, position int) uint32 {
	return uint32(uint16(int16(level)))<<16 | uint32(position)<<2
}

// SubmitSparse hands one picture in the sparse form to the GPU (mpeghip_video_submit_sparse): words hold, per
// macroblock from mbs[k].CoefOff on (in dwords), its coded blocks as count + pairs; snapshot blocks as 64 int32.
func (v *Video) SubmitSparse(pic *PicDesc, mbs []MbDesc, words []uint32) error {
	var mp, wp unsafe.Pointer
	if len(mbs) > 0 {
		mp = unsafe.Pointer(&mbs[0])
	}
	if len(words) > 0 {
		wp = unsafe.Pointer(&words[0])
	}
	return lastError(C.mpeghip_video_submit_sparse(v.h, (*C.mpeghip_pic_desc)(unsafe.Pointer(pic)), (*C.mpeghip_mb_desc)(mp),
		C.uint32_t(len(mbs)), (*C.uint32_t)(wp), C.size_t(len(words))))
}

// ReadPlanes fills host slices (len = Info.luma_bytes / chroma_bytes) with the slot's planes.
func (v *Video) ReadPlanes(slot int, y, cb, cr []byte) error {
	return lastError(C.mpeghip_video_read_planes(v.h, 0, C.uint32_t(slot),
		(*C.uint8_t)(&y[0]), (*C.uint8_t)(&cb[0]), (*C.uint8_t)(&cr[0])))
}

// ReadPlanesAsync queues the read-back of the slot's planes — luma | Cb | Cr, linear, Info.luma_bytes + 2 * Info.chroma_bytes — into
// dst (memory of Context.PinnedAlloc) behind everything submitted so far and returns at once; ReadWait(ticket) blocks until the
// planes are there.  What lets Video.Decode parse picture N+1 while picture N is on the device (go/patch/frame_hip.go).
func (v *Video) ReadPlanesAsync(slot int, dst []byte) (uint64, error) {
	var ticket C.uint64_t
	err := lastError(C.mpeghip_video_read_planes_async(v.h, 0, C.uint32_t(slot), (*C.uint8_t)(&dst[0]), &ticket))
	return uint64(ticket), err
}
func (v *Video) ReadWait(ticket uint64) error { return lastError(C.mpeghip_video_read_wait(v.h, C.uint64_t(ticket))) }

// HostMirror gives every slot a copy of its planes — luma | Cb | Cr, linear — in pinned host memory of the library, which the
// reconstruction launch itself keeps up to date (for submits small enough for the library's four-waves-per-chunk kernel: a lone
// decoder's); MirrorAsync names the slot's copy (C memory, valid while the mirror is on) and a ticket: ReadWait(ticket) returns when
// the copy holds the slot as it is after everything submitted so far.  No untiling launch, no copy — unless something else wrote
// the slot since.  The copy is overwritten by the next picture reconstructed into the slot: the lifetime the reference gives a
// returned *Frame (mpeg.go:413-415).
func (v *Video) HostMirror(on bool) error {
	flag := C.int(0)
	if on {
		flag = 1
	}
	return lastError(C.mpeghip_video_host_mirror(v.h, flag))
}
func (v *Video) MirrorAsync(slot int) ([]byte, uint64, error) {
	var planes *C.uint8_t
	var ticket C.uint64_t
	if err := lastError(C.mpeghip_video_mirror_async(v.h, 0, C.uint32_t(slot), &planes, &ticket)); err != nil {
		return nil, 0, err
	}
	n := int(v.Info.luma_bytes + 2*v.Info.chroma_bytes)
	return unsafe.Slice((*byte)(unsafe.Pointer(planes)), n), uint64(ticket), nil
}

// RGBA converts the slot on the device (Frame.RGBA) and copies width*height*4 bytes into dst.
func (v *Video) RGBA(slot int, dst []byte) error {
	if err := lastError(C.mpeghip_video_rgba_convert(v.h, C.uint32_t(slot), 0, 1)); err != nil {
		return err
	}
	return lastError(C.mpeghip_video_read_rgba(v.h, 0, C.uint32_t(slot), (*C.uint8_t)(&dst[0])))
}

// Audio is the MP2 synthesis state (V ring + vPos) of one stream.
type Audio struct{ h *C.mpeghip_audio }

func (c *Context) OpenAudio(fmaMode int) (*Audio, error) {
	a := &Audio{}
	if err := lastError(C.mpeghip_audio_open(c.h, 1, C.int(fmaMode), &a.h)); err != nil {
		return nil, err
	}
	return a, nil
}
func (a *Audio) Close() { C.mpeghip_audio_close(a.h) }

// Synth turns one frame of sub-band samples ([2][36][32]int32) into 2304 output elements.
func (a *Audio) Synth(samples *[2][36][32]int32, format int, out unsafe.Pointer) error {
	return lastError(C.mpeghip_audio_synth(a.h, (*C.int32_t)(unsafe.Pointer(samples)), 1, C.int(format), out))
}

// SynthAsync queues the synthesis of one frame: samples and out are memory of Context.PinnedAlloc (9 216 bytes of sub-band samples;
// 2 304 elements of the format's type) that stays untouched until SynthWait(ticket) returns.  UndoLast forgets the last launch: the V
// ring and vPos are again what they were before it (one level).
func (a *Audio) SynthAsync(samples []byte, format int, out []byte) (uint64, error) {
	var ticket C.uint64_t
	err := lastError(C.mpeghip_audio_synth_async(a.h, (*C.int32_t)(unsafe.Pointer(&samples[0])), 1, C.int(format), unsafe.Pointer(&out[0]), &ticket))
	return uint64(ticket), err
}
func (a *Audio) SynthWait(ticket uint64) error { return lastError(C.mpeghip_audio_synth_wait(a.h, C.uint64_t(ticket))) }
func (a *Audio) UndoLast() error               { return lastError(C.mpeghip_audio_undo_last(a.h)) }

// ---- many streams on one GPU (the shape of mpeg_amd/host/batch.cpp's VideoBatch / AudioBatch)

// OpenVideoStreams is OpenVideo for n independent streams of one picture size: PicDesc.Stream selects
// the stream, SubmitBatch reconstructs one picture of each of them with a single device call.
func (c *Context) OpenVideoStreams(width, height, n int) (*Video, error) {
	v := &Video{}
	if err := lastError(C.mpeghip_video_open(c.h, C.uint32_t(width), C.uint32_t(height), C.uint32_t(n), &v.h)); err != nil {
		return nil, err
	}
	C.mpeghip_video_info_get(v.h, &v.Info)
	return v, nil
}

func (v *Video) SetQuantOf(stream int, intra, nonIntra *[64]byte) error {
	return lastError(C.mpeghip_video_set_quant(v.h, C.uint32_t(stream), (*C.uint8_t)(&intra[0]), (*C.uint8_t)(&nonIntra[0])))
}

// SubmitBatch hands pictures of DIFFERENT streams to the GPU (pics[i].MbFirst/MbCount index mbs,
// mbs[j].Pic indexes pics, coefficient offsets are relative to coefs).
func (v *Video) SubmitBatch(pics []PicDesc, mbs []MbDesc, coefs []byte) error {
	if len(pics) == 0 {
		return nil
	}
	var mp, cp unsafe.Pointer
	if len(mbs) > 0 {
		mp = unsafe.Pointer(&mbs[0])
	}
	if len(coefs) > 0 {
		cp = unsafe.Pointer(&coefs[0])
	}
	return lastError(C.mpeghip_video_submit(v.h, (*C.mpeghip_pic_desc)(unsafe.Pointer(&pics[0])), C.uint32_t(len(pics)),
		(*C.mpeghip_mb_desc)(mp), C.uint32_t(len(mbs)), cp, C.size_t(len(coefs))))
}

// Stage is one submit assembled picture by picture; Put may be called from several goroutines for
// distinct i (one parser goroutine per stream), Commit from the goroutine that owns the context.
type Stage struct {
	h      *C.mpeghip_stage
	nMbs   []uint32 // device-packed stages: the sizes Map slices by
	nWords []uint64
}

// StageBegin reserves room for len(nMbs) pictures of the given sizes in the next pinned staging buffer.
func (v *Video) StageBegin(nMbs []uint32, coefBytes []uint64) (*Stage, error) {
	s := &Stage{}
	if len(nMbs) == 0 || len(nMbs) != len(coefBytes) {
		return nil, errors.New("mpeghip: StageBegin: nMbs and coefBytes must have the same, non-zero length")
	}
	sizes := make([]C.size_t, len(coefBytes))
	for i, b := range coefBytes {
		sizes[i] = C.size_t(b)
	}
	if err := lastError(C.mpeghip_video_stage_begin(v.h, C.uint32_t(len(nMbs)), (*C.uint32_t)(unsafe.Pointer(&nMbs[0])),
		&sizes[0], &s.h)); err != nil {
		return nil, err
	}
	return s, nil
}

// Put validates picture i and writes it into the staging buffer (mbs[].Pic is ignored, CoefOff is relative
// to coefs).  The slices are not retained.
func (s *Stage) Put(i int, pic *PicDesc, mbs []MbDesc, coefs []byte) error {
	var mp, cp unsafe.Pointer
	if len(mbs) > 0 {
		mp = unsafe.Pointer(&mbs[0])
	}
	if len(coefs) > 0 {
		cp = unsafe.Pointer(&coefs[0])
	}
	return lastError(C.mpeghip_video_stage_put(s.h, C.uint32_t(i), (*C.mpeghip_pic_desc)(unsafe.Pointer(pic)),
		(*C.mpeghip_mb_desc)(mp), cp))
}

// StageBeginSparse / PutSparse: the same stage for pictures in the sparse form, sizes in dwords
// (mpeghip_video_stage_begin_sparse / _put_sparse).
func (v *Video) StageBeginSparse(nMbs []uint32, nWords []uint64) (*Stage, error) {
	s := &Stage{}
	if len(nMbs) == 0 || len(nMbs) != len(nWords) {
		return nil, errors.New("mpeghip: StageBeginSparse: nMbs and nWords must have the same, non-zero length")
	}
	sizes := make([]C.size_t, len(nWords))
	for i, n := range nWords {
		sizes[i] = C.size_t(n)
	}
	if err := lastError(C.mpeghip_video_stage_begin_sparse(v.h, C.uint32_t(len(nMbs)), (*C.uint32_t)(unsafe.Pointer(&nMbs[0])),
		&sizes[0], &s.h)); err != nil {
		return nil, err
	}
	return s, nil
}

func (s *Stage) PutSparse(i int, pic *PicDesc, mbs []MbDesc, words []uint32) error {
	var mp, wp unsafe.Pointer
	if len(mbs) > 0 {
		mp = unsafe.Pointer(&mbs[0])
	}
	if len(words) > 0 {
		wp = unsafe.Pointer(&words[0])
	}
	return lastError(C.mpeghip_video_stage_put_sparse(s.h, C.uint32_t(i), (*C.mpeghip_pic_desc)(unsafe.Pointer(pic)),
		(*C.mpeghip_mb_desc)(mp), (*C.uint32_t)(wp)))
}

// StageBeginDevice opens a DEVICE-PACKED stage (mpeghip_video_stage_begin_device): PutSparse only copies the picture's arrays
// into pinned staging, Commit sends them as they are and the GPU validates and packs them in front of the reconstruction.
// Sparse pictures only.  ERRORS ARE DEFERRED: Commit returns nil with the work in flight; a malformed picture makes the
// whole commit reconstruct nothing and is reported once by the next call that waits for the device — Sync, ReadPlanes(Of),
// RGBA, or the StageBegin* / Submit* that reuses the commit's staging buffer (include/mpeghip.h).
func (v *Video) StageBeginDevice(nMbs []uint32, nWords []uint64) (*Stage, error) {
	s := &Stage{}
	if len(nMbs) == 0 || len(nMbs) != len(nWords) {
		return nil, errors.New("mpeghip: StageBeginDevice: nMbs and nWords must have the same, non-zero length")
	}
	sizes := make([]C.size_t, len(nWords))
	for i, n := range nWords {
		sizes[i] = C.size_t(n)
	}
	if err := lastError(C.mpeghip_video_stage_begin_device(v.h, C.uint32_t(len(nMbs)), (*C.uint32_t)(unsafe.Pointer(&nMbs[0])),
		&sizes[0], &s.h)); err != nil {
		return nil, err
	}
	s.nMbs, s.nWords = nMbs, nWords
	return s, nil
}

// Map returns picture i's arrays INSIDE the pinned staging buffer of a device-packed stage (C memory of the library, valid
// until Commit): a parser that records its descriptors and pair words there hands them over with PutMapped and no copy at all.
func (s *Stage) Map(i int) ([]MbDesc, []uint32, error) {
	var mp *C.mpeghip_mb_desc
	var wp *C.uint32_t
	if err := lastError(C.mpeghip_video_stage_map(s.h, C.uint32_t(i), &mp, &wp)); err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*MbDesc)(unsafe.Pointer(mp)), int(s.nMbs[i])), unsafe.Slice((*uint32)(unsafe.Pointer(wp)), int(s.nWords[i])), nil
}

// PutMapped marks picture i, written through Map, complete.
func (s *Stage) PutMapped(i int, pic *PicDesc) error {
	return lastError(C.mpeghip_video_stage_put_mapped(s.h, C.uint32_t(i), (*C.mpeghip_pic_desc)(unsafe.Pointer(pic))))
}

// Sync waits for everything queued on the handle and returns (once) the deferred error of a device-packed commit, if any.
func (v *Video) Sync() error { return lastError(C.mpeghip_video_sync(v.h)) }

// Verdict waits until the device has VALIDATED the device-packed commits queued so far — not until it has reconstructed them — and
// returns (once) their deferred error.  A refusal is per picture (ABI 3): the refused picture is not reconstructed, the commit's
// other pictures (other streams) are; Refused names the refused pictures (index in their commit) and their streams.
func (v *Video) Verdict() error { return lastError(C.mpeghip_video_verdict(v.h)) }
func (v *Video) Refused() (pics, streams []uint32, total uint64) {
	pics, streams = make([]uint32, 1024), make([]uint32, 1024)
	total = uint64(C.mpeghip_video_refused(v.h, (*C.uint32_t)(&pics[0]), (*C.uint32_t)(&streams[0]), 1024))
	n := total
	if n > 1024 {
		n = 1024
	}
	return pics[:n], streams[:n], total
}

// Commit sends the staged pictures and reconstructs them (asynchronous, like Submit); the Stage is over.
func (s *Stage) Commit() error {
	h := s.h
	s.h = nil
	return lastError(C.mpeghip_video_stage_commit(h))
}

func (v *Video) ReadPlanesOf(stream, slot int, y, cb, cr []byte) error {
	return lastError(C.mpeghip_video_read_planes(v.h, C.uint32_t(stream), C.uint32_t(slot),
		(*C.uint8_t)(&y[0]), (*C.uint8_t)(&cb[0]), (*C.uint8_t)(&cr[0])))
}

// OpenAudioStreams is OpenAudio for n independent streams.
func (c *Context) OpenAudioStreams(n, fmaMode int) (*Audio, error) {
	a := &Audio{}
	if err := lastError(C.mpeghip_audio_open(c.h, C.uint32_t(n), C.int(fmaMode), &a.h)); err != nil {
		return nil, err
	}
	return a, nil
}

// SynthMasked synthesises one frame of every stream with active[i] != 0 (samples: n x [2][36][32]int32,
// out: n x 2304 elements); the other streams keep their V ring and vPos.
func (a *Audio) SynthMasked(samples []int32, active []byte, format int, out unsafe.Pointer) error {
	return lastError(C.mpeghip_audio_synth_masked(a.h, (*C.int32_t)(unsafe.Pointer(&samples[0])), 1, C.int(format), out,
		(*C.uint8_t)(unsafe.Pointer(&active[0]))))
}
