//go:build hip

package mpeg

import "image"

// Frame gets two fields in video.go (both flavours; unused without the tag):
//
//	hipSlot  uint8   // device slot whose host copy Y/Cb/Cr.Data are
//	hipOwner *Video
//
// A *Frame returned by Video.Decode aliases decoder-owned storage that is valid until the next Decode
// (mpeg.go:413-415), exactly as in the reference; what changes is where the bytes come from.

// decode is Video.Decode under the hip tag (PATCH.md, video.go edit 7: the reference's Decode body becomes decodeNow, and
// `func (v *Video) Decode() *Frame { if hipBuild { return v.hip.decode(v) }; return v.decodeNow() }`).  It works ONE PICTURE AHEAD on
// the host — the transliteration of mpeg::Video::Decode (mpeg_amd/host/video.cpp), which is compiled and tested
// (tests/test_host_lookahead.py, tests/test_gpu_lookahead.py):
//
//	1  the hand-overs of the picture parsed during the previous call go to the device (replayHeld)
//	2  the frame this call returns is the slot's copy in the device store's HOST MIRROR, which the reconstruction launch itself
//	   keeps (mpeghip.Video.MirrorAsync: nothing is queued but a completion ticket) — or, without a mirror, a read-back queued
//	   behind them, asynchronously, into one of two pinned frames
//	3  the NEXT picture is parsed while the device works; its hand-overs are kept (flush, h.deferring), so the device's frame
//	   store is never ahead of the frames returned: Rewind / Seek drop the parsed picture (dropLookahead) and everything is as
//	   the reference has it — Frame.RGBA() of the frame in hand still finds its slot intact
//	4  only then the wait for (2)
//
// The returned *Frame is one of h.out[0..1], alternating; with the mirror its planes are the SLOT's copy — one of three, as the
// reference's returned *Frame is one of its three frames: valid until the next Decode call (mpeg.go:413-415) either way.  Go cannot
// intercept the access to Frame.Y.Data, so the read-back is eager unless the consumer opted out (Video.hip.NoPlanes, for players
// that only call Frame.RGBA()).
func (h *hipVideo) decode(v *Video) *Frame {
	var slot uint8
	var t float64
	h.aheadTried = false
	h.replayHeld(v)
	if h.ahead.valid {
		slot, t = h.ahead.slot, h.ahead.time
		h.ahead.valid = false
	} else {
		f := v.decodeNow() // video.go:209-268 as it is: parse + (through the hooks) submit
		if f == nil {
			return nil
		}
		slot, t = f.hipSlot, f.Time
	}
	b := h.outNext
	h.outNext ^= 1
	var ticket uint64
	var err error
	planes := h.outPlanes[b]
	if !h.NoPlanes {
		if h.mirrored {
			planes, ticket, err = h.dev.MirrorAsync(int(slot))
		} else {
			ticket, err = h.dev.ReadPlanesAsync(int(slot), planes)
		}
	}
	if h.lookahead {
		h.parseAhead(v)
	}
	if !h.NoPlanes && err == nil {
		_ = h.dev.ReadWait(ticket)
	}
	out := &h.out[b]
	out.Time, out.hipSlot = t, slot
	if h.mirrored && !h.NoPlanes && err == nil { // the frame's planes ARE the slot's copy in the mirror
		luma, chroma := v.lumaWidth*v.lumaHeight, v.chromaWidth*v.chromaHeight
		out.Y.Data, out.Cb.Data, out.Cr.Data = planes[:luma:luma], planes[luma:luma+chroma:luma+chroma], planes[luma+chroma:luma+2*chroma]
		out.imYCbCr.Y, out.imYCbCr.Cb, out.imYCbCr.Cr = out.Y.Data, out.Cb.Data, out.Cr.Data
	}
	return out
}

func (h *hipVideo) parseAhead(v *Video) {
	h.undo = &hipUndo{cur: v.frameCurrent, fwd: v.frameForward, bwd: v.frameBackward, pictureType: v.pictureType,
		hasReferenceFrame: v.hasReferenceFrame, blockDirty: h.blockDirty, blockData: v.blockData,
		motionForward: v.motionForward, motionBackward: v.motionBackward,
		stats: [8]int{h.Stats.Pictures, h.Stats.Submits, h.Stats.Macroblocks, h.Stats.CodedBlocks, h.Stats.RawMacroblocks,
			h.Stats.InvalidBlocks, h.Stats.DuplicateSplits, h.Stats.RangeSkips}}
	h.endedBeforeAhead, h.aheadTried = v.buf.HasEnded(), true
	h.deferring = true
	f := v.decodeNow()
	h.deferring = false
	if f != nil {
		h.ahead.valid, h.ahead.slot, h.ahead.time = true, f.hipSlot, f.Time
	} else if h.nHeld == 0 {
		h.undo = nil // nothing was consumed that a Rewind would have to give back
	}
}

func (h *hipVideo) replayHeld(v *Video) {
	h.undo = nil
	for i := 0; i < h.nHeld; i++ {
		k := &h.held[i]
		_ = h.dev.SubmitSparse(&k.pic, k.mbs, k.words)
		h.have[k.pic.Cur] = false
	}
	h.nHeld = 0
}

// dropLookahead is the first line of Video.Rewind (video.go:195; PATCH.md edit 8): a picture parsed ahead has not reached the
// device — forgetting it takes the kept hand-overs and the parser state that outlives a picture.
func (h *hipVideo) dropLookahead(v *Video) {
	h.ahead.valid, h.aheadTried, h.nHeld = false, false, 0
	u := h.undo
	if u == nil {
		return
	}
	h.undo = nil
	v.frameCurrent, v.frameForward, v.frameBackward = u.cur, u.fwd, u.bwd
	v.pictureType, v.hasReferenceFrame = u.pictureType, u.hasReferenceFrame
	h.blockDirty, v.blockData = u.blockDirty, u.blockData
	v.motionForward, v.motionBackward = u.motionForward, u.motionBackward
	h.Stats.Pictures, h.Stats.Submits, h.Stats.Macroblocks, h.Stats.CodedBlocks = u.stats[0], u.stats[1], u.stats[2], u.stats[3]
	h.Stats.RawMacroblocks, h.Stats.InvalidBlocks, h.Stats.DuplicateSplits, h.Stats.RangeSkips = u.stats[4], u.stats[5], u.stats[6], u.stats[7]
}

// time / hasEnded / setTime stand in Video.Time (video.go:183), Video.HasEnded (:203) and Video.SetTime (:189) (PATCH.md edit 8): a
// picture parsed ahead is the reference's NEXT picture — its time is the decoder's time, the stream has not ended while it waits
// to be returned, and an attempt to parse ahead that found nothing is the next call's business (until then HasEnded answers what
// it answered before the attempt).
func (h *hipVideo) time(v *Video) float64 {
	if h.ahead.valid {
		return h.ahead.time
	}
	return v.time
}
func (h *hipVideo) hasEnded(v *Video) bool {
	switch {
	case h.ahead.valid:
		return false
	case h.aheadTried:
		return h.endedBeforeAhead
	}
	return v.buf.HasEnded()
}
func (h *hipVideo) setTime(v *Video) { // (after the reference's two assignments)
	if h.ahead.valid {
		h.ahead.time = v.time
		v.framesDecoded++
		v.time = float64(v.framesDecoded) / v.frameRate
	}
}

// hipRGBA is Frame.RGBA() under the hip tag (video.go:31-36 calls it when hipBuild): the colour
// conversion of Go's image/draw (YCbCr 4:2:0 -> RGBA, JFIF full range, alpha 255) runs on the device,
// bit-identical for every (Y, Cb, Cr) (tests/test_gpu_video.py::test_rgba_of_every_possible_pixel), and only width*height*4 bytes cross
// PCIe — the planes need not have been read back at all.
func (f *Frame) hipRGBA() *image.RGBA {
	if v := f.hipOwner; v != nil && v.hip.dev != nil {
		_ = v.hip.dev.RGBA(int(f.hipSlot), f.imRGBA.Pix)
	}
	return &f.imRGBA
}
