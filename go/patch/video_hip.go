//go:build hip

package mpeg

import (
	"os"
	"strconv"
	"sync"

	"github.com/gen2brain/mpeg/internal/mpeghip" // = go/mpeghip of this repository, vendored
)

// One device context per process (a context is bound to one GPU + one HIP stream and is not
// thread-safe: decoders that share it must be driven from one goroutine, like the reference's own
// decoders; one process per GPU).  MPEG_HIP_DEVICE picks the ordinal.  Many streams per GPU go through
// the staged submit of go/mpeghip (Stage.Put from one goroutine per stream), INTEGRATION.md section 2.
var (
	hipOnce sync.Once
	hipCtx  *mpeghip.Context
	hipErr  error
)

func hipContext() (*mpeghip.Context, error) {
	hipOnce.Do(func() {
		dev, _ := strconv.Atoi(os.Getenv("MPEG_HIP_DEVICE"))
		hipCtx, hipErr = mpeghip.NewContext(dev)
	})
	return hipCtx, hipErr
}

// blockRec is what decodeBlock recorded for one block of the current macroblock.
type blockRec struct {
	valid    bool      // the block ended properly (an invalid block is not reconstructed, video.go:711-714)
	needsRaw bool      // cannot travel as int16 levels: see hipVideo.decodeBlock
	q        [64]int16 // quantised levels at their natural (de-zigzagged, row-major) positions; [0] = DC for intra
	touched  [64]uint8 // the positions this block wrote, in scan order
	nTouched int
	raw      [64]int32 // snapshot of blockData as idct() / the DC fast path would consume it
}

// heldSubmit is one hand-over recorded while a picture is parsed ahead.
type heldSubmit struct {
	pic   mpeghip.PicDesc
	mbs   []mpeghip.MbDesc
	words []uint32
}

// hipUndo is the parser state that outlives a picture: a look-ahead that is dropped (Rewind) puts it back.
type hipUndo struct {
	cur, fwd, bwd     Frame
	pictureType       int
	hasReferenceFrame bool
	blockDirty        bool
	blockData         [64]int
	motionForward     motion
	motionBackward    motion
	stats             [8]int
}

// hipVideo is embedded in Video (field `hip hipVideo`) under the hip tag.
type hipVideo struct {
	dev *mpeghip.Video

	mbs     []mpeghip.MbDesc // one picture's macroblocks (reused between pictures)
	words   []uint32         // the picture's coded blocks in the sparse form (mpeghip.h: MPEGHIP_PIC_SPARSE): count + pairs; snapshots as the count 64 + 64 int32
	written []bool           // macroblock address already emitted in the current submit

	// current macroblock
	active, intra, hasPred, backward bool
	outOfRange                       bool // a copyMacroblock call of this macroblock would panic in the reference
	mbX, mbY, mvX, mvY, qscale, cbp  int
	blocks                           [6]blockRec

	blockDirty   bool      // blockData holds stale coefficients of an earlier invalid block
	cur          *blockRec // the block between blockBegin and blockEnd
	dirtyAtStart bool      // blockDirty when that block began

	// Decode's look-ahead (frame_hip.go): the hand-overs of a picture parsed ahead are kept, not submitted
	deferring bool
	held      []heldSubmit
	nHeld     int
	lookahead bool // on after open; Video.SetLookahead(false): parse, submit, read back
	ahead     struct {
		valid bool
		slot  uint8
		time  float64
	}
	aheadTried, endedBeforeAhead bool
	undo                         *hipUndo // what a dropped look-ahead gives back (Rewind)
	out                          [2]Frame // the two frames Decode alternates between (valid until the next call, mpeg.go:413-415)
	outPlanes                    [2][]byte // pinned (mpeghip.Context.PinnedAlloc): luma | Cb | Cr
	outNext                      int
	mirrored                     bool // the device store keeps a host mirror (mpeghip.Video.HostMirror): Decode's frames point into it

	have     [3]bool // Frame.Y/Cb/Cr.Data of the Frame holding slot s equal the device's slot s
	NoPlanes bool    // the consumer only wants Frame.RGBA(): Decode skips the plane read-back

	Stats struct{ Pictures, Submits, Macroblocks, CodedBlocks, RawMacroblocks, InvalidBlocks, DuplicateSplits, RangeSkips int }
}

// open replaces the three initFrame calls of decodeSequenceHeader (video.go:324-326): the frame store
// lives on the device; the Frames keep host slices of the same sizes for read-back.
func (h *hipVideo) open(v *Video) bool {
	ctx, err := hipContext()
	if err != nil {
		return false // no GPU: this flavour has no CPU path
	}
	if h.dev != nil {
		h.dev.Close()
	}
	if h.dev, err = ctx.OpenVideo(v.width, v.height); err != nil {
		return false
	}
	if err = h.dev.SetQuant(&v.intraQuantMatrix, &v.nonIntraQuantMatrix); err != nil {
		return false
	}
	frames := [3]*Frame{&v.frameCurrent, &v.frameForward, &v.frameBackward}
	for s, f := range frames {
		// geometry, image headers, the RGBA buffer; Y/Cb/Cr.Data become the HOST COPY of device slot s.
		// The reference rotates Frame VALUES in decodePicture (video.go:406-409, 430-433: slice headers
		// move, bytes do not); hipSlot is part of the value, so the device slot travels with it and the
		// rotation code stays as it is.
		v.initFrame(f)
		f.hipSlot, f.hipOwner = uint8(s), v
		h.have[s] = true // zeros on both sides
	}
	h.written = make([]bool, v.mbSize)
	h.blockDirty = false
	// Decode's two frames: pinned planes the device fills itself (ReadPlanesAsync), Frame values set up like the decoder's own
	luma, chroma := v.lumaWidth*v.lumaHeight, v.chromaWidth*v.chromaHeight
	for i := range h.out {
		if h.outPlanes[i] != nil {
			ctx.PinnedFree(h.outPlanes[i])
		}
		if h.outPlanes[i] = ctx.PinnedAlloc(luma + 2*chroma); h.outPlanes[i] == nil {
			return false
		}
		f := &h.out[i]
		v.initFrame(f) // geometry, image headers, the RGBA buffer ...
		p := h.outPlanes[i]
		f.Y.Data, f.Cb.Data, f.Cr.Data = p[:luma:luma], p[luma:luma+chroma:luma+chroma], p[luma+chroma:luma+2*chroma]
		f.imYCbCr.Y, f.imYCbCr.Cb, f.imYCbCr.Cr = f.Y.Data, f.Cb.Data, f.Cr.Data // ... over the pinned planes
		f.hipOwner = v
	}
	h.lookahead, h.ahead.valid, h.nHeld, h.undo, h.aheadTried = true, false, 0, nil, false
	// the host mirror: the reconstruction launch writes every frame once more, linearly, into pinned host memory — Decode then hands
	// out that copy and queues no read-back at all (frame_hip.go); without it (no pinned memory to be had) the two frames above serve
	h.mirrored = h.dev.HostMirror(true) == nil
	return true
}

func (h *hipVideo) beginPicture(v *Video) {
	h.Stats.Pictures++
	h.mbs = h.mbs[:0]
	h.words = h.words[:0]
	for i := range h.written {
		h.written[i] = false
	}
}

// flush hands the recorded macroblocks to the device: ONE cgo call per picture (more only when a
// damaged stream addresses a macroblock twice).  Submit returns with the copy and the kernel in
// flight, so the parser works on the next picture meanwhile.
func (h *hipVideo) flush(v *Video) {
	if len(h.mbs) == 0 {
		return
	}
	// between the two halves of decodePicture's rotation frameCurrent is the destination and
	// frameForward / frameBackward are what copyMacroblock would read
	cur := v.frameCurrent.hipSlot
	pic := mpeghip.PicDesc{Cur: cur, Fwd: v.frameForward.hipSlot, Bwd: v.frameBackward.hipSlot, MbCount: uint32(len(h.mbs))}
	if h.deferring { // a picture parsed ahead (frame_hip.go: decode): its hand-over is kept; the arrays change places with the kept slot's
		if h.nHeld == len(h.held) {
			h.held = append(h.held, heldSubmit{})
		}
		k := &h.held[h.nHeld]
		h.nHeld++
		k.pic = pic
		k.mbs, h.mbs = h.mbs, k.mbs[:0]
		k.words, h.words = h.words, k.words[:0]
		h.Stats.Submits++
		h.Stats.Macroblocks += len(k.mbs)
		for i := range h.written {
			h.written[i] = false
		}
		return
	}
	if err := h.dev.SubmitSparse(&pic, h.mbs, h.words); err != nil {
		// descriptors are validated by the library; the recorder never produces an invalid one.  A
		// device failure leaves the slot as it was (the reference has no error path here either).
		_ = err
	}
	h.have[cur] = false
	h.Stats.Submits++
	h.Stats.Macroblocks += len(h.mbs)
	h.mbs = h.mbs[:0]
	h.words = h.words[:0]
	for i := range h.written {
		h.written[i] = false
	}
}

// beginMacroblock is called by decodeMacroblock once mbRow / mbCol / the type / quantizerScale of the
// macroblock are known, before predictMacroblock / the blocks — and for each skipped macroblock
// (video.go:510-517) with intra = false (v.macroblockIntra is stale there).
func (h *hipVideo) beginMacroblock(v *Video, intra bool) {
	addr := v.mbRow*v.mbWidth + v.mbCol
	if h.written[addr] { // macroblocks of one submit run concurrently: keep "last writer in bitstream order"
		h.flush(v)
		h.Stats.DuplicateSplits++
	}
	h.written[addr] = true
	h.active, h.intra, h.hasPred, h.backward, h.outOfRange = true, intra, false, false, false
	h.mbX, h.mbY, h.mvX, h.mvY, h.cbp = v.mbCol, v.mbRow, 0, 0, 0
	h.qscale = v.quantizerScale
}

// predict stands where predictMacroblock calls copyMacroblock (video.go:626-635): a later call
// overwrites an earlier one, which is exactly what the reference's second copy does to the first.
// copyMacroblock's legal read range (video_noasm.go:48-50) is [plane start, end of base); outside it the
// reference panics.  Here the WHOLE macroblock is dropped then (endMacroblock) — also when the call that would
// panic is one that a later call overwrites (a B macroblock's forward copy): the same rule as
// mpeg_amd/host/video.cpp and the oracle.
func (h *hipVideo) predict(v *Video, mh, mv int, backward bool) {
	h.hasPred, h.backward, h.mvX, h.mvY = true, backward, mh, mv
	lw, cw := v.lumaWidth, v.chromaWidth
	luma, chroma := v.lumaWidth*v.lumaHeight, v.chromaWidth*v.chromaHeight
	total := luma + 2*chroma + lw*16
	lsi := ((h.mbY<<4)+(mv>>1))*lw + (h.mbX << 4) + (mh >> 1)
	llast := lsi + (15+(mv&1))*lw + 15 + (mh & 1)
	cmh, cmv := mh/2, mv/2
	csi := ((h.mbY<<3)+(cmv>>1))*cw + (h.mbX << 3) + (cmh >> 1)
	clast := csi + (7+(cmv&1))*cw + 7 + (cmh & 1)
	if lsi < 0 || llast >= total || csi < 0 || clast >= total-luma-chroma {
		h.outOfRange = true
	}
}

func dequantPremult(level int, intra bool, qscale int, q byte, idx int) int32 {
	// video.go:719-744, on one level
	level <<= 1
	if !intra {
		if level < 0 {
			level--
		} else {
			level++
		}
	}
	level = (level * qscale * int(q)) >> 4
	if level&1 == 0 {
		if level > 0 {
			level--
		} else {
			level++
		}
	}
	if level > 2047 {
		level = 2047
	} else if level < -2048 {
		level = -2048
	}
	return int32(level * int(videoPremultiplierMatrix[idx]))
}

// ---- the hooks inside the reference's OWN decodeBlock (video.go:639-797; go/patch/PATCH.md lists where they go).  The VLC loop is
// NOT restated here (round 5 did, line for line — two loops that must read identical bits): the reference's loop runs as it is, keeps
// v.blockData exactly as the reference has it, and tells the recorder
//   blockBegin    (before the call, decodeMacroblock's block loop :556-561)  a coded block starts
//   blockLevel    (after deZigZagged is known, :716-717)                      one coded level, before it is dequantised
//   blockInvalid  (the invalid-block return, :711-714)                         nothing is reconstructed, blockData stays dirty
//   blockEnd      (in front of "Move block to its place", :747)                what idct() / the *ToDest functions would consume
// Nothing is reconstructed on the CPU: blockEnd makes decodeBlock return.
//
// v.blockData is all zero between blocks except after an invalid block (:711-714 returns before the clears): only then — or
// when this block holds an intra DC outside int16 — do its exact contents matter, and the block travels as an int32 snapshot of
// what idct() / the DC fast path would consume (MPEGHIP_MB_COEF_RAW).  Every other block travels as its (position, level) pairs.
func (h *hipVideo) blockBegin(v *Video, block int) {
	br := &h.blocks[block]
	br.valid, br.needsRaw, br.nTouched = false, false, 0
	br.q = [64]int16{}
	h.cbp |= 0x20 >> uint(block)
	h.cur = br
	h.dirtyAtStart = h.blockDirty
}

// blockLevel: video.go:716-717 — position deZigZagged (natural order, row*8+column) holds `level`, not yet dequantised (a coded
// zero level included: it travels as a pair and dequantises to +-1 like any other).
func (h *hipVideo) blockLevel(deZigZagged, level int) {
	br := h.cur
	br.q[deZigZagged] = int16(level)
	br.touched[br.nTouched] = uint8(deZigZagged)
	br.nTouched++
}

// blockInvalid: video.go:711-714 — no reconstruction, and blockData is NOT cleared.
func (h *hipVideo) blockInvalid(v *Video) {
	h.Stats.InvalidBlocks++
	h.blockDirty = true
}

// blockEnd: video.go:747 — the block's levels are in blockData (dequantised and premultiplied by the reference's own code,
// :719-744; an intra block's DC as predictor << 8, :656-672), n is the reference's scan position.  Records the block and clears
// blockData the way the reference's write-back does (:774-796).
func (h *hipVideo) blockEnd(v *Video, block, n int) {
	br := h.cur
	if v.macroblockIntra { // the DC level: the predictor the reference just saved (:669)
		plane := 0
		if block > 3 {
			plane = block - 3
		}
		dc := v.dcPredictor[plane]
		switch {
		case dc < -32768:
			br.needsRaw, br.q[0] = true, -32768
		case dc > 32767:
			br.needsRaw, br.q[0] = true, 32767
		default:
			br.q[0] = int16(dc)
		}
	}
	clear := func() { // video.go:774-777 / 787-790: only blockData[0] is used, and only it is cleared; :781-783 / 794-796: all of it
		if n == 1 {
			v.blockData[0] = 0
			return
		}
		for i := 0; i < 64; i++ {
			v.blockData[i] = 0
		}
	}
	if !h.dirtyAtStart && !br.needsRaw {
		clear() // the common case: nothing of blockData survives this block
		br.valid = true
		return
	}
	br.needsRaw = true
	clamp := func(x int) int32 { // beyond +-2^30 the pixel saturates whatever else is added: the clamp is exact, the snapshot int32
		if x > 1<<30 {
			return 1 << 30
		} else if x < -(1 << 30) {
			return -(1 << 30)
		}
		return int32(x)
	}
	br.raw = [64]int32{}
	if n == 1 {
		br.raw[0] = clamp(v.blockData[0])
	} else {
		for i := 0; i < 64; i++ {
			if n < 10 && (i>>3 >= 4 || i&7 >= 4) { // video.go:807-866: the reduced idct ignores rows, columns >= 4
				continue
			}
			br.raw[i] = clamp(v.blockData[i])
		}
	}
	clear()
	h.blockDirty = false
	for i := 0; i < 64; i++ {
		if v.blockData[i] != 0 {
			h.blockDirty = true
			break
		}
	}
	br.valid = true
}

// endMacroblock appends the macroblock's descriptor and its coded blocks: per block a count word and one pair per coded
// level, as the VLC loop produced them (an intra block's DC first; a coded zero level stays a pair); the blocks of a
// macroblock that travels raw as the count word 64 and a snapshot of 64 int32 values, column-major.
func (h *hipVideo) endMacroblock(v *Video) {
	if !h.active {
		return
	}
	h.active = false
	if !h.intra && !h.hasPred {
		return // cannot happen: every non-intra macroblock is predicted (video.go:543-544)
	}
	if !h.intra && h.outOfRange {
		h.Stats.RangeSkips++ // the reference panics here
		return
	}
	raw, cbp := false, 0
	for b := 0; b < 6; b++ {
		if h.cbp&(0x20>>uint(b)) != 0 && h.blocks[b].valid {
			cbp |= 0x20 >> uint(b)
			raw = raw || h.blocks[b].needsRaw
		}
	}
	if h.qscale < 1 && cbp != 0 {
		raw = true // quantiser_scale 0 (forbidden): keep the reference's arithmetic
	}
	d := mpeghip.MbDesc{MbX: uint16(h.mbX), MbY: uint16(h.mbY), MvX: int16(h.mvX), MvY: int16(h.mvY), Cbp: uint8(cbp)}
	switch {
	case h.intra:
		d.Flags = mpeghip.MbIntra
	case h.backward:
		d.Flags = mpeghip.MbRefBwd
	default:
		d.Flags = mpeghip.MbRefFwd
	}
	if raw {
		d.Flags |= mpeghip.MbCoefRaw
		h.Stats.RawMacroblocks++
	}
	qs := h.qscale
	if qs < 1 {
		qs = 1
	} else if qs > 31 {
		qs = 31
	}
	d.Qscale = uint8(qs)
	d.CoefOff = uint32(len(h.words)) // dwords

	quant := &v.nonIntraQuantMatrix
	if h.intra {
		quant = &v.intraQuantMatrix
	}
	for b := 0; b < 6; b++ {
		if cbp&(0x20>>uint(b)) == 0 {
			continue
		}
		br := &h.blocks[b]
		h.Stats.CodedBlocks++
		if raw {
			var snap [64]int32
			if br.needsRaw {
				snap = br.raw
			} else { // a clean block of a macroblock that travels raw: dequantise it here — its CODED levels, zeros included
				for k := 0; k < br.nTouched; k++ {
					i := int(br.touched[k])
					snap[i] = dequantPremult(int(br.q[i]), h.intra, h.qscale, quant[i], i)
				}
				if h.intra {
					snap[0] = int32(br.q[0]) << 8
				}
			}
			h.words = append(h.words, 64) // every block of the sparse form begins with its count word: 64 for a snapshot
			at := len(h.words)
			h.words = append(h.words, make([]uint32, 64)...)
			for i := 0; i < 64; i++ {
				h.words[at+(i&7)*8+(i>>3)] = uint32(snap[i])
			}
		} else {
			n := br.nTouched
			if h.intra {
				n++
			}
			h.words = append(h.words, uint32(n))
			if h.intra {
				h.words = append(h.words, mpeghip.Pair(int(br.q[0]), 0))
			}
			for k := 0; k < br.nTouched; k++ {
				i := int(br.touched[k]) // natural index row*8+column -> position column*8+row
				h.words = append(h.words, mpeghip.Pair(int(br.q[i]), (i&7)*8+(i>>3)))
			}
		}
	}
	h.mbs = append(h.mbs, d)
}
