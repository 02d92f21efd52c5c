//go:build hip

package mpeg

import (
	"unsafe"

	"github.com/gen2brain/mpeg/internal/mpeghip"
)

// hipAudio is embedded in Audio (field `hip hipAudio`) under the hip tag.  It replaces the synthesis
// loop of decodeFrame (audio.go:378-422): idct36 (audio.go:492-772), synthWindow (audio_noasm.go:8-38 /
// audio_amd64.s / audio_arm64.s) and the output scaling of all four formats (audio.go:386-418).
// Audio.v and Audio.vPos live on the device (zero at open, NOT cleared by Rewind, as in the reference:
// audio.go:149).
type hipAudio struct {
	dev *mpeghip.Audio
	// MPEGHIP_AUDIO_FMA_WINDOW reproduces the amd64 AVX2 flavour (golden hash 0x50f3ab75f5fb0fb5,
	// mpeg_test.go:195); the default is the pure-Go / SSE2 arithmetic (0xf1b76cdf8e6cdea5, :194)
	FMAWindow bool

	// Decode works ONE FRAME AHEAD on the host (decode below; the transliteration of mpeg::Audio::Decode, mpeg_amd/host/audio.cpp):
	// the frame's requantised sub-band samples in the device layout [ch][t][sb], t = (part*4 + granule)*3 + p, are recorded into one
	// of two PINNED buffers (mpeghip.Context.PinnedAlloc: the kernel reads them in place), its output lands in one of two pinned
	// buffers and is copied into one of two Samples — valid until the next Decode call (mpeg.go:435-437)
	in        [2][]byte // 2 x 36 x 32 int32 each
	inNext    int       // the buffer the frame being parsed records into
	parsed    int       // ... the one the last parsed frame is in
	pinnedOut [2][]byte // 2304 elements of the format's type each (float32 at most)
	out       [2]Samples
	lookahead bool
	ahead     struct {
		valid, failed bool
		buf           int
		time          float64
	}
	aheadTried, endedBeforeAhead bool
}

func (h *hipAudio) open(a *Audio) bool {
	ctx, err := hipContext()
	if err != nil {
		return false
	}
	mode := mpeghip.FMANone
	if h.FMAWindow {
		mode = mpeghip.FMAWindow
	}
	if h.dev, err = ctx.OpenAudio(mode); err != nil {
		return false
	}
	for i := 0; i < 2; i++ {
		h.in[i], h.pinnedOut[i] = ctx.PinnedAlloc(2*36*32*4), ctx.PinnedAlloc(2*SamplesPerFrame*4)
		if h.in[i] == nil || h.pinnedOut[i] == nil {
			return false
		}
		o := &h.out[i]
		o.S16, o.F32 = make([]int16, SamplesPerFrame*2), make([]float32, SamplesPerFrame*2)
		o.Left, o.Right = make([]float32, SamplesPerFrame), make([]float32, SamplesPerFrame)
		o.Interleaved = make([]float32, SamplesPerFrame*2)
	}
	h.lookahead = true
	return true
}

// record stands where the synthesis loop starts (audio.go:378), once per (part, granule): t0 is the
// index of the granule's first sub-block, (part*4 + granule) * 3.  a.sample[ch][sb][p] holds the three
// sub-blocks readSamples just produced (audio.go:440-490).
func (h *hipAudio) record(a *Audio, t0 int) {
	frame := (*[2][36][32]int32)(unsafe.Pointer(&h.in[h.inNext][0]))
	for ch := 0; ch < 2; ch++ {
		for p := 0; p < 3; p++ {
			row := &frame[ch][t0+p]
			for sb := 0; sb < 32; sb++ {
				row[sb] = int32(a.sample[ch][sb][p])
			}
		}
	}
}

// synth is called once at the end of decodeFrame (audio.go:426): the frame is complete in h.in[h.inNext]; the synthesis itself is
// queued by decode, which may be a call later.
func (h *hipAudio) synth(a *Audio) {
	h.parsed = h.inNext
	h.inNext ^= 1
}

// decode is Audio.Decode under the hip tag (PATCH.md, audio.go edit 3: the reference's Decode body becomes decodeNow, and
// `func (a *Audio) Decode() *Samples { if hipBuild { return a.hip.decode(a) }; return a.decodeNow() }`): a call hands the frame
// parsed during the previous call to the device (one launch: the kernel reads and writes pinned memory in place), parses the NEXT
// frame while the device works — held back until the next call, so the V ring, which Rewind keeps (audio.go:149-154), is never
// ahead of the frames returned — and only then waits.
func (h *hipAudio) decode(a *Audio) *Samples {
	var b int
	var t float64
	h.aheadTried = false
	switch {
	case h.ahead.failed: // the attempt this call stands for was made early, and found no frame (it consumed what it consumed)
		h.ahead.failed = false
		return nil
	case h.ahead.valid:
		b, t = h.ahead.buf, h.ahead.time
		h.ahead.valid = false
	default:
		s := a.decodeNow() // audio.go:163-182 as it is; decodeFrame's hooks record the frame
		if s == nil {
			return nil
		}
		b, t = h.parsed, s.Time
	}
	ticket, err := h.dev.SynthAsync(h.in[b], int(a.format), h.pinnedOut[b])
	if h.lookahead {
		// An attempt that fails for lack of data consumes nothing and is made again by the next call; one that fails on a bad header
		// HAS consumed bits (decodeHeader's hunt for a frame sync, audio.go:184-272): it is the next call's attempt, made early.
		before := a.buf.bitIndex
		h.endedBeforeAhead, h.aheadTried = a.buf.HasEnded(), true
		if s := a.decodeNow(); s != nil {
			h.ahead.valid, h.ahead.buf, h.ahead.time = true, h.parsed, s.Time
		} else {
			h.ahead.failed = a.nextFrameDataSize == 0 && a.buf.bitIndex != before
		}
	}
	o := &h.out[b]
	if err == nil && h.dev.SynthWait(ticket) == nil {
		p := unsafe.Pointer(&h.pinnedOut[b][0])
		switch a.format {
		case AudioF32N:
			copy(o.Interleaved, unsafe.Slice((*float32)(p), 2*SamplesPerFrame))
		case AudioF32:
			copy(o.F32, unsafe.Slice((*float32)(p), 2*SamplesPerFrame))
		case AudioS16:
			copy(o.S16, unsafe.Slice((*int16)(p), 2*SamplesPerFrame))
		case AudioF32NLR: // planar: 1152 L then 1152 R
			lr := unsafe.Slice((*float32)(p), 2*SamplesPerFrame)
			copy(o.Left, lr[:SamplesPerFrame])
			copy(o.Right, lr[SamplesPerFrame:])
		}
	}
	o.Time, o.format = t, a.format
	return o
}

// dropLookahead is the first line of Audio.Rewind (audio.go:149): a frame parsed ahead was never synthesised.
func (h *hipAudio) dropLookahead() { h.ahead.valid, h.ahead.failed, h.aheadTried = false, false, false }

// time / hasEnded / setTime stand in Audio.Time (audio.go:136), Audio.HasEnded (:156) and behind Audio.SetTime's assignments (:143)
func (h *hipAudio) time(a *Audio) float64 {
	if h.ahead.valid {
		return h.ahead.time
	}
	return a.time
}
func (h *hipAudio) hasEnded(a *Audio) bool {
	switch {
	case h.ahead.valid:
		return false
	case h.aheadTried:
		return h.endedBeforeAhead
	}
	return a.buf.HasEnded()
}
func (h *hipAudio) setTime(a *Audio) {
	if h.ahead.valid {
		h.ahead.time = a.time
		a.samplesDecoded += SamplesPerFrame
		a.time = float64(a.samplesDecoded) / float64(samplerate[a.samplerateIndex])
	}
}
