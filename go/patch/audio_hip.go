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
	// the frame's requantised sub-band samples in the device layout [ch][t][sb], t = (part*4 + granule)*3 + p
	frame [2][36][32]int32
	// MPEGHIP_AUDIO_FMA_WINDOW reproduces the amd64 AVX2 flavour (golden hash 0x50f3ab75f5fb0fb5,
	// mpeg_test.go:195); the default is the pure-Go / SSE2 arithmetic (0xf1b76cdf8e6cdea5, :194)
	FMAWindow bool
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
	h.dev, err = ctx.OpenAudio(mode)
	return err == nil
}

// record stands where the synthesis loop starts (audio.go:378), once per (part, granule): t0 is the
// index of the granule's first sub-block, (part*4 + granule) * 3.  a.sample[ch][sb][p] holds the three
// sub-blocks readSamples just produced (audio.go:440-490).
func (h *hipAudio) record(a *Audio, t0 int) {
	for ch := 0; ch < 2; ch++ {
		for p := 0; p < 3; p++ {
			row := &h.frame[ch][t0+p]
			for sb := 0; sb < 32; sb++ {
				row[sb] = int32(a.sample[ch][sb][p])
			}
		}
	}
}

// synth is called once at the end of decodeFrame: one cgo call per audio frame writes the 1152 sample
// pairs in the decoder's format straight into a.samples.
func (h *hipAudio) synth(a *Audio) {
	var out unsafe.Pointer
	switch a.format {
	case AudioF32N:
		out = unsafe.Pointer(&a.samples.Interleaved[0])
	case AudioF32:
		out = unsafe.Pointer(&a.samples.F32[0])
	case AudioS16:
		out = unsafe.Pointer(&a.samples.S16[0])
	case AudioF32NLR: // planar: 1152 L then 1152 R
		var lr [2 * SamplesPerFrame]float32
		if h.dev.Synth(&h.frame, int(a.format), unsafe.Pointer(&lr[0])) == nil {
			copy(a.samples.Left, lr[:SamplesPerFrame])
			copy(a.samples.Right, lr[SamplesPerFrame:])
		}
		return
	}
	_ = h.dev.Synth(&h.frame, int(a.format), out)
}
