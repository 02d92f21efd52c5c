//go:build !hip

package mpeg

import "image"

// The default flavours (noasm / amd64 / arm64) are untouched: hipBuild is a constant, so every
// `if hipBuild { ... }` hook in video.go / audio.go is dead code the compiler drops, and the recorder types
// below are empty.
const hipBuild = false

type hipVideo struct{}

func (h *hipVideo) open(v *Video) bool                              { return true }
func (h *hipVideo) beginPicture(v *Video)                           {}
func (h *hipVideo) flush(v *Video)                                  {}
func (h *hipVideo) beginMacroblock(v *Video, intra bool)            {}
func (h *hipVideo) endMacroblock(v *Video)                          {}
func (h *hipVideo) predict(v *Video, mh, mv int, backward bool)     {}
func (h *hipVideo) blockBegin(v *Video, block int)                  {}
func (h *hipVideo) blockLevel(deZigZagged, level int)               {}
func (h *hipVideo) blockInvalid(v *Video)                           {}
func (h *hipVideo) blockEnd(v *Video, block, n int)                 {}
func (h *hipVideo) decode(v *Video) *Frame                          { return nil }
func (h *hipVideo) dropLookahead(v *Video)                          {}
func (h *hipVideo) time(v *Video) float64                           { return 0 }
func (h *hipVideo) hasEnded(v *Video) bool                          { return false }
func (h *hipVideo) setTime(v *Video)                                {}

type hipAudio struct{}

func (h *hipAudio) open(a *Audio) bool                { return true }
func (h *hipAudio) record(a *Audio, t0 int)           {}
func (h *hipAudio) synth(a *Audio)                    {}
func (h *hipAudio) decode(a *Audio) *Samples          { return nil }
func (h *hipAudio) dropLookahead()                    {}
func (h *hipAudio) time(a *Audio) float64             { return 0 }
func (h *hipAudio) hasEnded(a *Audio) bool            { return false }
func (h *hipAudio) setTime(a *Audio)                  {}

func (f *Frame) hipRGBA() *image.RGBA { return &f.imRGBA }
