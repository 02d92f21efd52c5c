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
func (h *hipVideo) decodeBlock(v *Video, block int)                 {}
func (h *hipVideo) fetch(v *Video, f *Frame)                        {}

type hipAudio struct{}

func (h *hipAudio) open(a *Audio) bool                { return true }
func (h *hipAudio) record(a *Audio, t0 int)           {}
func (h *hipAudio) synth(a *Audio)                    {}

func (f *Frame) hipRGBA() *image.RGBA { return &f.imRGBA }
