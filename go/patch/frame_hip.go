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

// fetch is called by Video.Decode on the frame it is about to return (video.go:247-262): the planes
// are read back from the device into the Frame's own slices (one D2H copy through a pinned bounce
// buffer; synchronises with the reconstruction of that slot).  Go cannot intercept the access to
// Frame.Y.Data, so the copy is eager unless the consumer opted out (Video.hip.NoPlanes, for players
// that only call Frame.RGBA()).
func (h *hipVideo) fetch(v *Video, f *Frame) {
	if h.NoPlanes || h.have[f.hipSlot] {
		return
	}
	if err := h.dev.ReadPlanes(int(f.hipSlot), f.Y.Data, f.Cb.Data, f.Cr.Data); err == nil {
		h.have[f.hipSlot] = true
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
