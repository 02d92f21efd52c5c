//go:build hip

package mpeg

// hipBuild selects the recording flavour of the decoder at compile time: the hot functions of the
// reference (copyMacroblock, the dequantise + idct + *ToDest tail of decodeBlock, idct36 + synthWindow +
// output scaling) do not run on the CPU; the parser records what they would have been given and
// libmpeghip reconstructs on the GPU, one cgo call per picture / audio frame.  There is no CPU
// fallback in this flavour: without a gfx950 device the decoder constructors fail.
//
// NOT COMPILED: the image this repository is built in has no Go toolchain.  The same restructuring is
// implemented and tested in C++ (mpeg_amd/host/video.cpp, audio.cpp), of which these files are the
// transliteration; see PATCH.md for the edits to the reference's own files.
const hipBuild = true
