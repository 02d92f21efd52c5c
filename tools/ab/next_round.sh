#!/bin/bash
# Cache-policy experiments left for the next round (the reconstruction kernel turned out to be bound by the memory
# system — profiles/r4x, r4z — so what is worth a place in L2 is the open question).  Builds one library per variant into
# tools/ab/ (run here, hipcc cross-compiles; ONE group at a time — the A/B scripts run every library they find there), then on the GPU box:
#   PROFILES="typical dense" bash tools/gpu_ab_lib.sh r5a 4                 # nt_entries against the product
#   PROFILES="typical dense" bash tools/gpu_ab_lib.sh r5b 4 --rgba 1        # nt_rgba_fused, nt_frame_fused, nt_fused_both
#   bash tools/ab/audio_ab.sh                                               # nt_audio_out, nt_audio_in (3 rounds)
# Variants:  nt_entries      coefficient entries and block words loaded with `nt` (read once)
#            dense_med3      dense units: "(l - (l > 0)) | 1" as v_med3_i32(l - 1, l, 0) | 1 (one 4-clock instruction instead of two)
#            nt_rgba_fused   the fused instance's RGBA stores non-temporal
#            nt_frame_fused  the fused instance's frame stores non-temporal (the plain instance's already are)
#            nt_fused_both   both
#            nt_audio_out    audio output samples stored non-temporally
#            nt_audio_in     audio sub-band samples loaded (direct to LDS) with `nt`
#            chroma_pairs    frame-store layout: Cb | Cr of a macroblock side by side (one line instead of half of two;
#                            ~10 instead of ~12 cache lines per prediction window).  Host packer and kernels change
#                            together; the lane emulator built with the option is bit-exact (tests/test_kernel_emu_layouts.py).
#                            On the box:  PROFILES="typical dense" bash tools/gpu_ab_lib.sh r5d 4 ; then with --rgba 1 ;
#                            parity of the whole -m gpu suite against it:  cp tools/ab/libmpeghip_chroma_pairs.so mpeg_amd/libmpeghip.so && python -m pytest tests -m gpu -q
#            tile16          the wave's coefficient tile as int16 dequantised levels (premultiplied at the column read), the
#                            8x8 transposition between the IDCT passes across lanes (DPP) instead of through LDS, snapshot
#                            blocks read straight from HBM: 4 672 B of LDS and 64 vector registers = 8 waves per SIMD instead
#                            of 7.  Each chunk works out its lane constants again (49 registers, no scratch; -DMPG_LANE_ONCE
#                            keeps them across both chunks as the product does: 64 registers and three spills).  Bit-exact in the lane emulator
#                            (tests/test_kernel_emu_layouts.py); the DPP controls follow rocPRIM's use (row_shr:n = from lane - n).
#                            tile16_asm: the two in-quad exchange steps as 8 v_cndmask_b32_dpp each (inline asm) instead of
#                            the compiler's 8 DPP moves + 8 selects (28 instead of 44 vector instructions per transposition).
#                            On the box: parity first (cp ... && pytest -m gpu), then gpu_ab_lib.sh typical dense, --rgba 1
# Everything at once: `bash tools/ab/next_round.sh all` here, then ONE call on the box: `bash tools/ab/next_round_gpu.sh r5`
# (parity of every variant first, then the interleaved A/Bs; ~6 minutes).
set -eu
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -I include -I mpeg_amd/csrc"
build() { name=$1; shift; /opt/rocm/bin/hipcc $FLAGS "$@" mpeg_amd/csrc/mpeghip.hip -o tools/ab/libmpeghip_$name.so && echo built $name; }
case "${1:-video}" in
  video) build nt_entries -DMPG_NT_ENTRIES; build dense_med3 -DMPG_DENSE_MED3 ;;
  fused) build nt_rgba_fused -DMPG_NT_RGBA_FUSED; build nt_frame_fused -DMPG_NT_FRAME_FUSED; build nt_fused_both -DMPG_NT_RGBA_FUSED -DMPG_NT_FRAME_FUSED ;;
  layout) build chroma_pairs -DMPG_CHROMA_PAIRS=1 ;;
  tile16) build tile16 -DMPG_TILE16=1; build tile16_asm -DMPG_TILE16=1 -DMPG_TRANSPOSE_ASM; build tile16_chroma_pairs -DMPG_TILE16=1 -DMPG_CHROMA_PAIRS=1 ;;
  audio) build nt_audio_out -DMPG_NT_AUDIO_OUT; build nt_audio_in -DMPG_NT_AUDIO_IN ;;
  all) "$0" video; "$0" fused; "$0" layout; "$0" tile16 ;;   # (audio apart: tools/ab/audio_ab.sh runs every library it finds on the audio leg)
  *) echo "usage: $0 video|fused|layout|tile16|audio|all"; exit 2 ;;
esac
