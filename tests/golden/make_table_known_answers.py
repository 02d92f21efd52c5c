#!/usr/bin/env python3
"""The reference's constant tables as numbers: ISO 11172-2 picture rates / zig-zag scan / default intra matrix / the IDCT's
premultiplier matrix (video.go:1034-1086) and the ISO 11172-3 Layer II header, allocation and quantiser tables and the synthesis
window (audio.go:798-973; the MPEG-2 LSF rows, which neither the reference's decoder nor this build reaches — audio.go:217-221
rejects everything but MPEG-1 —, are left out).  Run in the build container, where /root/reference is mounted; the result is
data, committed as tests/golden/table_known_answers.json; tests/test_table_known_answers.py compares the tables in the product's
sources and in the oracle's with it (the golden streams run through one bit rate, one sample rate and a handful of allocation
rows: this covers the entries they do not).

    python tests/golden/make_table_known_answers.py [/root/reference]"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")


def block(text, var):
    m = re.search(r"var %s = [^\n{]*\{(.*?)\n\}" % var, text, re.S)
    assert m, var
    return re.sub(r"//[^\n]*", "", m.group(1))


def numbers(body, env=None):
    out = []
    for tok in re.findall(r"[-+]?(?:0x[0-9a-fA-F]+|\d+\.\d*|\d+)|[A-Za-z_]\w*", body):
        out.append(env[tok] if env and tok in env else (float(tok) if "." in tok else int(tok, 0)))
    return out


def rows(body, env=None):
    """a [][]T literal: the inner brace groups, one list each"""
    return [numbers(r, env) for r in re.findall(r"\{([^{}]*)\}", body)]


video = (REF / "video.go").read_text()
audio = (REF / "audio.go").read_text()
env = {m.group(1): eval(m.group(2)) for m in re.finditer(r"var (quantTab[A-D]) = byte\(([^)]*)\)", audio)}
t = {
    "source": "gen2brain/mpeg video.go:1034-1086, audio.go:798-973 (MPEG-1 rows)",
    "video_picture_rate": numbers(block(video, "videoPictureRate")),
    "video_zigzag": numbers(block(video, "videoZigZag")),
    "video_intra_quant_matrix": numbers(block(video, "videoIntraQuantMatrix")),
    "video_non_intra_quant_matrix": numbers(block(video, "videoNonIntraQuantMatrix")),
    "video_premultiplier_matrix": numbers(block(video, "videoPremultiplierMatrix")),
    "audio_samplerate": numbers(block(audio, "samplerate"))[:4],
    "audio_bitrate": numbers(block(audio, "bitrate"))[:14],
    "audio_scalefactor_base": numbers(block(audio, "scalefactorBase")),
    "audio_synthesis_window": numbers(block(audio, "synthesisWindow")),
    "audio_quant_lut_step1": rows(block(audio, "quantLutStep1")),
    "audio_quant_lut_step2": rows(block(audio, "quantLutStep2"), env),
    "audio_quant_lut_step3": rows(block(audio, "quantLutStep3"))[:2],
    "audio_quant_lut_step4": rows(block(audio, "quantLutStep4")),
    "audio_quant_tab": rows(block(audio, "quantTab")),
}
for k, v in t.items():
    if k != "source":
        print("%-30s %s" % (k, ("%d rows of %s" % (len(v), [len(r) for r in v])) if isinstance(v[0], list) else "%d values" % len(v)))
assert len(t["audio_synthesis_window"]) == 512 and len(t["video_zigzag"]) == 64 and len(t["audio_quant_tab"]) == 17
out = Path(__file__).resolve().parent / "table_known_answers.json"
out.write_text(json.dumps(t, separators=(",", ":")) + "\n")
print("wrote", out, out.stat().st_size, "bytes")
