#!/usr/bin/env python3
"""Known answers for the nine ISO 11172-2 variable-length code tables, taken from the REFERENCE's own decoder: every path through
each of its code trees (video.go:1088-1419, walked the way buffer.go:352-376 `readVlc` / `readVlcUint` walk them: state =
table[state.Index + bit] until Index <= 0) as the bit string consumed, the value returned, and whether the walk ended in one of the
tree's dead ends (Index -1: no valid code starts like that; the reference returns that entry's Value, 0, having consumed the bits).

Run in the build container, where /root/reference is mounted; the result is data (codes and values), committed as
tests/golden/vlc_known_answers.json and compared by tests/test_vlc_known_answers.py with the code lists the product's parser
(mpeg_amd/host/iso11172_vlc_codes.h) and the oracle (oracle/iso11172_vlc_codes.h) are built from — which are ONE transcription of
Annex B shared by checker and product: this fixture is what pins that transcription to the reference for the
codes no golden stream exercises.

    python tests/golden/make_vlc_known_answers.py [/root/reference/video.go]"""
import json
import re
import sys
from pathlib import Path

SRC = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/video.go")
TABLES = {  # the reference's variable -> the code list's name (without its orc_ / mpg_ prefix)
    "videoMacroblockAddressIncrement": "vlc_mba_increment",
    "videoMacroblockTypeIntra": "vlc_mb_type_i",
    "videoMacroblockTypePredictive": "vlc_mb_type_p",
    "videoMacroblockTypeB": "vlc_mb_type_b",
    "videoCodeBlockPattern": "vlc_coded_block_pattern",
    "videoMotion": "vlc_motion_code",
    "videoDctSizeLuminance": "vlc_dct_dc_size_luma",
    "videoDctSizeChrominance": "vlc_dct_dc_size_chroma",
    "videoDctCoeff": "vlc_dct_coeff",
}


def entries(text, var):
    m = re.search(r"var %s = \[\]vlc(?:Uint)?\{(.*?)\n\}" % var, text, re.S)
    body = re.sub(r"//[^\n]*", "", m.group(1))
    out = []
    for a, b in re.findall(r"\{\s*([^,{}]+?)\s*,\s*([^,{}]+?)\s*\}", body):
        out.append((int(eval(a, {"__builtins__": {}})), int(eval(b, {"__builtins__": {}}))))
    return out


def walk(table):
    """every root-to-leaf path: (bits, value, dead)"""
    out, todo = [], [(0, "")]
    while todo:
        index, bits = todo.pop()
        for bit in (0, 1):
            nxt, value = table[index + bit]
            if nxt > 0:
                todo.append((nxt, bits + str(bit)))
            else:
                out.append([bits + str(bit), value, 1 if nxt < 0 else 0])
    return sorted(out, key=lambda e: (len(e[0]), e[0]))


text = SRC.read_text()
result = {"source": "gen2brain/mpeg video.go:1088-1419 walked as buffer.go:352-376 does; [bits consumed, value returned, ended in a dead end]"}
for var, name in TABLES.items():
    result[name] = walk(entries(text, var))
    print("%-32s %4d tree entries -> %3d codes, %2d dead ends" % (var, len(entries(text, var)), sum(1 for e in result[name] if not e[2]), sum(e[2] for e in result[name])))
out = Path(__file__).resolve().parent / "vlc_known_answers.json"
out.write_text(json.dumps(result, indent=0, separators=(",", ":")) + "\n")
print("wrote", out, out.stat().st_size, "bytes")
