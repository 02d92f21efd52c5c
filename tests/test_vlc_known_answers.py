"""The ISO 11172-2 code lists of the product's parser and of the oracle against the REFERENCE's own code trees.

Checker and product decode with ONE transcription of Annex B (two copies of iso11172_vlc_codes.h: DESIGN.md section 4), so a code
that is wrong in the list and absent from the golden streams would pass every stream test.  tests/golden/vlc_known_answers.json
holds every root-to-leaf path of the reference's nine trees (video.go:1088-1419 walked as buffer.go:352-376 walks them; made by
tests/golden/make_vlc_known_answers.py): both lists must say exactly that — same codes, same values, same dead ends — and the
parser's lookup tables, built from the product's list, must answer every code followed by any bits the way the tree walk does."""
import ctypes as C
import json
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
KNOWN = json.loads((ROOT / "tests" / "golden" / "vlc_known_answers.json").read_text())
TABLES = [k for k in KNOWN if k != "source"]


def code_lists(header, prefix):
    text = Path(header).read_text()
    out = {}
    for m in re.finditer(r"static const %s_vlc_code %s_(vlc_\w+)\[\] = \{(.*?)\n\};" % (prefix, prefix), text, re.S):
        rows = re.findall(r'\{"([01]+)",\s*(-?(?:0x[0-9a-fA-F]+|\d+)),\s*([01])\}', m.group(2))
        out[m.group(1)] = sorted(([b, int(v, 0), int(d)] for b, v, d in rows), key=lambda e: (len(e[0]), e[0]))
    return out


@pytest.mark.parametrize("header,prefix", [("mpeg_amd/host/iso11172_vlc_codes.h", "mpg"), ("oracle/iso11172_vlc_codes.h", "orc")])
def test_code_lists_are_the_reference_trees(header, prefix):
    lists = code_lists(ROOT / header, prefix)
    assert sorted(lists) == sorted(TABLES)
    for name in TABLES:
        want = [[b, v & 0xffff if name == "vlc_dct_coeff" else v, d] for b, v, d in KNOWN[name]]
        got = [[b, v & 0xffff if name == "vlc_dct_coeff" else v, d] for b, v, d in lists[name]]
        # a dead end yields 0 whatever the list says its value is (VlcTable / vlc_build store 0 for it)
        want = [[b, 0 if d else v, d] for b, v, d in want]
        got = [[b, 0 if d else v, d] for b, v, d in got]
        assert got == want, name


def test_reference_trees_are_complete_prefix_codes():
    """(the fixture itself: every bit string is answered by exactly one path — Kraft sum 1 — so 'same set of paths' above means
    'same answer for every input')"""
    from fractions import Fraction
    for name in TABLES:
        assert sum(Fraction(1, 2 ** len(b)) for b, _, _ in KNOWN[name]) == 1, name
        bits = [b for b, _, _ in KNOWN[name]]
        assert len(set(bits)) == len(bits)


def test_parser_lookup_tables_answer_every_known_code():
    """mpeg_amd/host/vlc.hpp's two-level tables through the library: every path of the reference's trees, followed by zeros and
    by ones, decodes to the reference's value and consumes the reference's number of bits."""
    import hostlib
    H = hostlib.host()
    if not hasattr(H, "mpeghost_debug_vlc_decode"):
        pytest.skip("library without mpeghost_debug_vlc_decode")
    H.mpeghost_debug_vlc_decode.restype = C.c_int
    H.mpeghost_debug_vlc_decode.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for t, name in enumerate(TABLES):
        for bits, value, dead in KNOWN[name]:
            for fill in (0, 1):
                window = int(bits + str(fill) * (64 - len(bits)), 2)
                v, n = C.c_int(), C.c_int()
                assert H.mpeghost_debug_vlc_decode(t, window, C.byref(v), C.byref(n)) == 0
                want = 0 if dead else (value & 0xffff if name == "vlc_dct_coeff" else value)
                assert (v.value & 0xffff if name == "vlc_dct_coeff" else v.value, n.value) == (want, len(bits)), (name, bits, fill)
