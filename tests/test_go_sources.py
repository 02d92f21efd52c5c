"""The Go side cannot be compiled here (no Go toolchain in the image), so it is checked at source level:
  * every identifier go/patch/*.go takes from the binding package exists in go/mpeghip/mpeghip.go, and every call has
    the number of arguments the binding declares;
  * every C function the binding calls is declared in include/mpeghip.h with that many parameters;
  * every hook go/patch/PATCH.md describes cites lines of the reference that still hold the code it talks about
    (only where /root/reference is mounted: the build container).
(tests/test_abi.py checks the binding's descriptor structs against the header's layout.)"""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BINDING = (ROOT / "go" / "mpeghip" / "mpeghip.go").read_text()
PATCH = {p.name: p.read_text() for p in sorted((ROOT / "go" / "patch").glob("*.go"))}
HEADER = (ROOT / "include" / "mpeghip.h").read_text()
REFERENCE = Path("/root/reference")


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _args(src, at):
    """Number of top-level arguments of the call whose '(' is at src[at]."""
    depth, n, empty, i = 0, 1, True, at
    while True:
        c = src[i]
        if c in "([{":
            depth += 1
        elif c in ")]}":
            depth -= 1
            if depth == 0:
                return 0 if empty else n
        elif c == "," and depth == 1:
            n += 1
        elif depth >= 1 and not c.isspace():
            empty = False
        i += 1


def _params(sig):
    sig = sig.strip()
    if not sig:
        return 0
    depth, n = 0, 1
    for c in sig:
        depth += c in "([{"
        depth -= c in ")]}"
        n += c == "," and depth == 0
    return n


def binding_api():
    src = _strip_comments(BINDING)
    funcs, methods = {}, {}
    for m in re.finditer(r"^func (\w+)\(", src, flags=re.M):
        close = src.index(")", m.end())  # (no nested parentheses in these signatures)
        funcs[m.group(1)] = _params(src[m.end():close])
    for m in re.finditer(r"^func \(\w+ \*?(\w+)\) (\w+)\(", src, flags=re.M):
        depth, i = 1, m.end()
        while depth:
            depth += src[i] == "("
            depth -= src[i] == ")"
            i += 1
        methods.setdefault(m.group(1), {})[m.group(2)] = _params(src[m.end():i - 1])
    names = set(funcs) | set(re.findall(r"^type (\w+) ", src, flags=re.M)) | set(re.findall(r"^\t(\w+)\s+= C\.", src, flags=re.M))
    return funcs, methods, names


def test_every_binding_identifier_the_patch_uses_exists():
    funcs, methods, names = binding_api()
    for name, text in PATCH.items():
        src = _strip_comments(text)
        for m in re.finditer(r"mpeghip\.(\w+)", src):
            assert m.group(1) in names, "%s uses mpeghip.%s, which go/mpeghip/mpeghip.go does not define" % (name, m.group(1))
        for m in re.finditer(r"mpeghip\.(\w+)\(", src):
            if m.group(1) in funcs:
                assert _args(src, m.end() - 1) == funcs[m.group(1)], "%s: mpeghip.%s called with the wrong number of arguments" % (name, m.group(1))


def test_method_calls_on_binding_objects_match_their_declarations():
    _, methods, _ = binding_api()
    # receivers: struct fields / variables declared with a binding type (a file that does not declare the field itself —
    # frame_hip.go uses hipVideo.dev — takes the other files' declarations), what hipContext() returns, what Open* return
    declared = {}
    for name, text in PATCH.items():
        src = _strip_comments(text)
        declared[name] = {m.group(1): {m.group(2)} for m in re.finditer(r"(\w+)\s+\*mpeghip\.(\w+)", src)}
        for m in re.finditer(r"(\w+), \w+ :?= hipContext\(\)", src):
            declared[name][m.group(1)] = {"Context"}
    everywhere = {}
    for d in declared.values():
        for var, typs in d.items():
            everywhere.setdefault(var, set()).update(typs)
    checked = 0
    for name, text in PATCH.items():
        src = _strip_comments(text)
        for var, all_types in everywhere.items():
            typs = declared[name].get(var, all_types)
            for m in re.finditer(r"(?<![\w.])(?:\w+\.)*%s\.(\w+)\(" % re.escape(var), src):
                meth, n = m.group(1), _args(src, m.end() - 1)
                have = [methods[t][meth] for t in typs if t in methods and meth in methods[t]]
                assert have, "%s: %s.%s: no binding type among %s has such a method" % (name, var, meth, sorted(typs))
                assert n in have, "%s: %s.%s called with %d arguments, declared with %s" % (name, var, meth, n, have)
                checked += 1
    assert checked >= 9    # OpenVideo, OpenAudio, SubmitSparse, SetQuant, ReadPlanes, RGBA, Synth x2, Close ...


def test_every_c_function_the_binding_calls_is_declared_with_that_arity():
    header = _strip_comments(HEADER)
    decl = {}
    for m in re.finditer(r"\b(mpeghip_\w+)\s*\(", header):
        depth, i = 1, m.end()
        while depth:
            depth += header[i] == "("
            depth -= header[i] == ")"
            i += 1
        sig = header[m.end():i - 1].strip()
        decl[m.group(1)] = 0 if sig in ("", "void") else _params(sig)
    src = _strip_comments(BINDING)
    calls = list(re.finditer(r"C\.(mpeghip_\w+)\(", src))
    assert len(calls) >= 25
    for m in calls:
        assert m.group(1) in decl, "the binding calls %s, which include/mpeghip.h does not declare" % m.group(1)
        assert _args(src, m.end() - 1) == decl[m.group(1)], "%s: %d arguments in the binding, %d parameters in the header" % (
            m.group(1), _args(src, m.end() - 1), decl[m.group(1)])


# (file, first line, last line, what those lines of the reference must still contain) — the hooks of go/patch/PATCH.md
HOOKS = [
    ("video.go", 10, 22, "type Frame struct"), ("video.go", 57, 106, "type Video struct"),
    ("video.go", 31, 36, "func (f *Frame) RGBA() *image.RGBA"),
    ("video.go", 324, 326, "v.initFrame(&v.frameBackward)"),
    ("video.go", 406, 409, "frameTemp := v.frameForward"), ("video.go", 421, 427, "v.decodeSlice("),
    ("video.go", 430, 433, "v.frameCurrent = frameTemp"),
    ("video.go", 503, 510, "v.predictMacroblock()"), ("video.go", 529, 531, "v.quantizerScale = v.buf.read(5)"),
    ("video.go", 556, 561, "v.decodeBlock(block)"),
    ("video.go", 627, 627, "copyMacroblock(fwH, fwV"), ("video.go", 629, 629, "copyMacroblock(bwH, bwV"),
    ("video.go", 632, 632, "copyMacroblock(bwH, bwV"), ("video.go", 635, 635, "copyMacroblock(fwH, fwV"),
    ("video.go", 263, 263, "frame.Time = v.time"), ("video.go", 209, 209, "func (v *Video) Decode() *Frame"),
    # round 6: the hooks inside the reference's own decodeBlock, and the look-ahead's
    ("video.go", 639, 639, "func (v *Video) decodeBlock(block int)"), ("video.go", 713, 713, "return // invalid"),
    ("video.go", 716, 717, "deZigZagged := int(videoZigZag[n]) & 63"), ("video.go", 747, 747, "// Move block to its place"),
    ("video.go", 719, 744, "videoPremultiplierMatrix[deZigZagged]"),
    ("video.go", 183, 183, "func (v *Video) Time() float64"), ("video.go", 189, 189, "func (v *Video) SetTime("),
    ("video.go", 195, 195, "func (v *Video) Rewind()"), ("video.go", 204, 204, "func (v *Video) HasEnded() bool"),
    ("audio.go", 53, 81, "type Audio struct"), ("audio.go", 83, 104, "func NewAudio(buf *Buffer) *Audio"),
    ("audio.go", 378, 422, "synthWindow("), ("audio.go", 426, 426, "a.buf.align()"),
    ("audio.go", 163, 163, "func (a *Audio) Decode() *Samples"), ("audio.go", 137, 137, "func (a *Audio) Time() float64"),
    ("audio.go", 143, 143, "func (a *Audio) SetTime("), ("audio.go", 149, 149, "func (a *Audio) Rewind()"),
    ("audio.go", 157, 157, "func (a *Audio) HasEnded() bool"),
    ("mpeg.go", 460, 522, "func (m *MPEG) SeekFrame("),
]


def test_patch_md_names_every_hook_line():
    doc = (ROOT / "go" / "patch" / "PATCH.md").read_text()
    for _, a, b, _ in HOOKS:
        cite = ":%d" % a if a == b else ":%d-%d" % (a, b)
        assert cite in doc, "PATCH.md no longer cites %s" % cite


@pytest.mark.skipif(not REFERENCE.exists(), reason="the reference is only mounted in the build container")
def test_the_cited_reference_lines_still_hold_that_code():
    for f, a, b, must in HOOKS:
        lines = (REFERENCE / f).read_text().splitlines()[a - 1:b]
        assert any(must in ln for ln in lines), "%s:%d-%d no longer holds `%s`" % (f, a, b, must)
