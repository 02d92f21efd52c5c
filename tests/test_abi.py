"""The C-ABI library builds, loads, and exports every symbol include/mpeghip.h
declares; without a GPU it refuses to work instead of falling back."""
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    txt = (ROOT / "include" / "mpeghip.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpeghip_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from mpeg_amd import _build, abi
    _build.build_libmpeghip()
    lib = abi.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libmpeghip.so does not export %s" % n
        assert n in abi.SYMBOLS, "mpeg_amd.abi does not bind %s" % n
    assert lib.mpeghip_abi_version() == abi.ABI_VERSION == 3


def test_host_library_exports_exactly_what_its_header_declares():
    """include/mpeghost.h (the flat C API of libmpeghost) vs the library's dynamic symbol table vs the ctypes
    bindings the tests use: the three lists are the same."""
    import subprocess
    import sys
    sys.path.insert(0, str(ROOT / "tests"))
    import hostlib
    from mpeg_amd import _build
    txt = (ROOT / "include" / "mpeghost.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mpeghost_[a-z0-9_]+)\s*\(", txt)))
    L = hostlib.host()
    nm = subprocess.run(["nm", "-D", "--defined-only", str(_build.LIBMPEGHOST)], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\b(mpeghost_[a-z0-9_]+)\b", nm)))
    assert len(declared) >= 70
    assert declared == exported, (sorted(set(declared) - set(exported)), sorted(set(exported) - set(declared)))
    assert L.bound_names == declared, (sorted(set(declared) - set(L.bound_names)), sorted(set(L.bound_names) - set(declared)))


def test_descriptor_layouts_match_header():
    from mpeg_amd import desc
    assert desc.PIC_DTYPE.itemsize == 16 and desc.MB_DTYPE.itemsize == 32
    assert desc.MB_DTYPE.fields["coef_off"][1] == 16 and desc.MB_DTYPE.fields["mv_x"][1] == 8
    assert desc.PIC_DTYPE.fields["mb_first"][1] == 8


def test_no_cpu_fallback():
    """On a machine without a gfx950 device the product must fail loudly."""
    import torch
    from mpeg_amd import abi
    if torch.cuda.is_available():
        pytest.skip("GPU present: the loud-failure path is exercised on the CPU-only builder")
    with pytest.raises(abi.MpegHipError) as ei:
        abi.Context(0)
    assert ei.value.code == abi.ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value) or "no HIP device" in str(ei.value)


def test_product_does_not_reference_the_oracle():
    """Nothing under mpeg_amd/ or include/ may include, import or link oracle/."""
    for p in list((ROOT / "mpeg_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.is_file() and p.suffix in (".py", ".h", ".hpp", ".hip", ".cpp", ".c"):
            txt = p.read_text(errors="replace")
            assert "pyoracle" not in txt and "liboracle" not in txt and "mpeg_oracle" not in txt and "oracle_desc" not in txt, p


def test_no_kernel_uses_scratch():
    """Every gfx950 kernel must keep its state in registers: a run-time indexed local array silently
    moves to scratch memory (measured: 4x slower).  Checked on the compiler's resource-usage remarks."""
    import subprocess
    from mpeg_amd import _build
    flags = [f for f in _build.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]  # the product's own code generation flags
    cmd = [_build.hipcc_path(), *flags, "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-I", str(_build.INCLUDE), "-I", str(_build.CSRC),
           str(_build.CSRC / "mpeghip.hip"), "-o", "/dev/null"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stdout)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stdout)]
    assert len(names) == len(scratch) and len(names) >= 8
    bad = [(n, s) for n, s in zip(names, scratch) if s != 0]
    assert not bad, bad


def test_go_descriptor_structs_have_the_header_layout():
    """go/mpeghip hands []PicDesc / []MbDesc to C as they are: the Go structs (laid out by Go's rules: natural alignment,
    declaration order) must have the fields of mpeghip_pic_desc / mpeghip_mb_desc at the same offsets.  (The Go package
    cannot be compiled in this image; this reads its source.)"""
    from mpeg_amd import desc
    src = (ROOT / "go" / "mpeghip" / "mpeghip.go").read_text()
    size = {"uint8": 1, "int8": 1, "uint16": 2, "int16": 2, "uint32": 4, "int32": 4, "uint64": 8, "int64": 8}

    def go_layout(name):
        body = re.search(r"type %s struct \{(.*?)\n\}" % name, src, re.S).group(1)
        off, out, align_max = 0, {}, 1
        for line in body.splitlines():
            line = line.split("//")[0].strip()
            if not line:
                continue
            names, typ = line.rsplit(None, 1)
            count = 1
            m = re.match(r"\[(\d+)\](\w+)", typ)
            if m:
                count, typ = int(m.group(1)), m.group(2)
            for n in [x.strip() for x in names.split(",")]:
                a = size[typ]
                align_max = max(align_max, a)
                off = (off + a - 1) // a * a
                out[n] = off
                off += a * count
        return out, (off + align_max - 1) // align_max * align_max

    snake = lambda n: re.sub(r"(?<!^)(?=[A-Z])", "_", n).lower()
    for go_name, dtype in (("PicDesc", desc.PIC_DTYPE), ("MbDesc", desc.MB_DTYPE)):
        fields, total = go_layout(go_name)
        assert total == dtype.itemsize, go_name
        named = {snake(k): v for k, v in fields.items() if k != "_"}
        assert named, go_name
        for k, off in named.items():
            assert k in dtype.fields and dtype.fields[k][1] == off, (go_name, k, off)
        assert set(named) == {k for k in dtype.fields if not k.startswith("reserved")}, go_name
