"""Registers, LDS and scratch of every kernel of the sources in the tree -> stdout (profiles/<tag>_kernel_resources.txt).
hipcc cross-compiles: no GPU needed.   usage: python tools/kernel_resources.py"""
import re, subprocess, sys, tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def main():
    with tempfile.TemporaryDirectory() as tmp:
        asm = Path(tmp) / "mpeghip.s"
        from mpeg_amd import _build
        flags = [f for f in _build.HIPCC_FLAGS if f != "-shared"]
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-I", str(ROOT / "include"), "-I", str(ROOT / "mpeg_amd/csrc"), "--cuda-device-only", "-S",
                        str(ROOT / "mpeg_amd/csrc/mpeghip.hip"), "-o", str(asm)], check=True, stderr=subprocess.DEVNULL)
        text = asm.read_text()
    print("# kernel resources of the shipped sources (hipcc --offload-arch=gfx950 -O3 ... -S, .amdgpu_metadata); csrc sha256 %s" % bench.sources_sha256())
    print("# waves per SIMD = min(512 / VGPRs rounded up to 8, LDS: workgroups per CU (160000 / bytes) x waves per workgroup (from "
          ".max_flat_workgroup_size) / 4 SIMDs, 8)")
    print("%-74s %6s %6s %8s %8s %s" % ("kernel", "VGPR", "SGPR", "LDS B", "scratch", "waves/SIMD"))
    for block in text.split("  - .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\.%s:\s+(\S+)" % name, block)
            return m.group(1) if m else "?"
        sym = field("name")
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() or sym
        vgpr, sgpr, lds, scratch = int(field("vgpr_count")), int(field("sgpr_count")), int(field("group_segment_fixed_size")), int(field("private_segment_fixed_size"))
        by_regs = 512 // ((vgpr + 7) // 8 * 8) if vgpr else 8
        wg = field("max_flat_workgroup_size")
        waves_per_wg = max(1, int(wg) // 64) if wg != "?" else 1
        by_lds = 8 if lds == 0 else 160000 // lds * waves_per_wg // 4
        print("%-74s %6d %6d %8d %8d %d" % (name[:70], vgpr, sgpr, lds, scratch, min(8, by_regs, by_lds)))


if __name__ == "__main__":
    main()
