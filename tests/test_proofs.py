"""Machine-checked facts the kernels lean on (CPU only, exhaustive)."""
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent


def test_short_division_equals_ieee_division_for_every_float32(tmp_path):
    """audio_lane.h scale_short(): x / -1090519040 as x*y + two fma corrections is the correctly
    rounded quotient for all 2^32 bit patterns that scale_short_ok() lets through."""
    exe = tmp_path / "div_const"
    flags = ["-mfma"] if "fma" in Path("/proc/cpuinfo").read_text().split() else []
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", *flags, "-pthread", str(HERE / "proofs" / "div_const.c"), "-o", str(exe), "-lm"], check=True)
    r = subprocess.run([str(exe), str(min(64, os.cpu_count() or 8))], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    words = r.stdout.split()
    assert int(words[1]) > 3_700_000_000 and int(words[5]) == 0, r.stdout
