"""bench.py's stdout contract: ONE compact JSON line that fits the driver's record (it keeps a tail of ~8 000 characters: round 5's
14.5 kB line lost its dense / fused / mixed / audio legs there), the prose in a sidecar; and `python bench.py --gpus N` starting its N
ranks by itself (the command the driver's contract names, re-executed under torch.distributed.run)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
CANNED = ROOT / "tests" / "golden" / "bench_full_canned.json"   # a full result of a default run (GPU box), as the sidecar holds it


def test_compact_line_fits_the_drivers_record_and_keeps_every_leg():
    import bench
    full = json.loads(CANNED.read_text())
    line = bench.compact_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < bench.COMPACT_LIMIT == 7500, len(text)
    assert text.startswith('{"metric"')
    # the contract's fields, unchanged
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["scaling"] == "weak"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-4          # (5 significant digits)
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r) and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] == "port"
    # every leg: numbers + a boolean parity, the second half of the metric and the worst-case legs FIRST among the extras
    keys = list(line)
    first_extra = keys.index("audio")
    assert keys[first_extra:first_extra + 3] == ["audio", "dense", "dense_rgba_fused"]
    for k in ("audio", "dense", "dense_rgba_fused", "rgba_fused", "mixed", "sif", "audio_large", "audio_fma_window"):
        leg = line[k]
        assert leg["parity_ok"] is True and 0 < leg["frac"] < 1 and leg["ms"] > 0 and leg["alg_bytes"] > 0 and leg["value"] > 0, k
        assert "metric" not in leg and "parity" not in leg and "roofline" not in leg, k          # the prose stays in the sidecar
    assert line["sif_single"]["us_per_picture"] > 0 and line["single_stream"]["typical"]["us_per_picture"] > 0
    assert line["reference_benchmarks"]["decode_video_test_mpg"] > 0 and line["host_parsed"]["value"] > 0
    assert line["sidecar"] == "bench_legs.json" and len(line["csrc_sha256"]) == 16
    # nothing long anywhere: the line is numbers, booleans and short tags
    def walk(v):
        if isinstance(v, dict):
            for x in v.values():
                walk(x)
        elif isinstance(v, list):
            for x in v:
                walk(x)
        elif isinstance(v, str):
            assert len(v) <= 200, v
    walk(line)


def test_compact_line_of_a_multi_rank_result():
    """N > 1: only the primary leg, audio and host_fed exist; the line still has the contract's fields and stays short."""
    import bench
    full = json.loads(CANNED.read_text())
    for k in ("dense", "rgba_fused", "dense_rgba_fused", "mixed", "sif", "audio_large", "audio_fma_window", "single_stream", "reference_benchmarks",
              "host_parsed", "audio_host_parsed"):
        full[k] = None
    full.update(n_gpus=8, ranks=8, per_rank_value=[full["value"] / 8] * 8)
    full["config"]["devices"] = ["0000:%02x:00.0" % (5 + 16 * i) for i in range(8)]
    line = bench.compact_line(full)
    assert line["n_gpus"] == 8 and len(line["per_rank_value"]) == 8 and "dense" not in line and line["audio"]["parity_ok"] is True
    assert len(json.dumps(line, separators=(",", ":"))) < 4000


def test_bench_gpus_2_starts_two_ranks_by_itself():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run with two
    ranks on a free port; they rendezvous over gloo; rank 0 prints the one line, rank 1 nothing; the exit code is the launcher's."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-launch"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j == {"dry_launch": True, "gpus_asked": 2, "ranks": 2, "ranks_counted": 2, "local_ranks": [0, 1], "distinct_pids": 2}


def test_a_failing_rank_fails_the_launch():
    """The parent's exit code is the children's: without a GPU every rank exits non-zero ("needs a MI355X"), and so does bench.py."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs a MI355X" in r.stderr
