"""The documents cite evidence by file: every profile they name must be in profiles/."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "docs/history/rounds_1_to_4.md"]


def test_every_cited_profile_exists():
    have = {p.name for p in (ROOT / "profiles").iterdir()}
    tags = {re.match(r"(r\d+[a-z0-9]*?)_", n).group(1) for n in have if re.match(r"r\d+[a-z0-9]*?_", n)}
    missing = []
    for doc in DOCS:
        text = (ROOT / doc).read_text()
        for m in re.finditer(r"`(?:profiles/)?((?:r\d\w*?|round[456]_[a-z]+)_[\w.\-]+\.(?:txt|json|csv|log))`", text):  # full file names
            name = m.group(1)
            if "*" in name or "…" in name:
                continue
            if name not in have:
                missing.append((doc, name))
        for m in re.finditer(r"`(r\d[0-9a-z]{1,3})`", text):  # bare run tags: some file of that run is kept
            if m.group(1) not in tags:
                missing.append((doc, m.group(1) + "_*"))
    assert not missing, missing


def test_every_entry_point_the_documents_and_the_go_binding_name_is_declared():
    """mpeghip_* / mpeghost_* names in the documents, the Go sources and the patch notes exist in include/*.h (names that
    end in an underscore are families; one is a stated future entry point)."""
    declared = set()
    for h in ("mpeghip.h", "mpeghost.h"):
        declared |= set(re.findall(r"\b(mpegh(?:ip|ost)_[a-z0-9_]+)\b", (ROOT / "include" / h).read_text()))
    future = {"mpeghip_video_stage_put_sparse"}
    named = {}
    files = [ROOT / d for d in DOCS] + sorted((ROOT / "go").rglob("*.go")) + sorted((ROOT / "go").rglob("*.md"))
    for f in files:
        for sym in re.findall(r"\b(mpegh(?:ip|ost)_[a-z0-9_]+)\b", f.read_text()):
            named.setdefault(sym, f.name)
    unknown = {s: f for s, f in named.items()
               if s not in declared and s not in future and not s.endswith("_") and not any(d.startswith(s + "_") for d in declared)}
    assert not unknown, unknown
