"""The documents cite evidence by file: every profile they name must be in profiles/."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md"]


def test_every_cited_profile_exists():
    have = {p.name for p in (ROOT / "profiles").iterdir()}
    tags = {re.match(r"(r\d+[a-z0-9]*?)_", n).group(1) for n in have if re.match(r"r\d+[a-z0-9]*?_", n)}
    missing = []
    for doc in DOCS:
        text = (ROOT / doc).read_text()
        for m in re.finditer(r"`(?:profiles/)?(r\d\w*?_[\w.\-]+\.(?:txt|json|csv|log))`", text):  # full file names
            name = m.group(1)
            if "*" in name or "…" in name:
                continue
            if name not in have:
                missing.append((doc, name))
        for m in re.finditer(r"`(r\d[0-9a-z]{1,3})`", text):  # bare run tags: some file of that run is kept
            if m.group(1) not in tags:
                missing.append((doc, m.group(1) + "_*"))
    assert not missing, missing
