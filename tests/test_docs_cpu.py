"""Every file the documents, the bench line and the headers cite must exist in the tree (cited evidence had been
deleted by a collection script).  A citation is a repo-relative path under profiles/, tools/, tests/, oracle/, lwsnet_amd/,
include/ with a file extension; glob characters and `<...>` / `{...}` placeholders are expanded or skipped."""
import glob
import os
import re

from conftest import ROOT

DOCS = ["DESIGN.md", "INTEGRATION.md", "README.md", "profiles/NOTES.md", "profiles/r04/README.md", "profiles/r05/README.md", "profiles/r06/README.md", "tools/README.md", "bench.py",
        "include/lwsnet_hip.h", "lwsnet_amd/dist.py", "lwsnet_amd/launch.py", "lwsnet_amd/inference.py", "__graft_entry__.py", "tests/golden/kitti_pair/README.md"]
PAT = re.compile(r"(?<![\w/.])((?:profiles|tools|tests|oracle|lwsnet_amd|include)/[\w./*<>{},\-]+\.(?:py|sh|hip|md|txt|json|csv|npz|png|h|c)(?![\w]))")


def _expand(path):
    """`a_{x,y}.txt` -> both; `*` globs; placeholders in <> are skipped."""
    m = re.search(r"\{([^{}]*,[^{}]*)\}", path)
    if m:
        out = []
        for alt in m.group(1).split(","):
            out += _expand(path[:m.start()] + alt + path[m.end():])
        return out
    return [path]


def test_cited_files_exist():
    missing = []
    for doc in DOCS:
        p = os.path.join(ROOT, doc)
        if not os.path.isfile(p):
            continue
        text = open(p, encoding="utf-8").read()
        for cite in sorted(set(PAT.findall(text))):
            if "<" in cite or ">" in cite or "…" in cite:
                continue
            for c in _expand(cite):
                if "{" in c or "}" in c:
                    continue
                hits = glob.glob(os.path.join(ROOT, c)) if "*" in c else ([c] if os.path.exists(os.path.join(ROOT, c)) else [])
                if not hits:
                    missing.append(f"{doc}: {c}")
    assert not missing, "cited but absent:\n  " + "\n  ".join(missing)
