"""Every `file:line` citation of the boundary documents resolves against the reference tree.

`include/quest_hip.h` and `INTEGRATION.md` tell a maintainer which reference declaration each C-ABI entry replaces.  A
citation that points at the wrong lines is worse than none, and nobody re-reads them by hand: this test does, whenever
`/root/reference` is present (the build container; it is skipped on the GPU box, where the reference does not exist).

Checked for every citation `path/file.ext:A` or `:A-B` (and the `:C-D` continuations that follow it):
  * the file exists in the reference tree (by its path suffix; a bare basename must be unique or one candidate must fit),
  * the line range lies inside the file,
and, where the text names the symbol the citation is about -- `symbol (file:A-B`, `-> Symbol, file:A-B` --
  * that symbol occurs inside the cited lines.
"""
import os
import re

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["include/quest_hip.h", "INTEGRATION.md", "examples/pybind_binding.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")

CITE = re.compile(r"(?P<path>(?:[\w.-]+/)*[\w-]+\.(?:cuh|cu|h|py|sh|md|cmake|txt)):(?P<a>\d+)(?:-(?P<b>\d+))?")
CONT = re.compile(r"\s*(?:,|/|and|kernel|launch|math|planner|aux|dispatch)?\s*:(\d+)(?:-(\d+))?")
# the symbol a citation is about: `symbol (file:..`  |  `-> Symbol, file:..`  |  `Symbol, file:..` after an arrow
SYMBOL_BEFORE = re.compile(r"(?:->\s*)?([A-Za-z_][\w:]*)\s*(?:\(|,)\s*$")
DEFINITION_AFTER = re.compile(r"(?://[^\n]*\n)*(?:torch::Tensor|void|class|static \w+)\s+(\w+)")
# per document: at least this many citations / symbol-pinned citations must be found (the parser still works)
EXPECT = {"include/quest_hip.h": (20, 8), "INTEGRATION.md": (12, 0), "examples/pybind_binding.cpp": (8, 7)}
OWN_FILES = {"quest_hip.h", "INTEGRATION.md", "DESIGN.md", "SURVEY.md", "BASELINE.md", "VERDICT.md", "bench.py"}


def _index():
    files = {}
    for d, _, names in os.walk(REF):
        if "/.git" in d:
            continue
        for n in names:
            files.setdefault(n, []).append(os.path.join(d, n))
    return files


def _candidates(index, path):
    base = os.path.basename(path)
    return [f for f in index.get(base, []) if f.endswith("/" + path) or "/" not in path]


def _lines(path, cache={}):
    if path not in cache:
        with open(path, errors="replace") as f:
            cache[path] = f.read().splitlines()
    return cache[path]


def _citations(text):
    """(path, first line, last line, symbol or None, offset) for every citation of `text`, continuations included."""
    for m in CITE.finditer(text):
        path = m.group("path")
        if os.path.basename(path) in OWN_FILES or os.path.exists(os.path.join(ROOT, path)):
            continue  # a file of this repo, not of the reference
        before = text[max(0, m.start() - 80):m.start()].replace("\n * ", " ").replace("\n", " ")
        sym = SYMBOL_BEFORE.search(before)
        symbol = sym.group(1) if sym else None
        if symbol is None and before.rstrip().endswith("//"):
            # examples/pybind_binding.cpp: `// bsk_ops.h:A-B` on the line above the definition it mirrors
            nxt = DEFINITION_AFTER.match(text, text.find("\n", m.end()) + 1)
            symbol = nxt.group(1) if nxt else None
        a, b = int(m.group("a")), int(m.group("b") or m.group("a"))
        yield path, a, b, symbol, m.start()
        pos = m.end()
        while True:  # `, kernel :398-449` / `, :136` style continuations refer to the same file
            c = CONT.match(text, pos)
            if not c or text[pos:c.end()].count("\n") > 1:
                break
            yield path, int(c.group(1)), int(c.group(2) or c.group(1)), None, c.start()
            pos = c.end()


@pytest.mark.parametrize("doc", DOCS)
def test_every_citation_resolves_and_names_its_symbol(doc):
    index = _index()
    text = open(os.path.join(ROOT, doc)).read()
    problems, n_checked, n_symbols = [], 0, 0
    for path, a, b, symbol, off in _citations(text):
        line_no = text.count("\n", 0, off) + 1
        cands = _candidates(index, path)
        if not cands:
            problems.append(f"{doc}:{line_no}: {path}:{a}-{b}: no such file in the reference")
            continue
        fits = [c for c in cands if 1 <= a <= b <= len(_lines(c))]
        if not fits:
            problems.append(f"{doc}:{line_no}: {path}:{a}-{b}: outside the file ({[len(_lines(c)) for c in cands]} lines)")
            continue
        n_checked += 1
        # a code identifier (snake_case or CamelCase), not the English word that happens to precede a parenthesis
        if symbol and ("_" in symbol or re.search(r"[a-z][A-Z]", symbol)) and not symbol.isupper():
            plain = symbol.split("::")[-1]
            hit = any(plain in "\n".join(_lines(c)[a - 1:b]) for c in fits)
            # only symbols the reference file knows at all are pinned to the range (the sentence may name one of OUR
            # functions in front of a citation that explains it)
            known = any(plain in "\n".join(_lines(c)) for c in fits)
            if known:
                n_symbols += 1
                if not hit:
                    where = [i + 1 for c in fits for i, l in enumerate(_lines(c)) if plain in l][:6]
                    problems.append(f"{doc}:{line_no}: `{plain}` is not inside {path}:{a}-{b} (it occurs at lines {where})")
    assert not problems, "\n".join(problems)
    assert n_checked >= EXPECT[doc][0] and n_symbols >= EXPECT[doc][1], (n_checked, n_symbols)
