"""The investigation helpers under tools/ (micro-benchmarks, sweeps, profile reducers: ~100 one-off scripts) are not part of the
product, but they are how the numbers in DESIGN.md / profiles/ were produced: they must at least still PARSE after a kernel or
API change ("none of it is exercised by a test, so it rots").  Python files are byte-compiled, shell scripts checked
with `bash -n`; scripts that name a library symbol must name one that include/efx.h still declares."""
import os
import py_compile
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = os.path.join(ROOT, "tools")


def _files(ext):
    out = []
    for d, _, fs in os.walk(TOOLS):
        if "__pycache__" in d:
            continue
        out += [os.path.join(d, f) for f in fs if f.endswith(ext)]
    return sorted(out)


@pytest.mark.parametrize("path", _files(".py"), ids=lambda p: os.path.relpath(p, TOOLS))
def test_python_tool_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", _files(".sh"), ids=lambda p: os.path.relpath(p, TOOLS))
def test_shell_tool_parses(path):
    r = subprocess.run(["bash", "-n", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_tools_name_only_exported_symbols():
    header = open(os.path.join(ROOT, "include", "efx.h")).read()
    declared = set(re.findall(r"\b(efx_[a-z0-9_]+)\s*\(", header))
    unknown = {}
    for path in _files(".py") + _files(".cpp"):
        for sym in set(re.findall(r"\blib\(\)\.(efx_[a-z0-9_]+)|\b(efx_[a-z0-9_]+)\s*\(", open(path, errors="replace").read())):
            name = sym[0] or sym[1]
            if name and name not in declared and not name.startswith(("efx_blob_", "efx_debug_")):      # (debug-build-only hooks, guarded by hasattr)
                unknown.setdefault(os.path.relpath(path, ROOT), set()).add(name)
    assert not unknown, f"tools call symbols include/efx.h does not declare: {unknown}"
