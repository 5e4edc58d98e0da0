"""Console entry point ``llm-consensus-bench``: runs the repository's ``bench.py`` (kept at the
repository root because the measurement driver calls it there) from an installed package."""
from __future__ import annotations

import runpy
import sys
from pathlib import Path


def main() -> int:
    here = Path(__file__).resolve().parent.parent / "bench.py"
    if not here.exists():
        print("llm-consensus-bench needs a source checkout (bench.py lives at the repository root)", file=sys.stderr)
        return 2
    sys.argv[0] = str(here)
    try:
        runpy.run_path(str(here), run_name="__main__")
    except SystemExit as e:
        return int(e.code or 0)
    return 0
