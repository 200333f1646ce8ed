import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "map-code_amd"), os.path.join(ROOT, "tests", "golden"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_terminal_summary(terminalreporter):
    """What the golden gradient checks compared (tests/util.py: note_golden): the reference's fixture directly,
    or the oracle on the step's own ReLU pattern with N flipped units; also written to gpurun_out/ when that
    directory exists (the GPU box)."""
    try:
        from util import GOLDEN_LOG
    except Exception:
        return
    if not GOLDEN_LOG:
        return
    direct = [t for t, how, _ in GOLDEN_LOG if how == "direct"]
    fall = [(t, n) for t, how, n in GOLDEN_LOG if how != "direct"]
    tr = terminalreporter
    tr.write_sep("-", "golden gradient checks")
    tr.write_line(f"{len(direct)} compared with the reference's fixture directly; {len(fall)} through the oracle on "
                  f"the step's own ReLU pattern" + (": " + ", ".join(f"{t} ({n} flips)" for t, n in fall) if fall else ""))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        import json
        with open(os.path.join(out, "golden_modes.json"), "w") as f:
            json.dump({"direct": direct, "pattern_fallback": [{"case": t, "flips": n} for t, n in fall]}, f, indent=1)
