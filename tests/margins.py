"""Measured parity margins of the GPU tests.  `within(name, value, tol)` is the comparison the tests assert on; it also keeps
(value, tol) so that a session run with QSP_MARGINS_OUT=<file> leaves a JSON table of every measured error next to the
tolerance it was held to (the committed copy: profiles/r02_test_margins.json; each tolerance is at most 4x the value measured
there)."""
import json
import os

_REC = {}


def within(name, value, tol):
    value = float(value)
    r = _REC.setdefault(name, dict(measured=0.0, tol=float(tol), n=0))
    r["measured"] = max(r["measured"], value)
    r["tol"] = float(tol)
    r["n"] += 1
    return value <= tol


def dump():
    path = os.environ.get("QSP_MARGINS_OUT")
    if path and _REC:
        for r in _REC.values():
            r["tol_over_measured"] = (r["tol"] / r["measured"]) if r["measured"] > 0 else None
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump(dict(sorted(_REC.items())), f, indent=1)
