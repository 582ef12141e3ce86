"""CPU restatement of the reference's STL -> .bin script.  TEST INFRASTRUCTURE ONLY: nothing in
the product imports this file.

Follows prepare_data/data_formatter.js:9-18 statement by statement.  Pinned: the committed
fixtures tests/golden/*_from_node.bin are outputs of that very script run under node (see
tests/golden/make_formatter_golden.sh), and small_knight.bin of the reference is reproduced
byte for byte.
"""
import re

import numpy as np

_WS = " \t\n\r\v\f ﻿  "
_DEC = re.compile(r"[+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?\Z")
_INT = {"x": (16, re.compile(r"[0-9a-fA-F]+\Z")), "b": (2, re.compile(r"[01]+\Z")), "o": (8, re.compile(r"[0-7]+\Z"))}


def js_number(token):
    """ECMAScript Number(string) (data_formatter.js:13 `.map(Number)`)."""
    t = token.strip(_WS)
    if t == "":
        return 0.0
    if t in ("Infinity", "+Infinity"):
        return float("inf")
    if t == "-Infinity":
        return float("-inf")
    if len(t) > 2 and t[0] == "0" and t[1].lower() in _INT:
        base, pat = _INT[t[1].lower()]
        return float(int(t[2:], base)) if pat.match(t[2:]) else float("nan")
    if _DEC.match(t):
        return float(t)
    return float("nan")


def stl_format(text, scale=1.0):
    """data_formatter.js:13: split("\\n") -> filter(includes("vertex")) -> split("vertex ")[1]
    -> split(" ") -> Number -> * scale; :18 writeFloatLE (one rounding to f32)."""
    if isinstance(text, bytes):
        text = text.decode("utf-8")
    values = []
    for line in text.split("\n"):
        if "vertex" not in line:
            continue
        parts = line.split("vertex ")
        if len(parts) < 2:
            raise ValueError("line contains 'vertex' but not 'vertex ' (the script throws a TypeError)")
        for tok in parts[1].split(" "):
            values.append(js_number(tok) * float(scale))
    with np.errstate(over="ignore"):
        return np.asarray(values, dtype=np.float64).astype(np.float32)
