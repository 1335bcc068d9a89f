"""prints a bench JSON line readably: python tools/show.py <file> ..."""
import json
import sys


def short(v, depth=0):
    if isinstance(v, float):
        return f"{v:.5g}"
    if isinstance(v, str):
        return v if len(v) < 70 else v[:67] + "..."
    return v


def dump(d, ind=0):
    for k, v in d.items():
        if isinstance(v, dict):
            print(" " * ind + f"{k}:")
            dump(v, ind + 2)
        else:
            print(" " * ind + f"{k}: {short(v)}")


for f in sys.argv[1:]:
    for line in open(f):
        line = line.strip()
        if line.startswith("{"):
            dump(json.loads(line))
