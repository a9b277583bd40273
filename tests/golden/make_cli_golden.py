"""Fixture: flags and defaults of the reference's `ccsmeth call_mods`, `call_freqb`, `trainm`, `train` and `extract` sub-parsers (ccsmeth/ccsmeth.py), captured by
running the reference's own main() argument parser in THIS container (reference importable here only).
Writes tests/golden/cli_golden.json.  usage: python tests/golden/make_cli_golden.py"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _ref_import import import_reference  # noqa: E402

import_reference()
captured = {}
orig = argparse.ArgumentParser.parse_args


def grab(self, args=None, namespace=None):      # the reference builds its parser inside main(): intercept parse_args
    captured["parser"] = self
    raise SystemExit(0)


argparse.ArgumentParser.parse_args = grab
try:
    from ccsmeth import ccsmeth as ref_cli
    sys.argv = ["ccsmeth", "call_mods"]
    try:
        ref_cli.main()
    except SystemExit:
        pass
finally:
    argparse.ArgumentParser.parse_args = orig
top = captured["parser"]
choices = next(a for a in top._actions if isinstance(a, argparse._SubParsersAction)).choices


def flags_of(sub):
    flags = {}
    for a in sub._actions:
        if a.dest == "help":
            continue
        flags[a.dest] = dict(options=list(a.option_strings), default=a.default, required=bool(a.required),
                             kind=type(a).__name__, type=getattr(a.type, "__name__", None))
    return flags


out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_golden.json")
doc = dict(source="ccsmeth/ccsmeth.py sub-parsers", flags=flags_of(choices["call_mods"]), call_freqb=flags_of(choices["call_freqb"]),
           trainm=flags_of(choices["trainm"]), extract=flags_of(choices["extract"]),
           train=flags_of(choices["train"]))
json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, {k: len(v) for k, v in doc.items() if k != "source"})
