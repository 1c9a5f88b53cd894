"""Dump the reference's `bean run` flag set (bean/model/parser.py, imported by
path) to run_flags.json: option strings, dest, default, type, choices, nargs, action."""
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def dump(parser):
    out = []
    for a in parser._actions:
        if a.dest == "help":
            continue
        out.append({
            "flags": sorted(a.option_strings), "dest": a.dest, "default": a.default,
            "type": getattr(a.type, "__name__", None), "choices": list(a.choices) if a.choices else None,
            "nargs": a.nargs, "action": type(a).__name__, "required": a.required,
        })
    return out


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("ref_parser", "/root/reference/bean/model/parser.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    json.dump(dump(ref.parse_args()), open(os.path.join(HERE, "run_flags.json"), "w"), indent=1)
    print("wrote run_flags.json")
