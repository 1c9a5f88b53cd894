"""``bean`` command dispatcher (``bean/cli/execute.py:30-84``): only the ``run``
sub-command is in scope of this implementation."""
from __future__ import annotations

import argparse
import sys

from ..model.parser import parse_args as attach_run_args


def get_parser():
    parser = argparse.ArgumentParser(prog="bean")
    sub = parser.add_subparsers(dest="subcommand", help="bean subcommands")
    run = sub.add_parser("run", help="Quantify variant effect sizes from screen data (MI355X)")
    attach_run_args(run)
    return parser


def main(argv=None):
    parser = get_parser()
    args = parser.parse_args(argv)
    if args.subcommand != "run":
        parser.print_help()
        return 2
    from .run import main as run_main

    run_main(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
