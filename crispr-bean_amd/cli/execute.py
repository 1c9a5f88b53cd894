"""``bean`` command dispatcher (``bean/cli/execute.py:30-84``): the ``run`` sub-command, its companion
``build-prior`` and the mask-producing part of ``qc`` are in scope of this implementation."""
from __future__ import annotations

import argparse
import sys

from ..model.parser import parse_args as attach_run_args


def get_parser():
    parser = argparse.ArgumentParser(prog="bean")
    sub = parser.add_subparsers(dest="subcommand", help="bean subcommands")
    run = sub.add_parser("run", help="Quantify variant effect sizes from screen data (MI355X)")
    attach_run_args(run)
    from .build_prior import attach_args as attach_prior_args

    attach_prior_args(sub.add_parser("build-prior", help="obtain prior_params.pkl for batched runs"))
    from .qc import attach_args as attach_qc_args

    attach_qc_args(sub.add_parser("qc", help="QC of the screen: mask low-quality samples and outlier guides"))
    return parser


def main(argv=None):
    parser = get_parser()
    args = parser.parse_args(argv)
    if args.subcommand == "build-prior":
        from .build_prior import main as build_prior_main

        build_prior_main(args)
        return 0
    if args.subcommand == "qc":
        from .qc import main as qc_main

        qc_main(args)
        return 0
    if args.subcommand != "run":
        parser.print_help()
        return 2
    from .run import main as run_main

    try:
        run_main(args)
    finally:
        # `torchrun ... bean run`: leave the process group cleanly (RCCL communicators, store)
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
