#!/usr/bin/env python3
"""CLI of crispr-bean_amd/isa_check.py: python scripts/check_exec_join.py lib.so [...] [--all] [--strict]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bean_amd  # noqa: E402,F401
from bean_amd import isa_check  # noqa: E402

if __name__ == "__main__":
    sys.exit(isa_check.main(sys.argv[1:]))
