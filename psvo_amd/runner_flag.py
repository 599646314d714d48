"""Command-line entry: `python -m psvo_amd.runner_flag --flag=value ...` with the reference's flag
names and defaults (reference src/runner_flag.py:166-287; the registry lives in psvo_amd/flags.py)."""
import sys

from .flags import parse_flags
from .runner import main

if __name__ == "__main__":
    main(parse_flags(sys.argv[1:]))
