"""`python runner_flag.py --flag=value` -- same invocation as the reference's src/runner_flag.py."""
import sys

from psvo_amd.flags import parse_flags
from psvo_amd.runner import main

if __name__ == "__main__":
    main(parse_flags(sys.argv[1:]))
