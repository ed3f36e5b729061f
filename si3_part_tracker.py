#!/usr/bin/env python3
"""`si3_part_tracker.py` -- same command line as the reference script of that name
(flags -i -m -s -k -e -F -N -p), per-buoy advection on MI355X through libsitrk.
See sitrack_amd/driver.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from sitrack_amd.driver import main  # noqa: E402

if __name__ == '__main__':
    main()
