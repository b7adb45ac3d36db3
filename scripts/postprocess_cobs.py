#!/usr/bin/env python3
"""Drop-in for the reference's scripts/postprocess_cobs.py (`-n int`, stdin -> stdout)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phylign_amd.postprocess import main  # noqa: E402

if __name__ == "__main__":
    main()
