#!/usr/bin/env python3
"""1-D entry point (reference: main_1d.py): `python main_1d.py model=ffno_1d/ffno_1d dataset=synthetic/ks_512 ...`"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rpde.entry import run  # noqa: E402

if __name__ == "__main__":
    run(1)
