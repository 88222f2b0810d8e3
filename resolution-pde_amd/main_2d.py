#!/usr/bin/env python3
"""2-D entry point (reference: main_2d.py): `python main_2d.py model=ffno_2d/ffno_2d dataset=synthetic/ns_256 ...`
One process per GPU; launch N ranks with `python -m torch.distributed.run --nproc-per-node N main_2d.py ...`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rpde.entry import run  # noqa: E402

if __name__ == "__main__":
    run(2)
