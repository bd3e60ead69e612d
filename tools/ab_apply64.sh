#!/bin/bash
# A/B of the Schwarz apply kernels with the bench's 64-node boxes: headline grid and the per-GPU share of the 8-GPU run
python tools/ab_apply.py 214 target=64,apply_kind=6 target=64,apply_kind=0
python tools/ab_apply.py 107 target=64,apply_kind=6 target=64,apply_kind=0
