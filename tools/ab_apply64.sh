#!/bin/bash
# A/B of the Schwarz apply kernels with the bench's 64-node boxes (6 = K-split k_apply_mfma, 7 = warp-specialised k_apply_ws)
python tools/ab_apply.py 214 target=64,apply_kind=6 target=64,apply_kind=7 target=64,apply_kind=7,apply_span=128 target=64,apply_kind=7,apply_span=320
python tools/ab_apply.py 107 target=64,apply_kind=6 target=64,apply_kind=7
python tools/ab_apply.py 100 target=64,apply_kind=6 target=64,apply_kind=7
