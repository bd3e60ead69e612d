#!/bin/bash
# A/B of the Schwarz apply kernels with the bench's 64-node boxes (0 = batch table k_apply_bt where every box conforms,
# 6 = chunk records k_apply_mfma, 7 = warp-specialised k_apply_ws; apply_span = places per workgroup, 0 = one round with the table)
python tools/ab_apply.py 214 target=64,apply_kind=6 target=64,apply_kind=0 target=64,apply_kind=0,apply_span=96 target=64,apply_kind=0,apply_span=192 target=64,apply_kind=7
python tools/ab_apply.py 107 target=64,apply_kind=6 target=64,apply_kind=0 target=64,apply_kind=7
python tools/ab_apply.py 100 target=64,apply_kind=6 target=64,apply_kind=0 target=64,apply_kind=7
