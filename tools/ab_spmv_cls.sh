#!/bin/bash
python tools/ab_spmv.py 214 spmv_classes=0 spmv_classes=1
python tools/ab_spmv.py 107 spmv_pattern=0 spmv_pattern=1
python tools/gpu_timing.py 2>&1 | tail -12
