#!/bin/bash
python tools/ab_spmv.py 214 spmv_classes=0 spmv_classes=1
python tools/ab_spmv.py 107 spmv_pattern=2,spmv_classes=0 spmv_pattern=2,spmv_classes=1 spmv_pattern=1
