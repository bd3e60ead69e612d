#!/bin/bash
for M in 100 64; do
for o in gmres_dot_gy=1 gmres_dot_gy=2 gmres_dot_gy=3 gmres_dot_gy=4 gmres_dot_gy=0; do echo "M $M $o"; FEDD_SHARE_CELLS=$M python tools/share_walltime.py $o | grep -A6 "no synchron" | grep "^step\|gmres"; done
done
