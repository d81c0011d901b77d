#!/bin/bash
set -e
for n in 384 512 640 768; do python tools/opt_ab.py split_min_blocks 2,3,4,5,6,8 $n 300 | sort | awk '{print}' ; done
