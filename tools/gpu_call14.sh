#!/bin/bash
set -e
python tools/opt_ab.py pair_min 28,20,16,12,8 4096
python tools/opt_ab.py quad_min 40,28,24,20 4096
python tools/opt_ab.py la_rows 3072,2048,1024,0 4096
python tools/opt_ab.py lookahead 2,1 4096
