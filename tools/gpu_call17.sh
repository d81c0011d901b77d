#!/bin/bash
set -e
for v in potrf_stamps potrf_stamps_noprio potrf_stamps_sleep16; do echo "== $v"; timeout -k 5 60 ./tools/$v | grep -A9 "NB=128" | grep "stage\|NB="; timeout -k 5 60 ./tools/$v | grep "whole kernel"; done
