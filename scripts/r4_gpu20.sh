#!/bin/bash
for i in 1 2; do
AB_TAG=base python scripts/ab_headline.py 2>&1 | tail -1
KLARA_HIP_LIB=klara.jl_amd/lib/libklara_hip_nofold.so AB_TAG=nofold python scripts/ab_headline.py 2>&1 | tail -1
done
