#!/bin/bash
timeout 40 python tools/cpu_pair_parallel.py 256 2>&1 | tail -5
OMP_PROC_BIND=spread OMP_PLACES=cores timeout 40 python tools/cpu_pair_parallel.py 256 2>&1 | tail -4
