#!/bin/bash
export MEDGP_MULTI_CU=-1
MEDGP_CHOLINV_IL=0 timeout 600 python3 scratch/quick_il.py 2>&1 | grep -v Warn
MEDGP_CHOLINV_IL=1 timeout 600 python3 scratch/quick_il.py 2>&1 | grep -v Warn
