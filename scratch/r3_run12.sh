#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scratch/r3_ab.sh libmedgp_prio22.so libmedgp_prio22.so
timeout 2400 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -4
