#!/bin/bash
# Same-box A/B of library variants (scenedreamer_amd/lib/variants/*.so): bash tools/ab_libs.sh "<python command>" A B C ...
cmd=$1; shift
L=$GRAFT_REPO_ROOT/scenedreamer_amd/lib
cp $L/libsdnative.so /tmp/orig.so
for rep in 1 2; do for v in "$@"; do cp $L/variants/$v.so $L/libsdnative.so; echo "== $v (rep $rep)"; $cmd 2>&1 | grep -E "pose|cnn total|value"; done; done
cp /tmp/orig.so $L/libsdnative.so
