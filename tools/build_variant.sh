#!/bin/bash
# Build the current sources as a named library variant for same-box A/B runs (tools/ab_libs.sh), leaving libsdnative.so as the
# product build:  bash tools/build_variant.sh <name> [ablation]
set -e
cd "$(dirname "$0")/.."
L=scenedreamer_amd/lib
mkdir -p $L/variants
if [ "$2" = ablation ]; then SDN_MLP_ABLATION=1 python -m scenedreamer_amd.build > /dev/null 2>&1; cp $L/libsdnative.so $L/variants/$1.so; fi
python -m scenedreamer_amd.build > /dev/null 2>&1
[ "$2" = ablation ] || cp $L/libsdnative.so $L/variants/$1.so
ls -la $L/variants/$1.so
