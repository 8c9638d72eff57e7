#!/bin/bash
# A/B of two builds of libsdnative.so on the same box: tools/ab_lib.sh <rounds> <command...>
# runs <command> alternately with scenedreamer_amd/lib/libsdnative_base.so (A) and the current library (B).
L=scenedreamer_amd/lib
rounds=$1; shift
cp $L/libsdnative.so $L/_new.so
for i in $(seq $rounds); do
  cp $L/libsdnative_base.so $L/libsdnative.so; echo "A(base): $("$@" 2>&1 | tail -1)"
  cp $L/_new.so $L/libsdnative.so;            echo "B(new):  $("$@" 2>&1 | tail -1)"
done
