#!/bin/bash
for p in 2 36; do python tools/dbg_field_err.py $p; done > gpurun_out/$1_dbg_field_err.txt 2>&1; cat gpurun_out/$1_dbg_field_err.txt | grep -v "Warn\|warn"
