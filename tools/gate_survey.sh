#!/bin/bash
python tools/gate_survey.py > gpurun_out/$1_gate_survey.jsonl 2> gpurun_out/$1_gate_survey.err; echo rc=$?; cat gpurun_out/$1_gate_survey.jsonl
