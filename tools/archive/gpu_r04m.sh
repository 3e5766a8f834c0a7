#!/bin/bash
# Round 4, session m: does a second independent launch hide the ramp and tail of a stage launch?  (tools/concurrency_probe.py)
mkdir -p gpurun_out/r04m
{
python tools/concurrency_probe.py --nx 707 --ny 354 --nx2 707 --ny2 354
python tools/concurrency_probe.py --nx 816 --ny 408 --nx2 577 --ny2 289
python tools/concurrency_probe.py --nx 500 --ny 250 --nx2 500 --ny2 250
python tools/concurrency_probe.py --nx 578 --ny 289 --nx2 408 --ny2 204
python tools/concurrency_probe.py --nx 354 --ny 177 --nx2 354 --ny2 177
} > gpurun_out/r04m/probe.txt 2>&1
cat gpurun_out/r04m/probe.txt
