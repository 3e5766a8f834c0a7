#!/bin/bash
# Round 4, session v: chunked stepping on one device for meshes beyond the Infinity Cache, measured (tools/chunkbench.py)
mkdir -p gpurun_out/r04v
{
python tools/chunkbench.py --nx 200 --ny 100 --chunks 4 --steps 7 --check
python tools/chunkbench.py --nx 1414 --ny 707 --chunks 2
python tools/chunkbench.py --nx 1414 --ny 707 --chunks 3
python tools/chunkbench.py --nx 1414 --ny 707 --chunks 4
python tools/chunkbench.py --nx 2000 --ny 1000 --chunks 4
python tools/chunkbench.py --nx 2000 --ny 1000 --chunks 6
python tools/chunkbench.py --nx 2000 --ny 1000 --chunks 8
python tools/chunkbench.py --nx 2828 --ny 1414 --chunks 12
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r04v/chunkbench.txt
cat gpurun_out/r04v/chunkbench.txt
