#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python scripts/pgdb1_streams_time.py 20 > gpurun_out/t4_p1_streams.log 2>&1
cat gpurun_out/t4_p1_streams.log
