#!/bin/bash
# Round 5: the single-wavefront 16 x 16 Jacobi in isolation -- (a) at 1 / 2 / 3 / 4 resident wavefronts per SIMD (dynamic-LDS padding
# caps the residency; 8192 blocks x 20 decompositions each = steady state), (b) the register-delivered-pivot variant
# (jacobi_regpivot.hpp) and round 3's chain-first variant against the library's loop.  Binaries: see the hipcc line in jacobi_bench.hip
# (+ -DFBX_JACOBI_REGPIVOT / -DFBX_JACOBI_CHAIN_FIRST, -mllvm -amdgpu-sched-strategy=max-ilp as fbx_pgdb.hip is built).
cd "$(dirname "$0")"
for bin in jacobi_bench jacobi_bench_regpivot jacobi_bench_chain_first; do
  echo "== $bin"
  ./$bin 256 1 0            # one wavefront per CU: the bare chain
  ./$bin 1024 1 31744       # one per SIMD, every SIMD busy: the B = 1024 headline's regime
  for pad in 31744 11264 4400 1024; do ./$bin 8192 1 $pad; done
done
