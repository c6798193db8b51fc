#!/bin/bash
# Round 5: where the blocks of the 16 x 16 Jacobi travel between rounds -- the library's loop before round 5 (matrix and eigenvector
# blocks through LDS; -DFBX_JACOBI_V_THROUGH_LDS), the library's loop now (eigenvector columns through DPP row shifts), and the
# all-register experiment (jacobi_allreg.hpp: matrix block too, pivots never in LDS) -- at 1 / 2 / 3 / 4 resident wavefronts per
# SIMD; default instruction scheduling (as fbx_pgdb_lean.hip is built) and, for one wavefront per SIMD, max-ILP (fbx_pgdb.hip).
cd "$(dirname "$0")"
for bin in jacobi_bench_default jacobi_bench_vdpp jacobi_bench_allreg; do
  echo "== $bin"
  ./$bin 1024 1 31744
  for pad in 31744 11264 4400 1024; do ./$bin 8192 1 $pad; done
done
for bin in jacobi_bench jacobi_bench_vdpp_ilp jacobi_bench_allreg_ilp; do
  echo "== $bin (max-ilp)"
  ./$bin 256 1 0
  ./$bin 1024 1 31744
  ./$bin 8192 1 11264
done
