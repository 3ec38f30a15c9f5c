#!/bin/bash
# Evidence behind DESIGN.md's prefill-GEMM section, one box, one call:
#   gpurun_out/rNN_sqgemm_ablation.txt   the phased SmoothQuant GEMM (gemm_sqp.hip) and its compile-time ablations on the QKV
#                                        shape at K = 128 / 4096 / 8192 (fixed cost and per-K-tile slope), with the shader
#                                        clock every variant held (tllm_gemm_set_clock_probe)
#   gpurun_out/rNN_sqgemm_shapes.txt     production shapes (heuristic) and the alternatives on the four 7B projections
#   gpurun_out/rNN_lds_mfma_probe.txt    MFMA rate next to fragment reads / LDS-DMA without any barrier (csrc/tools/lds_mfma_probe.cpp)
#   gpurun_out/rNN_sqgemm_pmc.txt        SQ counters of the lock-step and the phased 256 x 192 kernels (tools/sqp_pmc.sh)
set -u
R=${1:-r02}
mkdir -p gpurun_out
{
  echo "# cfg ids: 6 = lock-step 256x192 (gemm_glds.hip), 20 = phased 256x192 (gemm_sqp.hip, production for QKV / gate / up);"
  echo "# ablations of 20 (results wrong on purpose): 31 no barriers, 32 no DMA waits, 21 no DMA, 23 no fragment reads,"
  echo "# 27 MFMA + barriers only, 33 MFMA only (no barriers), 24 fragment reads + barriers only, 25 DMA + barriers only, 26 no epilogue"
  CLOCKS=1 SHAPES="12288,128;12288,4096;12288,8192" python tools/gemm_sweep.py 1024 6 20 31 32 21 23 27 33 24 25 26
} > gpurun_out/${R}_sqgemm_ablation.txt 2>&1
CLOCKS=1 python tools/gemm_sweep.py 1024 0 6 8 20 15 18 > gpurun_out/${R}_sqgemm_shapes.txt 2>&1
trtllm-llama_amd/csrc/build/lds_mfma_probe > gpurun_out/${R}_lds_mfma_probe.txt 2>&1
tools/sqp_pmc.sh "6 20" > /dev/null 2>&1
mv gpurun_out/sqp_pmc.txt gpurun_out/${R}_sqgemm_pmc.txt
cat gpurun_out/${R}_sqgemm_ablation.txt gpurun_out/${R}_sqgemm_shapes.txt gpurun_out/${R}_lds_mfma_probe.txt
