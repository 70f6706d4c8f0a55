#!/bin/bash
# K3 evidence: per-kernel stats + per-layer table + MFMA utilisation counters of the CNN timing run (64 x 20 kb reads, default math)
mkdir -p gpurun_out/final_k3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 tools/gpu_cnn_time.py 64 20000 f16x3,bf16x6,fp32 > gpurun_out/final_k3/time.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_k3/stats -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > gpurun_out/final_k3/stats.log 2>&1
python3 tools/cnn_layers.py gpurun_out/final_k3/stats/k3_kernel_trace.csv 1200057 > gpurun_out/final_k3/layers.txt
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/final_k3/pmc -o k3 -- python3 tools/gpu_cnn_time.py 64 20000 f16x3 > gpurun_out/final_k3/pmc.log 2>&1
cat gpurun_out/final_k3/time.txt gpurun_out/final_k3/layers.txt
