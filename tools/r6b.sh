#!/bin/bash
mkdir -p gpurun_out/r6b; tools/_bin/ubench_mfma_lds > gpurun_out/r6b/ubench_mfma_lds.txt 2>&1; cat gpurun_out/r6b/ubench_mfma_lds.txt
