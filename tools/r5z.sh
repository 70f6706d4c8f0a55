#!/bin/bash
mkdir -p gpurun_out/r5z; tools/_bin/ubench_v4 > gpurun_out/r5z/ubench_v4.txt 2>&1; cat gpurun_out/r5z/ubench_v4.txt
