#!/bin/bash
OUT=gpurun_out/r6i; mkdir -p $OUT
cp dnascent_amd/lib/libdnascent_hip.so /tmp/lib_keep.so
cp tools/_bin/lib_wstrace/libdnascent_hip.so dnascent_amd/lib/libdnascent_hip.so
python3 tools/ws_trace.py > $OUT/ws_trace.txt 2>&1
cp /tmp/lib_keep.so dnascent_amd/lib/libdnascent_hip.so
cat $OUT/ws_trace.txt | head -60
