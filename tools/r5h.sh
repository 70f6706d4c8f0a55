#!/bin/bash
OUT=gpurun_out/r5h; mkdir -p $OUT
for a in 1 2 3 4 7; do echo "== ablation $a (1 no ring reads, 2 no ring writes, 4 no A-plane round trip)" >> $OUT/abl.txt; timeout 200 tools/_bin/k3_block64_abl$a 1200128 3 2>&1 | grep -E "BLOCK64=2|stage|conv" >> $OUT/abl.txt; done
cat $OUT/abl.txt
