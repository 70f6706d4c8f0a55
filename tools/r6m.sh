#!/bin/bash
OUT=gpurun_out/r6m; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python tools/variant_check.py > $OUT/variant_check.txt 2>&1; tail -3 $OUT/variant_check.txt
timeout 1200 python -m pytest tests -x -q -m gpu -k "cnn or golden or edges" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
