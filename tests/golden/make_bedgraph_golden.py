#!/usr/bin/env python3
"""Generates tests/golden/bedgraph_input.detect and tests/golden/bedgraph_expected.json.

BUILD CONTAINER ONLY: it runs the reference's own consumer, /root/reference/utils/dnascent2bedgraph.py (plain Python), on a
.detect file written by THIS framework's host layer (header + records of forward / reverse / indel reads, CPU only) and
stores what that script produced -- every bedgraph file it wrote, verbatim -- as the expected parse.  The reference's script is
executed where it lies; nothing of it is copied.  tests/test_detect_grammar.py then checks, without the reference,
  * that the host writer still produces bedgraph_input.detect byte for byte from the same deterministic inputs, and
  * that the restated parser of the test agrees with what the reference's parser extracted.
Run:  python tests/golden/make_bedgraph_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_SCRIPT = "/root/reference/utils/dnascent2bedgraph.py"


def main():
    from test_detect_grammar import detect_text_for_fixture
    from dnascent_amd import synth
    text = detect_text_for_fixture(synth.pore_model())
    path = os.path.join(HERE, "bedgraph_input.detect")
    with open(path, "w") as f:
        f.write(text)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "bg")
        subprocess.run([sys.executable, REF_SCRIPT, "-d", path, "-o", out], check=True, stdout=subprocess.DEVNULL)
        files = {}
        for d, _, fs in os.walk(out):
            for fn in fs:
                files[os.path.relpath(os.path.join(d, fn), out)] = open(os.path.join(d, fn)).read()
    json.dump({"generator": "tests/golden/make_bedgraph_golden.py", "consumer": "utils/dnascent2bedgraph.py (reference, executed in place)",
               "files": files}, open(os.path.join(HERE, "bedgraph_expected.json"), "w"), indent=0, sort_keys=True)
    print("wrote", path, "and bedgraph_expected.json:", len(files), "bedgraph files")


if __name__ == "__main__":
    main()
