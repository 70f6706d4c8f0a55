"""The product path has no CPU fallback and never touches the oracle (test infrastructure): checked statically and, on a machine
without a GPU, by behaviour."""
import os
import re

import pytest

from dnascent_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_package_never_references_the_oracle():
    pat = re.compile(r"pyoracle|dn_oracle|liboracle|oracle/|import oracle|from oracle")
    hits = []
    for d, _, files in os.walk(os.path.join(ROOT, "dnascent_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".c", ".h", ".hip")):
                p = os.path.join(d, f)
                if f == "build.py":
                    continue                                # build() compiles the checker next to the product; building it is not using it
                for n, line in enumerate(open(p, errors="replace"), 1):
                    code = line.split("//")[0].split("#")[0]  # comments may cite the oracle, code may not
                    if pat.search(code):
                        hits.append("%s:%d: %s" % (os.path.relpath(p, ROOT), n, line.strip()))
    assert not hits, "\n".join(hits)


def test_context_creation_fails_loudly_without_a_gpu():
    if hip.lib().dn_device_count() > 0:
        pytest.skip("a GPU is present: nothing to refuse")
    with pytest.raises(hip.DnError, match="no CPU fallback"):
        hip.Context(0)
