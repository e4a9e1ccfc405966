"""The dynamic-k record format and passes: the oracle (oracle/reflexiv_dynamic.c) against vectors made by
the REFERENCE'S OWN classes of P/ReflexivDSDynamicKmerFirstFour.java and P/ReflexivDSDynamicKmerIteration.java
(tests/golden/dynamic_vectors.npz, written by tests/golden/make_dynamic_vectors.py through tools/java2py.py): the rows after
EVERY operator of both drivers, keys of mixed lengths (k in 23..95), P in {1, 2, 3}, start iterations below and above 61."""
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = os.path.join(HERE, "golden", "dynamic_vectors.npz")


def cases():
    z = np.load(VEC)
    return sorted({k.split("/")[0] for k in z.files})


def rows_of(z, name):
    text = bytes(z[name]).decode()
    return [tuple(ln.split(",")) for ln in text.split("\n") if ln]


@pytest.mark.parametrize("case", cases())
def test_dynamic_passes_equal_the_reference_classes(case):
    z = np.load(VEC)
    P, start, end = (int(x) for x in z[case + "/meta"])
    in_rows = rows_of(z, case + "/in")
    ff, tr1 = O.dyn_first_four(in_rows, P)
    for tag, rows in tr1:
        want = rows_of(z, f"{case}/{tag}")
        assert rows == want, (case, tag, next((i, a, b) for i, (a, b) in enumerate(zip(rows, want)) if a != b) if len(rows) == len(want)
                              else (len(rows), len(want)))
    fin, tr2 = O.dyn_iterations(ff, P, start, end)
    for tag, rows in tr2:
        want = rows_of(z, f"{case}/{tag}")
        assert rows == want, (case, tag, next((i, a, b) for i, (a, b) in enumerate(zip(rows, want)) if a != b) if len(rows) == len(want)
                              else (len(rows), len(want)))
    assert fin == rows_of(z, case + "/final")
