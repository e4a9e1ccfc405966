"""The threaded form of the oracle (bench.py's CPU baseline: one logical partition per task, range-split sorts)
gives exactly what the serial form gives."""
import numpy as np
import pytest

from oracle import oracle as O


@pytest.fixture(autouse=True)
def serial_again():
    yield
    O.set_threads(1)


@pytest.mark.parametrize("k,fc,ec,twin,min_cov", [(31, 0, 0, O.TWIN_DS, 2), (21, 2, 3, O.TWIN_RDD, 1), (5, 0, 0, O.TWIN_DS, 1),
                                                  (31, 0, 0, O.TWIN_DS, 3)])
def test_count_reads_omp_equals_serial(k, fc, ec, twin, min_cov):
    g = O.synth_genome(7, 20_000)
    bases, off = O.synth_reads(7, g, 20_000, 0, 6000, 100)
    km = O.extract_canon(bases, off, k, fc, ec)
    wk, wc, wd = O.count_filter(km, min_cov, 10_000_000, twin)
    for t in (1, 3, 8):
        O.set_threads(t)
        keys, counts, nd, ni = O.count_reads_omp(bases, off, k, min_cov, 10_000_000, twin, fc, ec)
        assert ni == len(km) and nd == wd
        assert np.array_equal(keys, wk) and np.array_equal(counts, wc)


@pytest.mark.parametrize("k,min_cov,clips", [(63, 2, (0, 0)), (47, 1, (2, 3)), (33, 3, (0, 0))])
def test_count_reads_w2_omp_equals_serial_in_passes(k, min_cov, clips):
    g = O.synth_genome(9, 20_000)
    bases, off = O.synth_reads(9, g, 20_000, 0, 5000, 100)
    fc, ec = clips
    km = O.extract_canon_w(bases, off, k, fc, ec)
    wk, wc, wd = O.count_filter_w(km, k, min_cov)
    for t, cuts in ((1, [0, 4096]), (8, [0, 100, 1000, 1001, 4096])):
        O.set_threads(t)
        ks, cs, nd, ni = [], [], 0, 0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a, b, d, i = O.count_reads_omp(bases, off, k, min_cov, front_clip=fc, end_clip=ec, buckets=(lo, hi))
            ks.append(a); cs.append(b); nd += d; ni += i
        assert ni == len(km) and nd == wd
        assert np.array_equal(np.concatenate(ks), wk) and np.array_equal(np.concatenate(cs), wc)


def test_count_reads_omp_in_passes():
    g = O.synth_genome(7, 20_000)
    bases, off = O.synth_reads(7, g, 20_000, 0, 6000, 100)
    wk, wc, wd = O.count_filter(O.extract_canon(bases, off, 31), 2)
    O.set_threads(4)
    ks, cs, nd, ni = [], [], 0, 0
    for lo, hi in ((0, 7), (7, 2000), (2000, 4096)):
        a, b, d, i = O.count_reads_omp(bases, off, 31, 2, buckets=(lo, hi))
        ks.append(a); cs.append(b); nd += d; ni += i
    assert nd == wd and ni == 6000 * 70
    assert np.array_equal(np.concatenate(ks), wk) and np.array_equal(np.concatenate(cs), wc)


@pytest.mark.parametrize("k,P", [(31, 8), (31, 3), (63, 4)])
def test_threaded_driver_equals_serial(k, P):
    G, n_reads = 120_000, 40_000
    g = O.synth_genome(11, G)
    bases, off = O.synth_reads(11, g, G, 0, n_reads, 150)
    if k <= 31:
        keys, counts, _ = O.count_filter(O.extract_canon(bases, off, k), 3)
    else:
        k32, c64, _ = O.count_filter_w(O.extract_canon_w(bases, off, k), k, 3)
        keys, counts = O.counter_to_asm_w(k32, k), c64.astype(np.int32)
    assert len(counts) > (1 << 16)          # large enough for the range-split sort and gather
    prm = O.default_params(k=k, min_cov=3, partitions=P)
    want = O.assemble_from_counts(keys, counts, prm)
    for t in (2, 8):
        O.set_threads(t)
        got = O.assemble_from_counts(keys, counts, prm)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2]
        assert np.array_equal(got[3].key, want[3].key) and np.array_equal(got[3].ext, want[3].ext)
        O.set_threads(1)
