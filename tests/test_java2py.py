"""Conformance of tools/java2py.py -- the translator that runs the reference's own operator classes to make
tests/golden/{reference,dedup,dynamic}_vectors.npz -- against facts of the Java Language Specification, INDEPENDENTLY of
the oracle: every expected value below is written out by hand from the JLS rule it cites (or is plain two's-complement
arithmetic), none is computed by oracle/ or by the code under test.

Part 1: the integer model `J` and the runtime shims, fact by fact.
Part 2: one tiny hand-written Java class per statement / expression kind the operator classes use, translated from its
        source TEXT exactly as the generators translate the reference's classes, with a known answer.
Part 3 (skipped when /root/reference is absent, e.g. on the GPU box): the three fixture generators re-run in a scratch
        directory give files bit-identical to the committed ones.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import java2py as jp  # noqa: E402

J, Ch = jp.J, jp.Ch


def I(v): return J(v, 32)      # noqa: E704, E743
def L(v): return J(v, 64)      # noqa: E704


def same(x, v, w):
    assert isinstance(x, J) and x.w == w and x.v == v, (x, getattr(x, "w", None), v, w)


# ----------------------------------------------------------------------------------------------- part 1: J and the shims
def test_shift_counts_are_masked_jls_15_19():
    # "only the five lowest-order bits of the right-hand operand are used" (int), six for long
    same(I(1) << I(33), 2, 32)
    same(I(1) << I(32), 1, 32)
    same(L(1) << I(65), 2, 64)
    same(L(1) << I(64), 1, 64)
    same(I(1) << I(-1), -2147483648, 32)               # -1 & 31 = 31
    same(L(1) << I(-1), -9223372036854775808, 64)      # -1 & 63 = 63
    same(I(-8) >> I(1), -4, 32)                        # arithmetic
    same(I(-8) >> I(33), -4, 32)
    same(jp._ushr(I(-8), I(28)), 15, 32)               # logical on 32 bits
    same(jp._ushr(L(-8), I(60)), 15, 64)               # logical on 64 bits
    same(jp._ushr(L(-1), I(64)), -1, 64)               # count 64 = 0
    same(jp._ushr(I(-1), I(32)), -1, 32)
    # the type of a shift is the PROMOTED LEFT operand's alone: int << long stays int, count masked with 31
    same(I(1) << L(35), 8, 32)
    same(L(1) << L(35), 34359738368, 64)


def test_overflow_wraps_and_binary_promotion_jls_5_6_2():
    same(I(2147483647) + I(1), -2147483648, 32)
    same(I(-2147483648) - I(1), 2147483647, 32)
    same(I(65536) * I(65536), 0, 32)
    same(I(46341) * I(46341), -2147479015, 32)
    same(L(9223372036854775807) + L(1), -9223372036854775808, 64)
    same(I(2147483647) + L(1), 2147483648, 64)         # int + long -> long
    same(I(-1) & L(0xFFFFFFFF), 4294967295, 64)        # the int is SIGN-extended first
    same(I(5) | L(1 << 40), (1 << 40) | 5, 64)
    same(-I(-2147483648), -2147483648, 32)             # negation overflows
    same(~I(0), -1, 32)
    same(~L(0), -1, 64)


def test_division_and_remainder_truncate_towards_zero_jls_15_17():
    same(I(7) / I(2), 3, 32)
    same(I(-7) / I(2), -3, 32)
    same(I(7) / I(-2), -3, 32)
    same(I(-7) % I(2), -1, 32)                         # sign of the dividend
    same(I(7) % I(-2), 1, 32)
    same(L(-7) / L(2), -3, 64)
    same(L(-7) % L(2), -1, 64)
    same(I(-2147483648) / I(-1), -2147483648, 32)      # the one overflowing quotient
    same(L(61) / I(31), 1, 64)
    same(L(62) % I(31), 0, 64)


def test_casts_jls_5_1():
    same(jp._cast_int(L(0x1FFFFFFFF)), -1, 32)         # narrowing keeps the low 32 bits
    same(jp._cast_int(L(0x80000000)), -2147483648, 32)
    same(jp._cast_long(I(-1)), -1, 64)                 # widening sign-extends
    same(jp._cast_long(I(-2147483648)), -2147483648, 64)
    assert jp._cast_char(I(65)) == Ch("A")
    assert jp._cast_char(I(65 + 65536)) == Ch("A")     # low 16 bits
    same(jp._cast_int(Ch("A")), 65, 32)
    same(jp._cast_long(Ch("T")), 84, 64)


def test_number_of_leading_zeros_and_friends():
    same(jp._Long.numberOfLeadingZeros(L(0)), 64, 32)
    same(jp._Long.numberOfLeadingZeros(L(1)), 63, 32)
    same(jp._Long.numberOfLeadingZeros(L(-1)), 0, 32)
    same(jp._Long.numberOfLeadingZeros(L(1 << 62)), 1, 32)
    same(jp._Long.numberOfTrailingZeros(L(0)), 64, 32)
    same(jp._Long.numberOfTrailingZeros(L(8)), 3, 32)
    same(jp._Integer.numberOfLeadingZeros(I(0)), 32, 32)
    same(jp._Integer.numberOfLeadingZeros(I(-1)), 0, 32)
    same(jp._Long.SIZE, 64, 32)
    same(jp._Long.MAX_VALUE, 9223372036854775807, 64)
    same(jp._Long.MIN_VALUE, -9223372036854775808, 64)
    same(jp._Integer.MAX_VALUE, 2147483647, 32)
    same(jp._Long.parseLong("-9223372036854775808"), -9223372036854775808, 64)
    same(jp._Integer.parseInt("12"), 12, 32)
    with pytest.raises(jp._JavaThrow):
        jp._Integer.parseInt("1x")
    same(jp._Math.abs(I(-2147483648)), -2147483648, 32)   # Math.abs(Integer.MIN_VALUE) is itself
    same(jp._Math.max(I(3), L(2)), 3, 64)
    same(jp._Math.min(I(3), I(-2)), -2, 32)


def test_char_arithmetic_and_string_concatenation_jls_15_18_1():
    same(jp._add(Ch("A"), I(1)), 66, 32)               # char + int -> int
    same(Ch("C") - Ch("A"), 2, 32)                     # char - char -> int
    same(Ch("a") - I(32), 65, 32)
    assert Ch("A") < Ch("C") and Ch("T") >= Ch("T") and Ch("A") == Ch("A") and Ch("A") != Ch("C")
    assert Ch("A") == I(65)                            # numeric comparison of char and int
    assert jp._add("x", I(1)) == "x1"
    assert jp._add(jp._add("", L(-5)), Ch("N")) == "-5N"
    assert jp._add("b=", True) == "b=true" and jp._add("n=", None) == "n=null"
    assert jp._add(jp._add(I(1), I(2)), "s") == "3s"   # left to right: (1 + 2) + "s"
    assert jp._add("s", jp._add(I(1), I(2))) == "s3"
    # String methods the classes use
    assert jp._call("ACGT", "charAt", I(2)) == Ch("G")
    assert jp._call("ACGT", "substring", I(1), I(3)) == "CG"
    assert jp._call("a,b,,", "split", ",") == ["a", "b"]          # trailing empty strings are dropped
    same(jp._call("Ab", "hashCode"), 65 * 31 + 98, 32)
    same(jp._call("ACGT", "length"), 4, 32)
    with pytest.raises(jp._JavaThrow):
        jp._call("ACGT", "charAt", I(4))


def test_equality_and_boxing():
    assert jp._eq(I(3), L(3))                          # numeric ==, promoted
    assert not jp._eq(I(3), I(4))
    assert jp._eq(None, None) and not jp._eq(None, I(0))
    assert jp._call(L(3), "equals", L(3)) and not jp._call(L(3), "equals", I(3))     # Long.equals(Integer) is false
    same(jp._call(L((5 << 32) | 3), "hashCode"), 5 ^ 3, 32)
    with pytest.raises(jp._JavaThrow):
        jp._box_long(I(3))                             # (Long) of an Integer: ClassCastException
    with pytest.raises(jp._JavaThrow):
        jp._newarr("long", I(-1))
    a = jp._newarr("long", I(3))
    assert len(a) == 3 and all(x.w == 64 and x.v == 0 for x in a)
    same(jp._len(a), 3, 32)


def test_collections_shims():
    lst = jp._new("ArrayList")
    lst.add(L(1)); lst.add(L(2)); lst.add(I(0), L(0))          # add(index, element)
    assert [x.v for x in lst.items] == [0, 1, 2]
    assert lst.remove(I(0)).v == 0 and lst.size().v == 2       # remove(int index)
    assert lst.get(I(1)).v == 2
    with pytest.raises(Exception):
        lst.get(I(5))
    m = jp._new("HashMap")
    k1, k2 = [L(1)], [L(1)]
    m.put(k1, I(7))
    assert m.containsKey(k1) and not m.containsKey(k2)         # arrays hash by identity
    m.put(L(5), I(1)); m.put(L(5), I(2))
    assert m.get(L(5)).v == 2
    row = jp._RowFactory.create(L(9), I(2), "s", jp.Seq([L(4), L(5)]))
    same(row.getLong(I(0)), 9, 64)
    same(row.getInt(I(1)), 2, 32)
    assert row.getString(I(2)) == "s" and row.getSeq(I(3)).apply(I(1)).v == 5 and row.getSeq(I(3)).length().v == 2
    t = jp._new("Tuple2", L(1), "x")
    assert jp._call(t, "_1").v == 1 and jp._call(t, "_2") == "x"


# ----------------------------------------------------------------------------------------------- part 2: translated classes
JAVA = r'''
public class Conf {
    long field = 7L;
    int[] table = new int[]{1, 2, 3};

    public long shl(long a, int n) { return a << n; }
    public int ishl(int a, int n) { return a << n; }
    public long widenReturn(int a) { return a; }                       // returns are converted to the declared type
    public long widenThenShift(int a) { return widenReturn(a) << 40; }
    public int compoundNarrow(int a, long b) { int x = a; x += b; return x; }       // JLS 15.26.2: x = (int)(x + b)
    public int compoundShift(int a) { int x = a; x <<= 33; return x; }
    public long compoundLong(long a) { long x = a; x >>>= 60; x |= 1L << 40; x ^= 3; x -= 1; x *= 2; x /= 3; x %= 1000; return x; }
    public int charCompound() { char c = 'A'; c += 2; return c; }                    // c = (char)(c + 2)
    public int prec(int a, int b, int c) { return a & b >>> c; }                     // a & (b >>> c)
    public int prec2(int a, int b) { return a + b << 2; }                            // (a + b) << 2
    public boolean prec3(int a, int b) { return a == b || a > 0 && b < 0; }          // a == b || (a > 0 && b < 0)
    public int prec4(int a) { return a | 1 ^ 3 & 2; }                                // a | (1 ^ (3 & 2))
    public int ternary(int a) { return a > 0 ? a > 10 ? 2 : 1 : 0; }                 // right-associative
    public int incdec() { int i = 5; i++; ++i; i--; int j = i; j += i; return j; }
    public int loops(int n) {
        int s = 0;
        for (int i = 0; i < n; i++) { if (i == 2) continue; if (i == 7) break; s += i; }
        int k = 0;
        while (k < 3) { k++; s += 100; }
        do { s += 1000; k--; } while (k > 0);
        for (int v : table) s += v;
        return s;
    }
    public int continueInFor(int n) { int c = 0; for (int i = 0; i < n; i++) { if (i % 2 == 0) continue; c++; } return c; }
    public long arrays(int n) {
        long[] a = new long[n];
        for (int i = 0; i < a.length; i++) a[i] = (long) i << 31;
        long[][] m = new long[2][3];
        m[1][2] = a[n - 1];
        a[0] += 5;
        a[1]++;
        return m[1][2] + a[0] + a[1] + m[0][0];
    }
    public long fields(long x) { field = field + x; this.field++; return field; }
    public int intDivision(int a, int b) { return a / b * b + a % b; }                // == a
    public long mixed(int a, long b) { return a * b + a / 2; }
    public int castInt(long a) { return (int) (a >>> 2); }
    public long castLong(int a) { return (long) a << 32 >>> 32; }
    public long signExtend(int a) { long x = a; return x; }
    public int chars(String s) { int n = 0; for (int i = 0; i < s.length(); i++) { char c = s.charAt(i); if (c == 'A') n += 0; else if (c == 'C') n += 1; else if (c == 'G') n += 2; else n += 3; n = n << 2 >>> 1; } return n; }
    public String strings(int a, long b) { String s = ""; s += a; s = s + "," + b; s += 'x'; return s + (a + b); }
    public long nlz(long a) { return Long.numberOfLeadingZeros(a) / 2 + 1; }
    public int tryCatch(String s) { try { return Integer.parseInt(s); } catch (Exception e) { return -1; } }
    public long listAndRow(long v) {
        ArrayList<Row> rows = new ArrayList<Row>();
        long[] arr = new long[2]; arr[0] = v; arr[1] = v + 1;
        rows.add(RowFactory.create(v, 3, JavaConverters.collectionAsScalaIterableConverter(Arrays.asList(arr)).asScala().toSeq()));
        Row r = rows.get(0);
        Seq q = r.getSeq(2);
        long e = (Long) q.apply(1);
        return r.getLong(0) * 1000 + r.getInt(1) * 100 + e + q.length() + rows.size();
    }
    public int switchless(int marker) { if (marker == 1) { return 10; } else if (marker == 2) { return 20; } return 0; }
    public boolean logic(boolean a, boolean b) { return !a && (b || a) ; }
    public int hex() { return 0xFF + 010 + 7; }                                       // 255 + 8 (octal) + 7
    public long longLiteral() { return 0x7FFFFFFFFFFFFFFFL + 1L; }
    public int negShift(long a) { return (int) (a >>> -2); }                          // count -2 & 63 = 62
}
'''


@pytest.fixture(scope="module")
def conf(tmp_path_factory):
    p = tmp_path_factory.mktemp("j") / "Conf.java"
    p.write_text(JAVA)
    return jp.translate_plain_class(str(p), "Conf")(None)


def test_translated_shifts_returns_and_compound_assignment(conf):
    same(conf.shl(L(1), I(65)), 2, 64)
    same(conf.ishl(I(1), I(33)), 2, 32)
    same(conf.widenReturn(I(-1)), -1, 64)
    same(conf.widenThenShift(I(3)), 3 << 40, 64)            # would be 3 << 8 if the int were not widened by the return
    same(conf.compoundNarrow(I(5), L((1 << 40) + 7)), 12, 32)
    same(conf.compoundNarrow(I(2147483647), L(1)), -2147483648, 32)
    same(conf.compoundShift(I(1)), 2, 32)
    # -1 >>> 60 = 15; | 1<<40; ^3 -> 12; -1 -> 11 (+2^40); *2; /3; %1000: ((2^40 + 11) * 2 / 3) % 1000
    same(conf.compoundLong(L(-1)), (((1 << 40) + 11) * 2 // 3) % 1000, 64)
    same(conf.charCompound(), 67, 32)
    same(conf.incdec(), 12, 32)


def test_translated_precedence(conf):
    same(conf.prec(I(0xFF), I(-16), I(28)), 0xFF & 15, 32)
    same(conf.prec2(I(1), I(2)), 12, 32)
    assert conf.prec3(I(1), I(1)) is True and conf.prec3(I(1), I(-1)) is True and conf.prec3(I(-1), I(1)) is False
    same(conf.prec4(I(8)), 8 | (1 ^ (3 & 2)), 32)
    assert [conf.ternary(I(v)).v for v in (-1, 5, 11)] == [0, 1, 2]
    same(conf.hex(), 270, 32)
    same(conf.longLiteral(), -9223372036854775808, 64)
    same(conf.negShift(L(-1)), 3, 32)


def test_translated_control_flow_and_arrays(conf):
    # for: 0 + 1 + 3 + 4 + 5 + 6 = 19; while: +300; do-while: k = 3 -> three rounds = +3000; for-each: +6
    same(conf.loops(I(10)), 19 + 300 + 3000 + 6, 32)
    same(conf.continueInFor(I(7)), 3, 32)                   # the update still runs after `continue`
    # a = [0 + 5, (1 << 31) + 1, 2 << 31], m[1][2] = 2 << 31
    same(conf.arrays(I(3)), (2 << 31) + 5 + (1 << 31) + 1, 64)
    same(conf.fields(L(3)), 11, 64)
    same(conf.fields(L(0)), 12, 64)                         # state persists in the object
    assert [conf.intDivision(I(a), I(b)).v for a, b in ((7, 2), (-7, 2), (7, -2), (-7, -2))] == [7, -7, 7, -7]
    same(conf.mixed(I(-3), L(1 << 33)), -3 * (1 << 33) - 1, 64)
    same(conf.castInt(L(-4)), -1, 32)                       # (-4 >>> 2) = 0x3FFF...F -> low 32 bits = -1
    same(conf.castLong(I(-1)), 4294967295, 64)              # ((long) a << 32) >>> 32: the cast binds tighter than <<
    same(conf.signExtend(I(-5)), -5, 64)
    assert [conf.switchless(I(v)).v for v in (1, 2, 3)] == [10, 20, 0]
    assert conf.logic(False, True) is True and conf.logic(True, True) is False


def test_translated_strings_rows_and_exceptions(conf):
    # A0 C1 G2 else 3, each step n = (n << 2) >>> 1:  "ACGT" -> 0; (0+1)*2 = 2; (2+2)*2 = 8; (8+3)*2 = 22
    same(conf.chars("ACGT"), 22, 32)
    assert conf.strings(I(1), L(2)) == "1,2x3"
    same(conf.nlz(L(0)), 33, 64)                            # 64 / 2 + 1, widened by the long return
    same(conf.nlz(L(1 << 40)), 23 // 2 + 1, 64)
    same(conf.tryCatch("42"), 42, 32)
    same(conf.tryCatch("4x"), -1, 32)
    same(conf.listAndRow(L(9)), 9000 + 300 + 10 + 2 + 1, 64)


def test_translator_is_strict():
    """what it does not know raises at translation time; a long stored into an int without a cast (javac rejects it) raises
    at run time -- nothing is guessed"""
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "Bad.java")
        open(p, "w").write("public class Bad { public int f(int a) { switch (a) { case 1: return 2; } return 0; } }")
        with pytest.raises(SyntaxError):
            jp.translate_plain_class(p, "Bad")
        open(p, "w").write("public class Bad { public int f(long a) { int x = 0; x = a; return x; } }")
        with pytest.raises(TypeError):
            jp.translate_plain_class(p, "Bad")(None).f(L(1))


# ----------------------------------------------------------------------------------------------- part 3: the vectors themselves
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's sources are not on this box")
def test_regenerated_vectors_equal_the_committed_files(tmp_path):
    """The three fixture generators run again (the reference's classes translated and executed afresh, side by side to keep
    the CPU suite short) must reproduce the committed .npz files array for array: the fixtures are what the generators make
    from the reference TODAY, with today's translator -- not a leftover."""
    golden = os.path.join(ROOT, "tests", "golden")
    jobs = [("make_dedup_vectors.py", "dedup_vectors.npz", ["--jobs", "5"]), ("make_dynamic_vectors.py", "dynamic_vectors.npz", []),
            ("make_reference_vectors.py", "reference_vectors.npz", [])]
    procs = [(name, subprocess.Popen([sys.executable, os.path.join(golden, script), "--out", str(tmp_path / name)] + extra, cwd=ROOT,
                                     stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)) for script, name, extra in jobs]
    try:
        for name, p in procs:
            err = p.communicate(timeout=1500)[1]
            assert p.returncode == 0, (name, err[-3000:])
    finally:
        for _, p in procs:                                      # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    for _, name, _ in jobs:
        new, old = np.load(tmp_path / name, allow_pickle=False), np.load(os.path.join(golden, name), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), name
        for f in old.files:
            assert new[f].dtype == old[f].dtype and new[f].shape == old[f].shape and np.array_equal(new[f], old[f]), (name, f)
