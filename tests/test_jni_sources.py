"""The Java / JNI side of the boundary exists as source (jni/reflexiv_jni.c, java/...): there is no JDK in the
build container, so instead of compiling it the tests check that the three layers agree -- every C-ABI function
the shim calls is declared in include/reflexiv_hip.h with that many arguments, every JNI export has a `native`
method of the same name and arity in Rfx.java, and every Rfx call in the Java driver exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(*p):
    return open(os.path.join(ROOT, *p)).read()


def strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r'"(?:[^"\\\n]|\\.)*"', '""', src)          # string literals too
    return src.replace("->", ".")                                # (split_args treats < > as brackets for Java generics)


def split_args(s):
    """top-level comma split of an argument list"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{<":
            depth += 1
        elif ch in ")]}>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a for a in out if a.strip() and a.strip() != "void"]


def call_args(src, pos):
    """argument list of the call whose '(' is at src[pos]"""
    depth, i = 0, pos
    while True:
        if src[i] == "(":
            depth += 1
        elif src[i] == ")":
            depth -= 1
            if depth == 0:
                return split_args(src[pos + 1:i])
        i += 1


def header_decls():
    h = strip_comments(read("include", "reflexiv_hip.h"))
    decls = {}
    for m in re.finditer(r"\b(rfx_\w+)\s*\(", h):
        if re.search(r"typedef|struct", h[max(0, m.start() - 20):m.start()]):
            continue
        decls[m.group(1)] = len(call_args(h, m.end() - 1))
    return decls


def test_shim_calls_only_declared_entry_points_with_the_right_arity():
    decls = header_decls()
    c = strip_comments(read("jni", "reflexiv_jni.c"))
    used = {}
    for m in re.finditer(r"\b(rfx_\w+)\s*\(", c):
        name = m.group(1)
        if name in ("rfx_ctx", "rfx_params", "rfx_records"):
            continue
        used.setdefault(name, set()).add(len(call_args(c, m.end() - 1)))
    assert len(used) >= 16
    for name, arities in used.items():
        assert name in decls, f"{name} is not declared in include/reflexiv_hip.h"
        assert arities == {decls[name]}, (name, arities, decls[name])
    # every record operator of the path is bound
    for need in ("rfx_extract_canon", "rfx_count_filter", "rfx_rc_expand_subkmer", "rfx_sort_records",
                 "rfx_fork_filter_forward", "rfx_reflect_from_forward", "rfx_fork_filter_reflected",
                 "rfx_random_reflection", "rfx_extend_pass", "rfx_extend_pass_w", "rfx_contigs_text", "rfx_assemble_reads",
                 "rfx_extract_canon_w", "rfx_count_filter_w"):
        assert need in used, need


def jni_exports():
    c = strip_comments(read("jni", "reflexiv_jni.c"))
    out = {}
    for m in re.finditer(r"RFX_CLASS\((\w+)\)\s*\(", c):
        if m.group(1) == "name":
            continue
        out[m.group(1)] = len(call_args(c, m.end() - 1)) - 2          # minus JNIEnv*, jclass
    return out


def java_natives():
    j = strip_comments(read("java", "uni", "bielefeld", "cmg", "reflexiv", "gpu", "Rfx.java"))
    out = {}
    for m in re.finditer(r"public\s+static\s+native\s+[\w\[\]]+\s+(\w+)\s*\(", j):
        out[m.group(1)] = len(call_args(j, m.end() - 1))
    return out


def test_jni_exports_match_the_native_methods_of_rfx_java():
    ex, nat = jni_exports(), java_natives()
    assert ex == nat, (set(ex) ^ set(nat), {k: (ex[k], nat[k]) for k in ex if k in nat and ex[k] != nat[k]})
    assert len(ex) >= 17


def test_java_sources_use_only_existing_native_methods_and_record_fields():
    nat = java_natives()
    statics = {"ctxForThisTask": 1, "setGpuCount": 1}
    for path in (("java", "uni", "bielefeld", "cmg", "reflexiv", "pipeline", "ReflexivGpuMain.java"),):
        j = strip_comments(read(*path))
        calls = list(re.finditer(r"\bRfx\.(\w+)\s*\(", j))
        assert len(calls) >= 12
        for m in calls:
            name = m.group(1)
            n = len(call_args(j, m.end() - 1))
            if name in statics:
                assert n == statics[name], name
            else:
                assert name in nat and nat[name] == n, (name, n, nat.get(name))
    # the field names the shim looks up exist in RfxRecords with the JNI signatures it asks for
    rec = strip_comments(read("java", "uni", "bielefeld", "cmg", "reflexiv", "gpu", "RfxRecords.java"))
    c = read("jni", "reflexiv_jni.c")
    sig = {"J": "long", "I": "int", "[J": "long[]", "[I": "int[]"}
    for m in re.finditer(r'GetFieldID\(env, c, "(\w+)", "(\[?[JI])"\)', c):
        assert re.search(r"public\s+%s\s+%s\b" % (re.escape(sig[m.group(2)]), m.group(1)), rec), m.groups()


def test_driver_keeps_the_reference_operator_sequence():
    """the order of RDD operators in ReflexivGpuMain.assembly() is the one of ReflexivMain.assembly() (:147-310)"""
    j = strip_comments(read("java", "uni", "bielefeld", "cmg", "reflexiv", "pipeline", "ReflexivGpuMain.java"))
    body = j[j.index("public void assembly()"):j.index("public void assemblyResident()")]
    seq = re.findall(r"new (\w+)\(\)|\.(sortByKey|reduceByKey|zipWithIndex|saveAsTextFile|count|coalesce)\(", body)
    flat = [a or b for a, b in seq]
    want = ["FastqFilterWithQual", "FastqUnitFilter", "ReverseComplementKmerBinaryExtraction", "reduceByKey", "KmerCounting",
            "KmerCoverageFilter", "KmerReverseComplementAndForwardSubKmerExtraction", "sortByKey", "FilterForkSubKmer",
            "ReflectedSubKmerExtractionFromForward", "sortByKey", "FilterForkReflectedSubKmer", "kmerRandomReflection",
            "sortByKey", "ExtendReflexivKmer", "sortByKey", "sortByKey", "ExtendReflexivKmerToArrayFirstTime",
            "ExtendReflexivKmerToArrayLoop", "count", "coalesce", "sortByKey", "KmerToContig", "zipWithIndex", "TagContigID",
            "saveAsTextFile"]
    flat = [x for x in flat if x not in ("JavaSparkContext",)]
    assert flat == want, flat


def test_jni_shim_compiles_against_the_jni_prototypes():
    """gcc -fsyntax-only of jni/reflexiv_jni.c against tests/jni_stub/jni.h (the JNI specification's prototypes for the
    functions the shim uses; test-only, there is no JDK here): argument counts and types of every JNI and rfx_* call."""
    import subprocess
    r = subprocess.run(["gcc", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "tests", "jni_stub"),
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "reflexiv_jni.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_no_critical_region_around_gpu_work():
    """GetPrimitiveArrayCritical blocks the collector and forbids blocking calls; every native method here waits for the GPU
    (the sharded ones for other tasks too: ADVICE r03, a deadlock with several tasks in one JVM)."""
    src = strip_comments(read("jni", "reflexiv_jni.c"))
    assert "PrimitiveArrayCritical" not in src
