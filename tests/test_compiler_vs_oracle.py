"""CPU-side parity of the table compiler (gorp_amd/csrc/gx_compile.cpp) with the
oracle: the tables are decoded from the blob and walked by tests/blob_interp.py
(the kernel contract), so these tests need no GPU.  The -m gpu tests repeat the
comparisons through the real kernels."""
import os
import random

import numpy as np
import pytest

from blob_interp import Blob, units_of
from gorp_amd import _native as N
from gorp_amd.gorp import DefinitionParseException, FlattenedExtraction, Gorp, GorpError, PolyMatcher, RegexHelper
from oracle import oracle as O


def product_blob(autom, jdk):
    from gorp_amd.gorp import _create
    h = _create(autom, jdk, N.GX_CREATE_HOST_ONLY)
    # the table-image builders run for host-only handles too; forcing the record tier (fused and two-pass) runs its
    # self-check -- every (state, class) of the range records against the dense rows -- on this definition as well
    # (a definition with an extraction that is run as a program -- gx_stat 27 -- has the per-line kernel alone: tier 0)
    for flags in (N.GX_CREATE_TIER_RECORDS, N.GX_CREATE_TIER_RECORDS | N.GX_CREATE_NO_FUSED):
        hf = _create(autom, jdk, N.GX_CREATE_HOST_ONLY | flags)
        assert N.lib().gx_stat(hf.ptr, 7) in ((0,) if N.lib().gx_stat(hf.ptr, 27) else (2, 3, 4))
    n = N.lib().gx_blob_size(h.ptr)
    out = np.zeros(n, np.uint8)
    assert N.lib().gx_blob_copy(h.ptr, out.ctypes.data, n) == 0
    return Blob(out)


def extract_both_ways(b, units):
    """Two-phase tables and, when present, the fused single-pass automaton must agree."""
    got = b.extract(units)
    if b.union_ok:
        assert b.extract_union(units) == got, units
    return got


def both(extractions):
    fl = [FlattenedExtraction(e["name"], e["pieces"], e.get("append")) for e in extractions]
    built = [f.build() for f in fl]
    autom, jdk = [b[0] for b in built], [b[1] for b in built]
    # the product's RegexHelper must agree with the oracle's restatement of it
    for e, b in zip(extractions, built):
        assert O.build_regex_strings(e["pieces"]) == b
    return product_blob(autom, jdk), O.OracleGorp(autom, jdk)


def test_regexhelper_golden(golden):
    g = golden("regexhelper")
    for src, exp in g["quoteLiteralAsRegexp"]:
        assert RegexHelper.quoteLiteralAsRegexp(src) == exp
    for src, exp in g["massageRegexpForAutomaton"]:
        assert RegexHelper.massageRegexpForAutomaton(src) == exp
    for src, exp in g["massageRegexpForJDK"]:
        assert RegexHelper.massageRegexpForJDK(src) == exp
    with pytest.raises(ValueError, match="Unrecognized backslash escape"):
        RegexHelper.massageRegexpForAutomaton("\\q")
    with pytest.raises(ValueError, match="negated character class"):
        RegexHelper.massageRegexpForAutomaton("[a\\S]")
    assert RegexHelper.massageRegexpForAutomaton("[\\Sx]") == "[^ \b\f\n\r\tx]"


def test_regexhelper_random_vs_oracle():
    rng = random.Random(7)
    alpha = list("ab \t.()[]\\{}|*?+$^<>\"&dswDSWntrfbx0-") + ["\\d", "\\s", "\\w", "\\S", "[", "]", "\\n", "\\.", "\\\\"]
    for _ in range(3000):
        s = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 12)))
        assert RegexHelper.quoteLiteralAsRegexp(s) == O.quote_literal_as_regexp(s)
        assert RegexHelper.massageRegexpForJDK(s) == O.massage_regexp_for_jdk(s)
        try:
            exp = O.massage_regexp_for_automaton(s)
        except O.OracleError:
            with pytest.raises(ValueError):
                RegexHelper.massageRegexpForAutomaton(s)
            continue
        assert RegexHelper.massageRegexpForAutomaton(s) == exp


def test_multipattern_golden(golden):
    g = golden("multipattern")
    b = product_blob(g["patterns"], None)
    for c in g["cases"]:
        assert b.match(units_of(c["input"])) == c["match"]


def test_polymatch_golden(golden):
    for t in golden("polymatch")["tests"]:
        b, orc = both(t["extractions"])
        for c in t["cases"]:
            assert b.match(units_of(c["input"])) == c["match"]


def test_full_extraction_golden(golden):
    for t in golden("full_extraction")["tests"]:
        b, orc = both(t["extractions"])
        for c in t["cases"]:
            got = extract_both_ways(b, units_of(c["input"]))
            assert got == orc.extract(c["input"]), (t["name"], c["input"])
            assert got[0] >= 0
            if "id" in c:
                assert t["extractions"][got[0]]["name"] == c["id"]


def test_configs_golden(golden):
    g = golden("configs")
    for key in ("simple_grp", "readme_3"):
        b, orc = both(g[key]["extractions"])
        for c in g[key]["cases"]:
            u = units_of(c["input"])
            assert b.match(u) == c["match"]
            assert extract_both_ways(b, u) == orc.extract(c["input"])


def test_fused_automaton_is_built_for_the_benchmark_definitions(golden):
    g = golden("configs")
    for key in ("simple_grp", "readme_3"):
        b, _ = both(g[key]["extractions"])
        assert b.union_ok
        assert b.uni["n_regs"] <= 8  # value-shared registers: one per distinct live position


def test_blob_roundtrip(golden):
    g = golden("configs")["readme_3"]
    fl = [FlattenedExtraction(e["name"], e["pieces"]) for e in g["extractions"]]
    gorp = Gorp.construct(fl, host_only=True)
    blob = gorp.blob()
    again = Gorp.from_blob(blob, gorp.getExtractions(), host_only=True)
    assert bytes(again.blob()) == bytes(blob)
    assert gorp.max_groups == 4 and again.num_groups(2) == 4


def test_minimised_match_automaton_is_no_larger_than_reference(golden):
    # product states + 1 dead state <= reference product states + 1 (SURVEY Appendix B sizes)
    g = golden("configs")["readme_3"]
    b, orc = both(g["extractions"])
    assert b.m_states <= orc.num_states + 1
    assert b.ncls <= orc.num_points


def test_errors_are_loud():
    with pytest.raises(DefinitionParseException, match="Invalid regexp"):
        PolyMatcher.create("a)", host_only=True)
    with pytest.raises(DefinitionParseException, match="Invalid regexp"):
        PolyMatcher.create("(a", host_only=True)
    with pytest.raises(DefinitionParseException, match="Invalid regexp"):
        PolyMatcher.create('"abc', host_only=True)
    from gorp_amd.gorp import _create
    for rx in ["a\\b", "(?=a)a", "a++", "\\1", "[a&&b]", "\\p{L}", "^a", "a$"]:
        with pytest.raises(DefinitionParseException) as ei:
            _create(["a"], [rx], N.GX_CREATE_HOST_ONLY)
        assert ei.value.code == N.GX_E_UNSUPPORTED_CONSTRUCT, rx
    for rx in ["*a", "(a", "a)", "[a", "a{2,1}", "[b-a]", "a{"]:
        with pytest.raises(DefinitionParseException) as ei:
            _create(["a"], [rx], N.GX_CREATE_HOST_ONLY)
        assert ei.value.code == N.GX_E_REGEX_SYNTAX, rx


def test_no_device_means_error_not_fallback():
    """Without a GPU the extract entry points must fail; they never compute on the CPU."""
    if N.lib().gx_device_count() > 0:
        pytest.skip("a GPU is present")
    g = Gorp.construct([FlattenedExtraction("r", [["text", "a"]])], host_only=True)
    from gorp_amd.gorp import GorpError
    with pytest.raises(GorpError) as ei:
        g.extract("a")
    assert ei.value.code == N.GX_E_DEVICE
    with pytest.raises(GorpError) as ei:
        g.extract_batch(np.zeros(1, np.uint8), np.array([0, 1], np.uint32))
    assert ei.value.code == N.GX_E_DEVICE
    with pytest.raises(GorpError) as ei:
        Gorp.construct([FlattenedExtraction("r", [["text", "a"]])])
    assert ei.value.code == N.GX_E_DEVICE


# ---------------------------------------------------------------------------
# Randomised differential tests
# ---------------------------------------------------------------------------
ATOMS = ["a", "b", "c", "x", "\\d", "\\w", "\\S", "\\s", "[ \\t]", "[a-c]", "[^ab]", ".", "\\.", "=", ":", "\\[", "\\]",
         "\\D", "\\W", "[\\d.]", "[^\\s\"]", "\\\""]
QUANT = ["", "", "", "+", "*", "?", "{2}", "{1,3}", "{2,}"]
LAZY = ["+?", "*?", "??", "{1,2}?"]
ALPHA = "abcx019 =:.[]\"\t_-Z\r\x0b\x08\n"


def gen_pattern(rng, depth=0, lazy=False):
    """A Gorp *pattern* (pre-massage): both dialect strings are derived from it."""
    n = rng.randint(1, 3)
    parts = []
    q = QUANT + (LAZY if lazy else [])
    for _ in range(n):
        r = rng.random()
        if depth < 1 and r < 0.2:
            parts.append("(" + gen_pattern(rng, depth + 1, lazy) + ")" + rng.choice(q))
        elif depth < 1 and r < 0.3:
            parts.append("(" + gen_pattern(rng, depth + 1, lazy) + "|" + gen_pattern(rng, depth + 1, lazy) + ")")
        else:
            parts.append(rng.choice(ATOMS) + rng.choice(q))
    return "".join(parts)


def gen_pieces(rng, depth=0):
    out = []
    for _ in range(rng.randint(1, 3)):
        r = rng.random()
        if r < 0.35:
            out.append(["text", rng.choice(["a", "b ", " c", "x=", ": ", "[", "] ", ".", "  ", "a b", "\"", "GET", "k"])])
        elif r < 0.7 or depth >= 1:
            out.append(["pattern", gen_pattern(rng)])
        else:
            out.append(["extractor", "g%d" % rng.randint(0, 999), gen_pieces(rng, depth + 1)])
    return out


def gen_line(rng):
    return "".join(rng.choice(ALPHA) for _ in range(rng.randint(0, 12)))


def sample_from_match_automaton(b, rng, max_len=24):
    """A random line that the product's match automaton accepts (or nearly): a random
    walk over live transitions, using one representative byte per class."""
    reps = {}
    for c in range(256):
        reps.setdefault(int(b.cls256[c]), []).append(c)
    st, out = 0, []
    for _ in range(max_len):
        if b.m_accept_first[st] >= 0 and rng.random() < 0.3:
            break
        live = [c for c in range(b.ncls) if c in reps and b.m_next[st, c] != b.m_dead]
        if not live:
            break
        c = rng.choice(live)
        printable = [x for x in reps[c] if 32 <= x < 127]
        out.append(rng.choice(printable if printable and rng.random() < 0.9 else reps[c]))
        st = int(b.m_next[st, c])
    return bytes(out).decode("latin-1")


# What the library refuses although java.util.regex would compile it: the names gx_regex.cpp gives those constructs.  INTEGRATION.md
# section 3a is the record for a caller; test_refusal_record_is_what_the_library_refuses keeps the three in step.
REFUSED_CONSTRUCTS = (
    "a quantifier applied to a quantifier",
    "possessive quantifier",
    "a repeated capturing group that can match the empty string",
    "\\x{...}",
    "escape \\",
    "special group (?...)",
    "anchor '^'",
    "anchor '$'",
    "nested character class",
    "character class intersection",
)


def construct_both(make_product, make_oracle, tally):
    """The product and the oracle built SEPARATELY from one definition: (product, oracle), or None when either refuses.  A refusal
    by one side alone must not vanish (core/jdkre/JDKRegexpExtractionCooker.java:23 refuses nothing the JDK accepts): the product
    alone may refuse only with GX_E_UNSUPPORTED_CONSTRUCT naming a construct of the record, or GX_E_LIMIT; both kinds are counted."""
    p_err = o_err = None
    prod = orc = None
    try:
        prod = make_product()
    except ValueError:     # RegexHelper's own errors (IllegalArgumentException in the reference, which throws there too): before either side
        tally["front_end"] = tally.get("front_end", 0) + 1
        return None
    except GorpError as e:  # (DefinitionParseException is one: gx_create_* refused the regexps)
        p_err = e
    try:
        orc = make_oracle()
    except O.OracleError as e:
        o_err = e
    if p_err is None and o_err is None:
        return prod, orc
    if p_err is not None and o_err is not None:
        tally["both_refuse"] = tally.get("both_refuse", 0) + 1
        return None
    if p_err is not None:
        assert p_err.code in (N.GX_E_UNSUPPORTED_CONSTRUCT, N.GX_E_LIMIT), (p_err.code, str(p_err))
        if p_err.code == N.GX_E_UNSUPPORTED_CONSTRUCT:
            assert any(name in str(p_err) for name in REFUSED_CONSTRUCTS), str(p_err)
        tally["product_only"] = tally.get("product_only", 0) + 1
        tally.setdefault("product_only_messages", []).append(str(p_err))
    else:
        tally["oracle_only"] = tally.get("oracle_only", 0) + 1
        tally.setdefault("oracle_only_messages", []).append(str(o_err))
    return None


def test_random_definitions_match_and_extract():
    """Whole-path differential: random flattened extractions (1-4 per definition)."""
    rng = random.Random(1234)
    n_defs = n_lines = n_hits = n_exc = 0
    tally = {}
    while n_defs < 600:
        exts = [{"name": "e%d" % i, "pieces": gen_pieces(rng)} for i in range(rng.randint(1, 4))]
        fl = [FlattenedExtraction(e["name"], e["pieces"], e.get("append")) for e in exts]

        def regexes():
            built = [f.build() for f in fl]
            for e, b in zip(exts, built):     # the product's RegexHelper must agree with the oracle's restatement of it
                assert O.build_regex_strings(e["pieces"]) == b
            return [b[0] for b in built], [b[1] for b in built]

        pair = construct_both(lambda: product_blob(*regexes()), lambda: O.OracleGorp(*regexes()), tally)
        if pair is None:
            continue
        b, orc = pair
        n_defs += 1
        lines = [gen_line(rng) for _ in range(10)] + [sample_from_match_automaton(b, rng) for _ in range(20)]
        for ln in lines:
            u = units_of(ln)
            assert b.match(u) == orc.match(ln), (exts, ln)
            got, exp = extract_both_ways(b, u), orc.extract(ln)
            assert got == exp, (exts, ln, got, exp)
            n_lines += 1
            n_hits += got[0] >= 0
            n_exc += got[0] <= -2
    assert n_lines == 600 * 30
    assert n_hits > 3000
    # no definition of this generator's grammar is refused by one side alone (a silent refusal would show here)
    print("refusals:", {k: v for k, v in tally.items() if not k.endswith("_messages")})
    assert tally.get("product_only", 0) == 0, tally.get("product_only_messages")
    assert tally.get("oracle_only", 0) == 0, tally.get("oracle_only_messages")


def test_refusal_record_is_what_the_library_refuses():
    """The constructs the library refuses where java.util.regex compiles: (1) each of them IS refused, by name, with
    GX_E_UNSUPPORTED_CONSTRUCT (and by the oracle too, so the differentials never see them); (2) the names are exactly the strings
    gx_regex.cpp passes to unsupported(); (3) INTEGRATION.md section 3a lists exactly these names."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    samples = {
        "a quantifier applied to a quantifier": "x{2}{3}",
        "possessive quantifier": "a*+b",
        "a repeated capturing group that can match the empty string": "(a*)*b",
        "\\x{...}": "\\x{41}",
        "escape \\": "a\\bc",
        "special group (?...)": "(?=a)b",
        "anchor '^'": "^ab",
        "anchor '$'": "ab$",
        "nested character class": "[a[bc]]",
        "character class intersection": "[a-z&&b]",
    }
    assert set(samples) == set(REFUSED_CONSTRUCTS)
    for name, rx in samples.items():
        with pytest.raises(GorpError) as ei:
            product_blob(["ab"], [rx])
        assert ei.value.code == N.GX_E_UNSUPPORTED_CONSTRUCT and name in str(ei.value), (name, str(ei.value))
        if name.startswith("anchor"):
            # the ONE-SIDED refusal of the record: the oracle restates java.util.regex's anchors (no MULTILINE: '^' at the start of
            # the input, '$' at its end or before a final line terminator), the product refuses them -- a definition that uses one is
            # turned away by gx_create_*, loudly, and is outside every differential
            O.OracleGorp(["ab"], [rx])
        else:
            with pytest.raises(O.OracleError):
                O.OracleGorp(["ab"], [rx])
    src = open(os.path.join(root, "gorp_amd", "csrc", "gx_regex.cpp"), encoding="utf-8").read()
    in_source = set()
    for m in re.finditer(r'unsupported\((?:std::string\()?"((?:[^"\\]|\\.)*)"', src):
        in_source.add(m.group(1).encode("ascii").decode("unicode_escape"))
    in_source.discard("")
    assert in_source == set(REFUSED_CONSTRUCTS), in_source
    text = open(os.path.join(root, "INTEGRATION.md"), encoding="utf-8").read()
    block = text.split("<!-- refused-constructs:begin -->")[1].split("<!-- refused-constructs:end -->")[0]
    recorded = set(re.findall(r"^\s*\* `([^`]+)`", block, flags=re.M))
    assert recorded == set(REFUSED_CONSTRUCTS), recorded


def _sample_accepted(rng, rx, tries=40):
    """Random strings accepted by a JDK-dialect regex (found by rejection on the oracle)."""
    out = []
    for _ in range(tries):
        s = "".join(rng.choice("abcx019 =:.\"\t_") for _ in range(rng.randint(0, 9)))
        if O.jdk_matches(rx, s) is not None:
            out.append(s)
    return out


def test_random_raw_regex_pairs_capture_parity():
    """Capture automaton vs the backtracking restatement on raw JDK-dialect regexes,
    including lazy quantifiers and groups under quantifiers/alternations."""
    rng = random.Random(99)
    checked = matched = 0
    n = 0
    while n < 1500:
        pat = gen_pattern(rng, lazy=True)
        jdk = pat  # capturing groups kept as written
        autom = ".*"  # let every line through the matcher so the capture automaton decides
        try:
            b = product_blob([autom], [jdk])
            orc = O.OracleGorp([autom], [jdk])
        except (O.OracleError, DefinitionParseException):
            continue
        n += 1
        lines = [gen_line(rng) for _ in range(15)] + _sample_accepted(rng, jdk)
        for ln in lines:
            got, exp = extract_both_ways(b, units_of(ln)), orc.extract(ln)
            assert got == exp, (jdk, ln, got, exp)
            checked += 1
            matched += got[0] >= 0
    assert checked > 20000 and matched > 2000


def test_utf16_lines_and_high_classes():
    autom = ["[^a]+", "é+", "中.", "[Ā-࿿]x"]
    b = product_blob(autom, None)
    orc = O.OracleGorp(autom)
    for ln in ["éé", "中x", "Āx", "࿿x", "ကx", "a", "", "￿", "😀"]:
        assert b.match(units_of(ln)) == orc.match(ln), ln


def test_dialect_disagreement_yields_exception_code():
    for pieces, line, want in [
        ([["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]], "a\rb", -2),
        ([["text", "k"], ["pattern", "\\s"], ["text", "v"]], "k\x08v", -2),
        ([["text", "k"], ["pattern", "\\s"], ["text", "v"]], "k\x0bv", -1),
        ([["pattern", "[(]x"]], "(x", 0),
        ([["pattern", "[(]x"]], "?x", -1),
    ]:
        b, orc = both([{"name": "r", "pieces": pieces}])
        got = extract_both_ways(b, units_of(line))
        assert got == orc.extract(line)
        assert got[0] == want, (pieces, line)


def test_tile_image_tiers_by_definition_size():
    """Host-only handles run the table-image builders too (their self-check compares the record tier with the dense
    rows state by state): small definitions keep dense rows in LDS, the 64-extraction definition of BASELINE configs[2]
    keeps them in global memory with range records in LDS for match-only batches."""
    from gorp_amd import _native as N
    from gorp_amd import workloads as W
    g = Gorp.construct(W.readme3_definition(), host_only=True)
    assert g.stat(7) == 1 and g.stat(6) >= 10
    for flags, tier in ((N.GX_CREATE_TIER_RECORDS, 3), (N.GX_CREATE_TIER_L2, 2), (N.GX_CREATE_NO_TILES, 0),
                        (N.GX_CREATE_TIER_RECORDS | N.GX_CREATE_NO_FUSED, 3), (N.GX_CREATE_TIER_RECORDS_GLOBAL, 4)):
        assert Gorp.construct(W.readme3_definition(), host_only=True, flags=flags).stat(7) == tier
    rules, _ = W.syslog_definition(64, seed=3)
    g = Gorp.construct(rules, host_only=True)
    assert g.stat(7) == 3 and g.stat(9) == 3 and g.stat(10) == 16 and g.stat(11) == 16  # records in LDS, 16 waves of the lane kernel
    g = Gorp.construct(rules, host_only=True, flags=N.GX_CREATE_TIER_L2)
    assert g.stat(7) == 2 and g.stat(9) == 2       # forced: dense rows in global memory
    g = Gorp.construct(rules, host_only=True, flags=N.GX_CREATE_TIER_RECORDS)
    assert g.stat(7) == 3 and g.stat(6) >= 5       # records in LDS, at least 5 waves of staging left


def test_many_groups_that_may_be_empty():
    """CSV-like extractions: every field a group that may match the empty string.  The capture automaton's registers must not remember
    which fields were empty (one register per end of a group that can be set apart from the other end): 2^n states otherwise -- fourteen
    such fields once hit the state limit, twelve took seconds.  Compiles in no time, and agrees with the oracle on lines with every mix of
    empty and non-empty fields.  (Fields separated by blanks are another matter: a blank in a template is [ \\t]+, and which blank run separates
    which optional fields is a real ambiguity -- those automata do grow with 2^n.)"""
    import time
    rng = random.Random(5)
    for n_fields, sep, pat in ((32, ",", "[^,]*"), (20, ";", "[a-z\"\\\\]*"), (24, "|", "[^|]*")):
        pieces = []
        for k in range(n_fields):
            if sep == ";":
                pieces.append(["text", "%d=" % k])
            pieces.append(["extractor", "f%d" % k, [["pattern", pat]]])
            if k + 1 < n_fields or sep == ";":
                pieces.append(["text", sep])
        t0 = time.time()
        b, orc = both([{"name": "csv", "pieces": pieces}, {"name": "other", "pieces": [["text", "#"], ["extractor", "rest", [["pattern", ".*"]]]]}])
        assert time.time() - t0 < 5.0
        assert b.union_ok
        alphabet = "abxyz\"\\" if sep == ";" else "abc1 2\"" if sep == "," else "abc12, "
        for _ in range(300):
            fields = ["".join(rng.choice(alphabet) for _ in range(rng.choice([0, 0, 1, 2, 5]))) for _ in range(n_fields)]
            if sep == ";":
                line = "".join("%d=%s;" % (k, f) for k, f in enumerate(fields))
            else:
                line = sep.join(f.replace(sep, "") for f in fields)
            if rng.random() < 0.1:
                line = "#" + line
            if rng.random() < 0.1:
                line = line[:rng.randrange(len(line) + 1)]
            assert extract_both_ways(b, units_of(line)) == orc.extract(line), line


def _blank_separated_optional_fields(n_fields):
    pieces = []
    for k in range(n_fields):
        pieces.append(["extractor", "f%d" % k, [["pattern", "\\S*"]]])
        if k + 1 < n_fields:
            pieces.append(["text", " "])
    return [{"name": "fields", "pieces": pieces}, {"name": "other", "pieces": [["text", "#"], ["extractor", "rest", [["pattern", ".*"]]]]}]


def test_extractions_too_ambiguous_for_an_automaton_are_run_as_programs():
    """Fields that may be empty, separated by blanks ($a(\\S*) $b(\\S*) ...: a blank in a template is [ \\t]+, and which run of blanks
    separates which fields is a real ambiguity): the capture automaton grows with 2^n and java.util.regex simply backtracks
    (core/jdkre/JDKRegexpCookedExtraction.java:36-39).  Until round 4 fourteen such fields were refused (GX_E_LIMIT); now the
    extraction keeps its program and the kernels run it as it is -- thread lists in priority order (tests/blob_interp.py: pike_capture
    states the contract).  The blob carries the program (version 3) and survives a round trip."""
    rng = random.Random(14)
    for n_fields in (14, 20):
        b, orc = both(_blank_separated_optional_fields(n_fields))
        assert b.is_pike(0) and not b.is_pike(1) and not b.union_ok
        for _ in range(120):
            fields = ["".join(rng.choice("ab1") for _ in range(rng.choice([0, 1, 2, 4]))) for _ in range(n_fields)]
            line = " ".join(fields)
            if rng.random() < 0.3:     # more blanks than separators: runs of them, tabs
                line = line.replace(" ", rng.choice(["  ", " \t", "   "]), rng.randrange(1, 4))
            if rng.random() < 0.1:
                line = "#" + line
            if rng.random() < 0.15:    # too few separators: the match automaton says no (null)
                line = " ".join(fields[: n_fields // 2])
            assert b.extract(units_of(line)) == orc.extract(line), line
    # ten such fields still get their automaton (1 025 states, built in a second or two): nothing changes below the threshold
    b10, _ = both(_blank_separated_optional_fields(10))
    assert not b10.is_pike(0) and b10.union_ok
