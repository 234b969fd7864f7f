"""Pins the parity oracle (oracle/) against the reference's own known-answer
tests, transcribed under tests/golden/ (SURVEY.md Appendix B)."""
import pytest

from oracle import oracle as O


def _build(extractions):
    rs = [O.build_regex_strings(e["pieces"]) for e in extractions]
    return O.OracleGorp([r[0] for r in rs], [r[1] for r in rs]), [r[2] for r in rs]


def _as_map(extractions, names, k, caps, line, id_as):
    """ExtractionResult.asMap (core/ExtractionResult.java:65-88)."""
    m = {}
    if id_as is not None:
        m[id_as] = extractions[k]["name"]
    for nm, c in zip(names[k], caps):
        m[nm] = None if c is None else line[c[0]:c[1]]
    m.update(extractions[k].get("append") or {})
    return m


def test_regexhelper(golden):
    g = golden("regexhelper")
    for src, exp in g["quoteLiteralAsRegexp"]:
        assert O.quote_literal_as_regexp(src) == exp
    for src, exp in g["massageRegexpForAutomaton"]:
        assert O.massage_regexp_for_automaton(src) == exp
    for src, exp in g["massageRegexpForJDK"]:
        assert O.massage_regexp_for_jdk(src) == exp


def test_regexhelper_errors():
    with pytest.raises(O.OracleError, match="Unrecognized backslash escape"):
        O.massage_regexp_for_automaton("\\q")
    with pytest.raises(O.OracleError, match="negated character class"):
        O.massage_regexp_for_automaton("[a\\S]")
    # negated class directly after '[' is legal (RegexHelper.java:124,193-197)
    assert O.massage_regexp_for_automaton("[\\Sx]") == "[^ \b\f\n\r\tx]"


def test_multipattern(golden):
    g = golden("multipattern")
    m = O.OracleGorp(g["patterns"])
    for c in g["cases"]:
        assert m.match(c["input"]) == c["match"], c
        assert m.match(c["input"].encode("latin-1")) == c["match"], c
    # SURVEY.md B.1 [scratch] sizes, re-derived here
    assert (m.num_states, m.num_points) == (12, 10)
    assert m.component_states() == [3, 4, 4, 2, 2, 4]
    assert m.points().tolist() == [0] + [ord(c) for c in "abcdefgvw"]


def test_polymatch(golden):
    for t in golden("polymatch")["tests"]:
        gorp, _ = _build(t["extractions"])
        for c in t["cases"]:
            assert gorp.match(c["input"]) == c["match"], (t["name"], c)


def test_polymatch_sizes(golden):
    # SURVEY.md B.2 [scratch] product sizes re-derived by the oracle
    sizes = {}
    for t in golden("polymatch")["tests"]:
        gorp, _ = _build(t["extractions"])
        sizes[t["name"]] = (gorp.num_states, gorp.num_points)
    assert sizes == {"testSimple": (26, 25), "testIntermediate": (34, 47), "testQuoted": (17, 30),
                     "testComplex": (74, 54)}


def test_full_extraction(golden):
    for t in golden("full_extraction")["tests"]:
        gorp, names = _build(t["extractions"])
        for c in t["cases"]:
            k, caps = gorp.extract(c["input"])
            assert k >= 0, (t["name"], c)
            if c.get("not_null"):
                continue
            assert t["extractions"][k]["name"] == c["id"]
            m = _as_map(t["extractions"], names, k, caps, c["input"], c.get("id_as"))
            for key, val in c["map"].items():
                assert m[key] == val, (t["name"], key, m)
            if "map_size" in c:
                assert len(m) == c["map_size"]


def test_full_intermediate_all_captures(golden):
    t = [x for x in golden("full_extraction")["tests"] if x["name"] == "testIntermediate"][0]
    gorp, names = _build(t["extractions"])
    line = t["cases"][0]["input"]
    k, caps = gorp.extract(line)
    got = {n: line[c[0]:c[1]] for n, c in zip(names[k], caps)}
    assert got == {"eventTimeStamp": "2015-05-12T20:57:53.302858+00:00", "logAgent": "10.1.11.141",
                   "logSrcIp": "10.10.5.3"}


def test_configs(golden):
    g = golden("configs")
    for key, states, points, comps in [("simple_grp", 17, 28, None), ("readme_3", 31, 34, [15, 15, 13])]:
        cfg = g[key]
        gorp, names = _build(cfg["extractions"])
        assert (gorp.num_states, gorp.num_points) == (states, points)
        if comps:
            assert gorp.component_states() == comps
        for c in cfg["cases"]:
            assert gorp.match(c["input"]) == c["match"], c
            k, caps = gorp.extract(c["input"])
            if not c["match"]:
                assert k == -1
                continue
            assert k == c["match"][0]
            assert cfg["extractions"][k]["name"] == c["id"]
            if "map" in c:
                got = {n: c["input"][b:e] for n, (b, e) in zip(names[k], caps)}
                assert got == c["map"]


def test_transition_table_contract():
    """Automata.step/accept (core/autom/Automata.java:133-139): walking the exported
    int32 table by hand gives the same answer as match()."""
    m = O.OracleGorp(["ab+", "abc+", "ab?c", "v", "v.*", "(def)+"])
    tr, pts = m.transitions(), m.points().tolist()
    assert tr.shape == (12, 10)
    assert tr[0].tolist().count(-1) == 10 - 3  # only 'a', 'd', 'v' leave the start state

    def cls(ch):
        i = 0
        while i + 1 < len(pts) and pts[i + 1] <= ch:
            i += 1
        return i
    for s in ["ab", "abc", "ac", "v", "vzzz", "defdef", "defde", "x"]:
        p = 0
        for ch in s:
            p = tr[p, cls(ord(ch))]
            if p == -1:
                break
        assert (m.accept(p) if p >= 0 else []) == m.match(s)


def test_extraction_exception_and_null():
    """core/Gorp.java:162-164 (null) and :173-177 (DFA says yes, regex says no).

    '.' is any char for the automaton but excludes \\r for java.util.regex
    (SURVEY.md Appendix A.3), so 'a\\rb' trips the exception path."""
    a, j, _ = O.build_regex_strings([["text", "a"], ["extractor", "x", [["pattern", ".*"]]], ["text", "b"]])
    g = O.OracleGorp([a], [j])
    assert g.extract("a--b") == (0, [(1, 3)])
    assert g.extract("zzz") == (-1, [])
    assert g.match("a\rb") == [0]
    assert g.extract("a\rb") == (-2, [])
    # \s differs too: BS(0x08) vs VT(0x0B)
    a, j, _ = O.build_regex_strings([["text", "k"], ["pattern", "\\s"], ["text", "v"]])
    g = O.OracleGorp([a], [j])
    assert g.extract("k\x08v")[0] == -2
    assert g.extract("k\x0bv")[0] == -1
    assert g.extract("k v")[0] == 0


def test_non_ascii_utf16_walk():
    g = O.OracleGorp(["[^a]+", "é+", "中."])
    assert g.match("éé") == [0, 1]
    assert g.match("中x") == [0, 2]
    assert g.match("éé".encode("latin-1")) == [0, 1]


def test_brics_quirks():
    # '"' always opens a quoted literal; '(' ')' is the empty string; '|' at atom position is a literal
    g = O.OracleGorp(['"a+b"c', "x()y", "(|a)", "a{2,3}", "a{2,}", "[a-]+", "[^a-y]"])
    assert g.match("a+bc") == [0]
    assert g.match("xy") == [1]
    assert g.match("|a") == [2]
    assert g.match("aa") == [3, 4, 5]
    assert g.match("aaaa") == [4, 5]
    assert g.match("a-a") == [5]
    assert g.match("z") == [6]
    with pytest.raises(O.OracleError, match="Invalid regexp"):
        O.OracleGorp(["a)"])
    with pytest.raises(O.OracleError, match="Invalid regexp"):
        O.OracleGorp(["(a"])


def test_minimisation_regression_all_accepting_states():
    """Found by the compiler-vs-oracle differential test: with no live non-accepting
    state the refinement loop must not stop after one round."""
    g = O.OracleGorp(["(\\[[^ ]?)*([^0-9]*)"])
    assert g.match("__[9\t") == []
    assert g.match("[9x") == [0]
    assert g.match("_x") == [0]
    assert g.match("[9[8__") == [0]
    assert g.match("[98") == []


def test_read_lines_is_buffered_reader_readline():
    """oracle.read_lines (the ingestion checker): the three readLine() terminators, no line after a final one."""
    off, lines, flags = O.read_lines(b"a\r\nb\rc\n\nd")
    assert lines == [b"a", b"b", b"c", b"", b"d"]
    assert off.tolist() == [0, 3, 5, 7, 8, 9]
    assert flags.tolist() == [0, 0, 0, 0, 0]
    off, lines, flags = O.read_lines(b"\xe9\n\r")
    assert lines == [b"\xe9", b""] and off.tolist() == [0, 2, 3] and flags.tolist() == [1, 0]
    assert O.read_lines(b"")[1] == [] and O.read_lines(b"\n")[1] == [b""]
    # a vertical tab / form feed / NEL are ordinary bytes of a line, as for readLine()
    assert O.read_lines(b"a\x0bb\x0cc\x85d")[1] == [b"a\x0bb\x0cc\x85d"]


def test_results_to_jsonl_is_asmap_serialised():
    """oracle.results_to_jsonl (the materialisation checker): ExtractionResult.asMap(idAs) as a LinkedHashMap -- id,
    captures in group order (null for an unset group), append entries; a key put again keeps its place."""
    import numpy as np
    names = ["first", "second"]
    extractor_names = [["a", "b"], ["v", "v"]]
    appends = [{"b": 7, "extra": {"k": [1, None]}}, None]
    lines = [b'x"y' + bytes([0x5C]) + b"z", b"plain", b"nothing"]
    mid = np.array([0, 1, -1], np.int32)
    caps = np.array([[0, 3, -1, -1], [0, 2, 2, 5], [-1, -1, -1, -1]], np.int32)
    text, offs = O.results_to_jsonl(lines, mid, caps, names, extractor_names, appends, id_as="id")
    bs = bytes([0x5C])  # one backslash
    line0 = b'{"id":"first","a":"x' + bs + b'"y","b":7,"extra":{"k":[1,null]}}' + b"\n"
    line1 = b'{"id":"second","v":"ain"}' + b"\n"
    assert text == line0 + line1
    assert offs.tolist() == [0, len(line0), len(line0) + len(line1), len(line0) + len(line1)]
    t2, _ = O.results_to_jsonl([bytes([1, 9, 0xE9])], np.array([0], np.int32), np.array([[0, 3, -1, -1]], np.int32), names,
                               extractor_names, [None, None])
    assert t2 == b'{"a":"' + bs + b"u0001" + bs + b"t" + bytes([0xC3, 0xA9]) + b'","b":null}' + b"\n"
