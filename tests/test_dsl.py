"""The native definition-language front-end (gorp_amd/csrc/gx_dsl.cpp) replayed against the reference's own
parser tests.  Every case below is a vector transcribed from the cited JUnit test (definition text as input
data, asserted structure / error substring as expected output); test/ =
gorp-core/src/test/java/com/salesforce/gorp/.  verifyException in the reference is a case-insensitive
substring check (test/TestBase.java:12-23), mirrored by `raises_with`."""
import contextlib
import os

import pytest

from gorp_amd.gorp import DefinitionParseException, DefinitionReader, _definition_json


@contextlib.contextmanager
def raises_with(*substrings):
    with pytest.raises(DefinitionParseException) as ei:
        yield
    msg = ei.value.message.lower()
    assert any(s.lower() in msg for s in substrings), ei.value.message


def part(p):
    return (p["class"], p["text"])


# --- test/io/InputLineReaderTest.java:13-54, test/io/InputLineTest.java:8-22 --------------------------------
def test_input_line_reader_simple():
    text = "\n".join(["line 1", "line 2", "# commentary", "   ", "line 3\\", " with continuation \\", "or two...\\",
                      " or three!", "    # more comments"])
    lines = _definition_json(text, "<test>", "lines")["lines"]
    assert [ln["contents"] for ln in lines] == ["line 1", "line 2", "line 3 with continuation or two... or three!"]
    assert [ln["row"] for ln in lines] == [1, 2, 5] and lines[2]["rows"] == 4


def test_input_line_reader_fail():
    with raises_with("unexpected end-of-input when expecting line continuation") as _:
        _definition_json("line 1\nline 2\ncombo... \\", "<test>", "lines")
    with pytest.raises(DefinitionParseException) as ei:
        _definition_json("line 1\nline 2\ncombo... \\", "<test>", "lines")
    assert "row 3" in ei.value.message and "<test>" in ei.value.message


def test_input_line_row_col_across_continuations():
    # InputLineTest.testMultiLine: offsets past a join report the physical row and a 1-based column
    text = "#1\n#2\n#3\n#4\npattern %a x\\\nyy\\\nzz%\n"
    with pytest.raises(DefinitionParseException) as ei:
        DefinitionReader.reader(text).readUncooked()
    assert "[<input string> (7,4)]" in ei.value.message and "Orphan '%'" in ei.value.message


# --- test/UncookedDefTest.java ------------------------------------------------------------------------------
def test_uncooked_simple():  # :13-55
    d = DefinitionReader.reader(
        "pattern %ws \\s+\n"
        "pattern %optws \\s*\n"
        "pattern %'phrase' \\S+\n"
        "pattern %\"maybeUUID\" %'phrase'\n"
        "# hyphen not valid, must be quoted:\n"
        "pattern %'host-name' %\"phrase\"\n"
        "\n"
        "template @simple Prefix:\n"
        "template @'base' %phrase%optws(sic!) @simple %'host-name'\n"
        "\n"
        "extract FooMessage {  \n"
        "  template @base ($authStatus(Accepted))\n"
        "  append \"service\":\"ssh\", \"logType\":\"security\"  \n"
        "}\n").readUncooked()
    assert list(d["patterns"]) == ["ws", "optws", "phrase", "maybeUUID", "host-name"]
    assert list(d["templates"]) == ["simple", "base"]
    assert list(d["extractions"]) == ["FooMessage"]
    assert d["extractions"]["FooMessage"]["append"] == {"service": "ssh", "logType": "security"}


def test_uncooked_pattern_refs_in_patterns():  # :57-97
    d = DefinitionReader.reader(
        "pattern %wsChar \\s\n"
        "pattern %optws %wsChar*%%\n"
        "pattern %word ([a-z]+)\n"
        "pattern %phrase3   %word %word2%word3\n").readUncooked()
    assert len(d["patterns"]) == 4
    assert [part(p) for p in d["patterns"]["optws"]] == [("PatternReference", "wsChar"), ("LiteralPattern", "*%")]
    assert [part(p) for p in d["patterns"]["phrase3"]] == [
        ("PatternReference", "word"), ("LiteralPattern", " "), ("PatternReference", "word2"), ("PatternReference", "word3")]


def test_uncooked_template_refs():  # :99-141
    d = DefinitionReader.reader(
        "pattern %wsChar \\s\n"
        "\n"
        "template @base Stuff:\n"
        "template @actual @'base'%'wsChar'and%{\\s}more\n").readUncooked()
    assert [part(p) for p in d["templates"]["base"]["parts"]] == [("LiteralText", "Stuff:")]
    assert [part(p) for p in d["templates"]["actual"]["parts"]] == [
        ("TemplateReference", "base"), ("PatternReference", "wsChar"), ("LiteralText", "and"), ("LiteralPattern", "\\s"),
        ("LiteralText", "more")]


def test_uncooked_extractors():  # :143-164
    d = DefinitionReader.reader("template @actual value=$value(Accepted$$%{\\d+})\n").readUncooked()
    parts = d["templates"]["actual"]["parts"]
    assert part(parts[0]) == ("LiteralText", "value=") and part(parts[1]) == ("ExtractorExpression", "value")
    assert [part(p) for p in parts[1]["parts"]] == [("LiteralText", "Accepted$"), ("LiteralPattern", "\\d+")]


def test_uncooked_extractors2():  # :166-210
    d = DefinitionReader.reader(
        "pattern %w [a-zA-Z]+\n"
        "template @base value=$value(%w)\n"
        "template @full @base extra=$extra($prop1(%w),$prop2(%w))\n").readUncooked()
    base = d["templates"]["base"]["parts"]
    assert part(base[0]) == ("LiteralText", "value=") and part(base[1]) == ("ExtractorExpression", "value")
    assert [part(p) for p in base[1]["parts"]] == [("PatternReference", "w")]
    full = d["templates"]["full"]["parts"]
    assert [part(p) for p in full] == [("TemplateReference", "base"), ("LiteralText", " extra="), ("ExtractorExpression", "extra")]
    assert [part(p) for p in full[2]["parts"]] == [("ExtractorExpression", "prop1"), ("LiteralText", ","), ("ExtractorExpression", "prop2")]


def test_uncooked_failures():  # :214-243
    with raises_with("duplicate"):
        DefinitionReader.reader("pattern %'ws' \\s+\npattern %optws \\s*\npattern %ws \\S+\n").readUncooked()
    with raises_with("Orphan '%'"):
        DefinitionReader.reader("pattern %'ws' \\s+%\n").readUncooked()


# --- test/PatternResolutionTest.java ------------------------------------------------------------------------
def test_pattern_resolution():  # :12-38
    d = DefinitionReader.reader(
        "pattern %a a\n"
        "pattern %b b\n"
        "pattern %c stuff!\n"
        "pattern %abba (%a%b %'b'-%a)\n"
        "pattern %full %abba %c\n").resolveTemplates()
    assert d["patterns"] == {"a": "a", "b": "b", "c": "stuff!", "abba": "(ab b-a)", "full": "(ab b-a) stuff!"}


def test_pattern_resolution_failures():  # :40-70
    with raises_with("non-existing pattern '%c'"):
        DefinitionReader.reader("pattern %a Ok: %b\npattern %b But... %c\n").resolveTemplates()
    with raises_with("cyclic pattern reference to '%a'"):
        DefinitionReader.reader("pattern %a Kaboom: %a\n").resolveTemplates()
    with raises_with("cyclic pattern reference to '%a'"):
        DefinitionReader.reader("pattern %a %b\npattern %b %a").resolveTemplates()


# --- test/TemplateResolutionTest.java -----------------------------------------------------------------------
def test_template_resolution_simplest():  # :13-44
    d = DefinitionReader.reader("template @base (%{a}:foo)\ntemplate @full @base...\n").resolveTemplates()
    assert d["patterns"] == {}
    assert [part(p) for p in d["templates"]["base"]] == [("LiteralText", "("), ("LiteralPattern", "a"), ("LiteralText", ":foo)")]
    assert [part(p) for p in d["templates"]["full"]] == [("LiteralText", "("), ("LiteralPattern", "a"), ("LiteralText", ":foo)"),
                                                         ("LiteralText", "...")]


def test_template_resolution_simple():  # :46-91
    d = DefinitionReader.reader(
        "pattern %a a\n"
        "template @base (%a:foo)\n"
        "template @full @base...%{[.*{2}]}--%a\n").resolveTemplates()
    assert d["patterns"] == {"a": "a"}
    assert [part(p) for p in d["templates"]["full"]] == [
        ("LiteralText", "("), ("LiteralPattern", "a"), ("LiteralText", ":foo)"), ("LiteralText", "..."),
        ("LiteralPattern", "[.*{2}]"), ("LiteralText", "--"), ("LiteralPattern", "a")]


def test_template_resolution_with_extractors():  # :93-142
    d = DefinitionReader.reader(
        "pattern %w [a-zA-Z]+\n"
        "template @base value=$value(%w)\n"
        "template @full @base extra=$extra($prop1(%w),$prop2(%w))\n").resolveTemplates()
    assert [part(p) for p in d["templates"]["base"]] == [("LiteralText", "value="), ("ExtractorExpression", "value")]
    full = d["templates"]["full"]
    assert [part(p) for p in full] == [("LiteralText", "value="), ("ExtractorExpression", "value"), ("LiteralText", " extra="),
                                       ("ExtractorExpression", "extra")]
    assert [part(p) for p in full[3]["parts"]] == [("ExtractorExpression", "prop1"), ("LiteralText", ","), ("ExtractorExpression", "prop2")]


# --- test/ExtractionResolutionTest.java ---------------------------------------------------------------------
def test_extraction_resolution_append():  # :12-41
    g = DefinitionReader.reader(
        "pattern %a a\n"
        "template @base (%a:foo)\n"
        "extract rule1 {  \n"
        "  template @base value=$MyValue(%a:%{\\w+})\n"
        "  append { \"enabled\" : true, \"x\" : 3 }\n"
        "}").read(host_only=True)
    extras = g.getExtractions()
    assert len(extras) == 1
    appends = extras[0].getExtra()
    assert appends == {"enabled": True, "x": 3} and appends["enabled"] is True and isinstance(appends["x"], int)
    # the commented-out structure check of the reference test (:43-63), on the flattened pieces
    fl, _ = DefinitionReader.reader(
        "pattern %a a\ntemplate @base (%a:foo)\nextract rule1 {  \n  template @base value=$MyValue(%a:%{\\w+})\n}").flatten()
    assert fl[0].pieces == [["text", "("], ["pattern", "a"], ["text", ":foo)"], ["text", " value="],
                            ["extractor", "MyValue", [["pattern", "a"], ["text", ":"], ["pattern", "\\w+"]]]]


def test_extraction_resolution_failures():  # :68-96
    with raises_with("No extraction definitions found"):
        DefinitionReader.reader("pattern %a a\ntemplate @base (%a:foo)\n").read(host_only=True)
    with raises_with("Duplicate extractor name"):
        DefinitionReader.reader("pattern %word \\w+\ntemplate @extr $value(%word)\nextract match {  \n  template @extr @extr\n}\n").read(host_only=True)


# --- test/ParametricTemplateTest.java:42-155, test/ParametricExtractorTest.java:42-64 -----------------------
def test_parametric_errors():
    with raises_with("Missing parameter list") as _:
        DefinitionReader.reader(
            "pattern %word ([a-zA-Z]+)\ntemplate @pair() @1:@2\ntemplate @full @pair\nextract Result {  \n  template @full\n}\n").read(host_only=True)
    with raises_with("@pair"):
        DefinitionReader.reader(
            "pattern %word ([a-zA-Z]+)\ntemplate @pair() @1:@2\ntemplate @full @pair\nextract Result {  \n  template @full\n}\n").read(host_only=True)
    with raises_with("Invalid variable reference"):
        DefinitionReader.reader("template @pair @1:@2\ntemplate @full xyz\nextract Result {  \n  template @full\n}\n").read(host_only=True)
    with raises_with("Unexpected end of line"):
        DefinitionReader.reader(
            "template @pair() @1:@2\ntemplate @a    a\ntemplate @full @pair(@a\nextract Result {  \n  template @full\n}\n").read(host_only=True)
    with raises_with("non-existing template '@ab'"):
        DefinitionReader.reader(
            "template @constant text\ntemplate @abc @full(@ab(@c,@1))\ntemplate @full() @1\nextract Result {  \n  template @full\n}\n").read(host_only=True)
    for args in ("@foo", "@foo,@foo,@foo"):
        with raises_with("Parameter mismatch"):
            DefinitionReader.reader(
                "template @pair() @1:@2\ntemplate @foo foosball\ntemplate @fooPair @pair(%s)\nextract Result {  \n  template @fooPair\n}\n"
                % args).read(host_only=True)
    with raises_with("duplicate extractor name") as _:
        DefinitionReader.reader(
            "pattern %num ([0-9]+)\npattern %word ([a-zA-Z]+)\npattern %ip [a-zA-Z\\.]+\ntemplate @ip %ip\ntemplate @port %num\n"
            "template @endpoint() $1(@ip):$2(@port)\nextract Net {  \n"
            "  template @endpoint($srcIp,$srcPort)/%word @endpoint($srcIp,$whatever)\n}\n").read(host_only=True)


# --- the definitions of the extraction tests, through the DSL: pieces must equal the hand-flattened fixtures -
def test_dsl_reproduces_fixture_pieces(golden):
    for name in ("polymatch", "full_extraction"):
        for t in golden(name)["tests"]:
            fl, d = DefinitionReader.reader(t["def"]).flatten()
            assert [f.name for f in fl] == [e["name"] for e in t["extractions"]], t["name"]
            assert [f.pieces for f in fl] == [e["pieces"] for e in t["extractions"]], t["name"]
            for f, e in zip(fl, t["extractions"]):
                assert (f.append or None) == (e.get("append") or None)
            # and the strings the C++ side builds are the ones the Python-side builder derives from the pieces
            for f, x in zip(fl, d["extractions"]):
                a, j, names = f.build()
                assert (a, j, names) == (x["automaton_rx"], x["jdk_rx"], x["extractor_names"])


def test_sample_file_and_readme_definition(golden):
    """samples/simple.grp and the README multi-matcher definition (README.md:114-135), as text."""
    simple = (
        "### First, let's define basic patterns using \"patterns\" (regexps)\n\n"
        "# inline whitespace is understood, but for more explicit usage may also define:\n"
        "pattern %ws \\s+\npattern %optws \\s*\n"
        "# 'phrase' means non-space-sequence of characters; 'word' letters; 'num' digits\n"
        "pattern %word \\w+\npattern %phrase \\S+\npattern %num \\d+\n"
        "# more semantic macros, loosely defined\npattern %ts %phrase\npattern %ip %phrase\n"
        "# may need basic \"rest of content\" matcher too\npattern %any .*\n\n"
        "template @base <%num>$eventTimeStamp(%ts)\n\n"
        "extract sampleMatch {\n  template @base ($authStatus(Accepted)) \n}\n")
    fl, d = DefinitionReader.reader(simple).flatten()
    assert [f.pieces for f in fl] == [e["pieces"] for e in golden("configs")["simple_grp"]["extractions"]]
    assert d["extractions"][0]["jdk_rx"].endswith("[ \t]+")  # the trailing blank of the template line is significant
    readme = (
        "pattern %num \\d+\npattern %word \\w+\npattern %phrase \\S+\n\n"
        "extract PutRequest {\n   # comment inside a block\n"
        "   template [$timestamp(%num)]: $verb(PUT) $timeTakenInMsec(%num)ms\\\n $path(%phrase)\n"
        "   append { \"marker\" : \"EXTRACTED\" }\n}\n"
        "extract GetRequest {\n   template [$timestamp(%num)]: $verb(GET) $timeTakenInMsec(%num)ms\\\n $path(%phrase)\n"
        "   append { \"marker\" : \"EXTRACTED\" }\n}\n"
        "extract OtherRequest {\n   template [$timestamp(%num)]: $verb(%word) $timeTakenInMsec(%num)ms\\\n $path(%phrase)\n"
        "   append { \"marker\" : \"EXTRACTED\" }\n}\n")
    fl, _ = DefinitionReader.reader(readme).flatten()
    want = golden("configs")["readme_3"]["extractions"]
    assert [(f.name, f.pieces, f.append) for f in fl] == [(e["name"], e["pieces"], e["append"]) for e in want]


def test_reader_accepts_path(tmp_path):
    p = tmp_path / "d.grp"
    p.write_text("extract a {\n template x=$v(%{\\d+})\n}\n")
    fl, _ = DefinitionReader.reader(p).flatten()
    assert fl[0].pieces == [["text", "x="], ["extractor", "v", [["pattern", "\\d+"]]]]
    with pytest.raises(DefinitionParseException) as ei:
        (tmp_path / "bad.grp").write_text("bogus line\n")
        DefinitionReader.reader(tmp_path / "bad.grp").flatten()
    assert "Unrecognized keyword \"bogus\"" in ei.value.message and os.path.basename(str(tmp_path / "bad.grp")) in ei.value.message


def test_quirks_preserved():
    # duplicate extraction name silently replaces the earlier one in its original slot
    fl, _ = DefinitionReader.reader(
        "extract a {\n template one\n}\nextract b {\n template two\n}\nextract a {\n template three\n}\n").flatten()
    assert [(f.name, f.pieces) for f in fl] == [("a", [["text", "three"]]), ("b", [["text", "two"]])]
    # a template referenced BEFORE its declaration resolves to nothing (CookedDefinitions.java:233-236 resolves
    # the freshly constructed, still empty, template)
    fl, _ = DefinitionReader.reader("template @first [@second]\ntemplate @second inner\nextract x {\n template @first @second\n}\n").flatten()
    assert fl[0].pieces == [["text", "["], ["text", "]"], ["text", " "]]
    # doubled sigils are literals; inline patterns may nest braces
    fl, _ = DefinitionReader.reader("extract x {\n template 100%% @@home $$5 %{a{2}}\n}\n").flatten()
    assert fl[0].pieces == [["text", "100% @home $5 "], ["pattern", "a{2}"]]


def test_gx_create_from_definition_matches_the_two_step_path():
    import ctypes as C

    import numpy as np

    from gorp_amd import _native as N
    text = "pattern %num \\d+\nextract a {\n template id=$id(%num) $rest(%{.*})\n append \"k\": 1\n}\nextract b {\n template $all(%{.+})\n}\n"
    L = N.lib()
    h = C.c_void_p()
    assert L.gx_create_from_definition(text.encode(), None, N.GX_CREATE_HOST_ONLY, C.byref(h)) == 0, N.last_error()
    n = L.gx_blob_size(h)
    blob = np.zeros(n, np.uint8)
    assert L.gx_blob_copy(h, blob.ctypes.data, n) == 0
    assert L.gx_num_extractions(h) == 2 and L.gx_num_groups(h, 0) == 2 and L.gx_num_groups(h, 1) == 1
    L.gx_destroy(h)
    g = DefinitionReader.reader(text).read(host_only=True)
    assert bytes(g.blob()) == bytes(blob)
    # errors come back as GX_E_DEFINITION with the reference's message shape
    assert L.gx_create_from_definition(b"extract a {\n template %nope\n}\n", b"unit", N.GX_CREATE_HOST_ONLY, C.byref(h)) == N.GX_E_DEFINITION
    assert "non-existing pattern '%nope'" in N.last_error() and "[unit (2," in N.last_error()
    # an invalid regex inside a pattern surfaces like Gorp.construct's wrapper (core/Gorp.java:84-90)
    rc = L.gx_create_from_definition(b"extract a {\n template %{(a}\n}\n", None, N.GX_CREATE_HOST_ONLY, C.byref(h))
    assert rc == N.GX_E_REGEX_SYNTAX and "problem with PolyMatcher construction" in N.last_error()


def test_extraction_cooker_plugin_api():
    """core/ExtractionCooker.java:16-30 mirrored: Gorp.construct(defs, cooker) drives the cooker's append* methods
    exactly as Gorp._buildExtractor does (core/Gorp.java:94-129) and keeps what cook() returns."""
    from gorp_amd.gorp import ExtractionCooker, HipExtractionCooker, FlattenedExtraction, Gorp

    calls = []

    class Recording(HipExtractionCooker):
        def appendPattern(self, pattern, buffer):
            calls.append(("pattern", pattern))
            super().appendPattern(pattern, buffer)

        def appendLiteral(self, literal, buffer):
            calls.append(("literal", literal))
            super().appendLiteral(literal, buffer)

        def appendStartExpression(self, buffer):
            calls.append(("start",))
            super().appendStartExpression(buffer)

        def appendFinishExpression(self, buffer):
            calls.append(("finish",))
            super().appendFinishExpression(buffer)

        def cook(self, index, regexpSource, extr):
            calls.append(("cook", index, regexpSource))
            return super().cook(index, regexpSource, extr)

    digit = chr(92) + "d"                      # the pattern \\d, spelled without escapes in this file
    ext = FlattenedExtraction("x", [["text", "k= "], ["extractor", "v", [["pattern", "(a|b)+"], ["extractor", "w", [["pattern", digit]]]]]], {"t": 1})
    g = Gorp.construct([ext], Recording(), host_only=True)
    want = "k=[ " + chr(9) + "]+((?:a|b)+(" + digit + "))"   # quoteLiteralAsRegexp: blank run -> [ TAB]+
    assert calls == [("literal", "k= "), ("start",), ("pattern", "(a|b)+"), ("start",), ("pattern", digit), ("finish",), ("finish",),
                     ("cook", 0, want)]
    x = g.getExtractions()[0]
    assert x.getName() == "x" and x._extractorNames == ["v", "w"] and x.getExtra() == {"t": 1}
    assert x.getRegexpSource() == want
    with pytest.raises(NotImplementedError):
        ExtractionCooker().cook(0, "", ext)
