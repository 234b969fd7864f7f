#!/usr/bin/env python3
"""Writes tests/golden/*.json.

These are NOT outputs of running the reference (no JVM exists in the build
image).  They are the known-answer vectors that the reference's own JUnit tests
assert, transcribed as data: definition text, input line, expected result.
Each case cites the reference test it was transcribed from
(test/ = gorp-core/src/test/java/com/salesforce/gorp/).

`pieces` is the hand-flattened extraction (what CookedDefinitions.resolveExtractions
hands to Gorp.construct, core/Gorp.java:58-79): a list of
    ["text", s] | ["pattern", s] | ["extractor", name, [pieces]]
`def` is the DSL text of the reference test, kept as input data for the DSL
front-end.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

TAB = "\t"


def T(s):
    return ["text", s]


def P(s):
    return ["pattern", s]


def X(name, *kids):
    return ["extractor", name, list(kids)]


# ---------------------------------------------------------------------------
# B.1 raw multi-pattern DFA: test/autom/MultiPatternTest.java:10-28
# ---------------------------------------------------------------------------
multipattern = {
    "source": "test/autom/MultiPatternTest.java:12-27",
    "patterns": ["ab+", "abc+", "ab?c", "v", "v.*", "(def)+"],
    "cases": [
        {"input": "ab", "match": [0]},
        {"input": "abc", "match": [1, 2]},
        {"input": "ac", "match": [2]},
        {"input": "", "match": []},
        {"input": "v", "match": [3, 4]},
        {"input": "defdef", "match": [5]},
        {"input": "defde", "match": []},
        {"input": "abbbbb", "match": [0]},
    ],
}

# ---------------------------------------------------------------------------
# B.2 DSL -> PolyMatcher.match: test/PolyMatchTest.java
# ---------------------------------------------------------------------------
W = P("(\\w+)")
PH_TAB = P("[^ " + TAB + "]+")
NUMP = P("([0-9]+)")
WORDAZ = P("[a-zA-Z]+")

complex_base = [
    T("<"), NUMP, T(">"), X("eventTimeStamp", PH_TAB), T(" "), X("logAgent", PH_TAB),
    T(' RealSource: "'), X("logSrcIp", PH_TAB),
    T('" Environment: "'), X("environment", PH_TAB),
    T('" UUID: "'), X("uuid", PH_TAB),
    T('" RawMsg: <'), NUMP, T(">"),
    X("rawMsgTS", WORDAZ, T(" "), NUMP, T(" "), PH_TAB),
    T(" "), X("logSrcHostname", PH_TAB), T(" "), X("appname", WORDAZ), T("["), X("appPID", NUMP), T("]"),
]

polymatch = {
    "source": "test/PolyMatchTest.java",
    "tests": [
        {
            "name": "testSimple", "source": "test/PolyMatchTest.java:13-36",
            "def": "pattern %word (\\w+)\n"
                   "template @base %word\n"
                   "extract rule1 {  \n"
                   "  template @base value=$value(%word) value2=$value2(%word)\n"
                   "}\n"
                   "extract rule2 {  \n"
                   "  template value=%word\n"
                   "}\n",
            "extractions": [
                {"name": "rule1", "pieces": [W, T(" value="), X("value", W), T(" value2="), X("value2", W)]},
                {"name": "rule2", "pieces": [T("value="), W]},
            ],
            "cases": [
                {"input": "value=stuff", "match": [1]},
                {"input": "prefix value=a value2=b", "match": [0]},
            ],
        },
        {
            "name": "testIntermediate", "source": "test/PolyMatchTest.java:38-57",
            "def": "pattern %phrase \\S+\n"
                   "pattern %num \\d+\n"
                   "pattern %ts %phrase\n"
                   "extract interm {  \n"
                   "  template <%num> (foo)[bar] $eventTimeStamp(%ts) end:'$timestamp(%ts)' THE END.\n"
                   "}\n",
            "extractions": [
                {"name": "interm", "pieces": [T("<"), P("\\d+"), T("> (foo)[bar] "), X("eventTimeStamp", P("\\S+")),
                                              T(" end:'"), X("timestamp", P("\\S+")), T("' THE END.")]},
            ],
            "cases": [
                {"input": "<123> (foo)[bar] 12:30:58 end:'15:07:00Z' THE END.", "match": [0]},
            ],
        },
        {
            "name": "testQuoted", "source": "test/PolyMatchTest.java:59-84",
            "def": "pattern %word (\\w+)\n"
                   "pattern %quoted \\\"[^\\\"]*\\\"\n"
                   "extract quoted {  \n"
                   "  template header value=$value(%quoted)\n"
                   "}\n"
                   "extract unquoted {  \n"
                   "  template header value=$value(%word)\n"
                   "}\n",
            "extractions": [
                {"name": "quoted", "pieces": [T("header value="), X("value", P('\\"[^\\"]*\\"'))]},
                {"name": "unquoted", "pieces": [T("header value="), X("value", W)]},
            ],
            "cases": [
                {"input": "header value=stuff", "match": [1]},
                {"input": 'header value="stuff"', "match": [0]},
            ],
        },
        {
            "name": "testComplex", "source": "test/PolyMatchTest.java:86-118",
            "def": "pattern %word [a-zA-Z]+\n"
                   "pattern %phrase [^ \t]+\n"
                   "pattern %num ([0-9]+)\n"
                   "pattern %ts %phrase\n"
                   "pattern %ip %phrase\n"
                   "pattern %maybeUUID %phrase\n"
                   "pattern %hostname %phrase\n"
                   "template @base <%num>$eventTimeStamp(%ts) $logAgent(%ip) RealSource: \"$logSrcIp(%ip)\"\\\n"
                   " Environment: \"$environment(%phrase)\"\\\n"
                   " UUID: \"$uuid(%maybeUUID)\"\\\n"
                   " RawMsg: <%num>$rawMsgTS(%word %num %phrase) $logSrcHostname(%hostname) $appname(%word)[$appPID(%num)]\n"
                   "\n"
                   "extract baseMatch {\n"
                   "  template @base\n"
                   "}\n",
            "extractions": [{"name": "baseMatch", "pieces": complex_base}],
            "cases": [
                {"input": '<86>2015-05-12T20:57:53.302858+00:00 10.1.11.141 RealSource: "10.10.5.3"'
                          ' Environment: "TEST"'
                          ' UUID: "NONE"'
                          ' RawMsg: <123>something 1324 keyboard-interactive/pam google.com sshd[137]',
                 "match": [0]},
            ],
        },
    ],
}

# ---------------------------------------------------------------------------
# B.3 full extract: id + captured values
# ---------------------------------------------------------------------------
WZ = P("([a-zA-Z]+)")
PHS = P("\\S+")
NUMD = P("\\d+")

full_base = [
    T("<"), NUMD, T(">"), X("eventTimeStamp", PHS), T(" "), X("logAgent", PHS),
    T(' RealSource: "'), X("logSrcIp", PHS),
    T('" Environment: "'), X("environment", PHS), T('" UUID: "'), X("uuid", PHS),
    T('" RawMsg: <'), NUMD, T(">"),
    X("rawMsgTS", WORDAZ, T(" "), NUMD, T(" "), PHS),
    T(" "), X("logSrcHostname", PHS), T(" "), X("appname", WORDAZ), T("["), X("appPID", NUMD), T("]"),
]
sshd = full_base + [
    T(": "), X("authStatus", T("Accepted")), T(" "), X("sshAuthMethod", PHS), T(" for "), X("user", PHS),
    T(" from "), X("srcIP", PHS), T(" port "), X("srcPort", NUMD), T(" "), X("sshProtocol", PHS),
]

FULL_DEF = (
    "### First, let's define basic patterns using 'patterns' (regexps)\n"
    "# 'phrase' means non-space-sequence of characters; 'word' letters; 'num' digits\n"
    "pattern %word [a-zA-Z]+\n"
    "pattern %phrase \\S+\n"
    "pattern %num \\d+\n"
    "# more semantic macros, loosely defined\n"
    "pattern %ts %phrase\n"
    "pattern %ip %phrase\n"
    "pattern %maybeUUID %phrase\n"
    "pattern %hostname %phrase\n"
    "pattern %any .*\n"
    "\n"
    "template @base <%num>$eventTimeStamp(%ts) $logAgent(%ip) RealSource: '$logSrcIp(%ip)'\\\n"
    " Environment: '$environment(%phrase)' UUID: '$uuid(%maybeUUID)'\\\n"
    " RawMsg: <%num>$rawMsgTS(%word %num %phrase) $logSrcHostname(%hostname)\\\n"
    " $appname(%word)[$appPID(%num)]\n"
    "\n"
    "extract sshdMatch {\n"
    "  template @base: $authStatus(Accepted) $sshAuthMethod(%phrase) for $user(%hostname)\\\n"
    " from $srcIP(%ip) port $srcPort(%num) $sshProtocol(%phrase)\n"
    "  append 'service':'ssh', 'logType':'security', 'serviceType':'authentication' \n"
    "}\n"
    "extract baseMatch {\n"
    "  template @base\n"
    "}\n"
).replace("'", '"')

FULL_IN1 = ("<86>2015-05-12T20:57:53.302858+00:00 10.1.11.141 RealSource:   '10.10.5.3'"
            " Environment: 'TEST' UUID: 'NO'"
            " RawMsg: <123>something 1324 more-or-less google.com sshd[137]").replace("'", '"')
FULL_IN2 = FULL_IN1 + ": Accepted keyboard-interactive/pam for badguy.ru from 1.2.3.4 port 58216 ssh2"

full = {
    "source": "test/FullExtractionTest.java, test/ParametricExtractorTest.java, "
              "test/ParametricTemplateTest.java, README.md:70-100, samples/simple.grp",
    "tests": [
        {
            "name": "testSimple", "source": "test/FullExtractionTest.java:11-42",
            "def": "pattern %word ([a-zA-Z]+)\n"
                   "template @base %word\n"
                   "extract double {  \n"
                   "  template @base value=$value(%word) value2=$value2(%word)\n"
                   "}\n"
                   "extract single {  \n"
                   "  template value=$value(%word)\n"
                   "}\n",
            "extractions": [
                {"name": "double", "pieces": [WZ, T(" value="), X("value", WZ), T(" value2="), X("value2", WZ)]},
                {"name": "single", "pieces": [T("value="), X("value", WZ)]},
            ],
            "cases": [
                {"input": "value=foobar", "id": "single", "id_as": "id",
                 "map": {"id": "single", "value": "foobar"}, "map_size": 2},
                {"input": "prefix value=a value2=b", "id": "double", "id_as": "id",
                 "map": {"id": "double", "value": "a", "value2": "b"}, "map_size": 3},
            ],
        },
        {
            "name": "testIntermediate", "source": "test/FullExtractionTest.java:44-66",
            "def": "pattern %ws \\s+\n"
                   "pattern %word [a-zA-Z]+\n"
                   "pattern %phrase \\S+\n"
                   "pattern %num \\d+\n"
                   "pattern %ts %phrase\n"
                   "pattern %ip %phrase\n"
                   "extract interm {  \n"
                   "  template <%num>$eventTimeStamp(%ts) $logAgent(%ip) RealSource: \"$logSrcIp(%ip)\"\n"
                   "}\n",
            "extractions": [
                {"name": "interm", "pieces": [T("<"), NUMD, T(">"), X("eventTimeStamp", PHS), T(" "),
                                              X("logAgent", PHS), T(' RealSource: "'), X("logSrcIp", PHS), T('"')]},
            ],
            "cases": [
                {"input": '<86>2015-05-12T20:57:53.302858+00:00 10.1.11.141 RealSource: "10.10.5.3"',
                 "id": "interm", "id_as": None, "map": {"logSrcIp": "10.10.5.3"}},
            ],
        },
        {
            "name": "testFull", "source": "test/FullExtractionTest.java:68-130",
            "def": FULL_DEF,
            "extractions": [
                {"name": "sshdMatch", "pieces": sshd,
                 "append": {"service": "ssh", "logType": "security", "serviceType": "authentication"}},
                {"name": "baseMatch", "pieces": full_base},
            ],
            "cases": [
                {"input": FULL_IN1, "not_null": True},
                {"input": FULL_IN2, "id": "sshdMatch", "id_as": None,
                 "map": {"user": "badguy.ru", "sshProtocol": "ssh2"}},
            ],
        },
        {
            "name": "parametricExtractor", "source": "test/ParametricExtractorTest.java:12-34",
            "def": "pattern %num ([0-9]+)\n"
                   "pattern %word ([a-zA-Z]+)\n"
                   "pattern %ip [a-zA-Z\\.]+\n"
                   "template @ip %ip\n"
                   "template @port %num\n"
                   "template @endpoint() $1(@ip):$2(@port)\n"
                   "extract Net {  \n"
                   "  template @endpoint($srcIp,$srcPort)/%word\n"
                   "}\n",
            "extractions": [
                {"name": "Net", "pieces": [X("srcIp", P("[a-zA-Z\\.]+")), T(":"), X("srcPort", NUMP), T("/"), WZ]},
            ],
            "cases": [
                {"input": "foo.bar.com:8080/user", "id": "Net", "id_as": None,
                 "map": {"srcIp": "foo.bar.com", "srcPort": "8080"}, "map_size": 2},
            ],
        },
        {
            "name": "parametricTemplate", "source": "test/ParametricTemplateTest.java:12-34",
            "def": "pattern %word ([a-zA-Z]+)\n"
                   "pattern %num ([0-9]+)\n"
                   "pattern %ip [a-zA-Z\\.]+\n"
                   "template @ip %ip\n"
                   "template @port %num\n"
                   "template @colonPair() @1:@2\n"
                   "extract Net {  \n"
                   "  template $endpoint(@colonPair(@ip,@port))/%word\n"
                   "}\n",
            "extractions": [
                {"name": "Net", "pieces": [X("endpoint", P("[a-zA-Z\\.]+"), T(":"), NUMP), T("/"), WZ]},
            ],
            "cases": [
                {"input": "foo.bar.com:8080/user", "id": "Net", "id_as": None,
                 "map": {"endpoint": "foo.bar.com:8080"}, "map_size": 1},
            ],
        },
        {
            "name": "readmeUsage", "source": "README.md:70-100",
            "def": "pattern %num \\d+\n"
                   "pattern %word \\w+\n"
                   "template @extractTime time=$time(%num)\n"
                   "template @extractVerb() verb=$1(%word)\n"
                   "extract SimpleEntry {\n"
                   "   template prefix: @extractTime @extractVerb($verb)\n"
                   "}\n",
            "extractions": [
                {"name": "SimpleEntry", "pieces": [T("prefix: "), T("time="), X("time", NUMD), T(" "),
                                                   T("verb="), X("verb", P("\\w+"))]},
            ],
            "cases": [
                {"input": "prefix: time=12546778 verb=PUT", "id": "SimpleEntry", "id_as": None,
                 "map": {"time": "12546778", "verb": "PUT"}, "map_size": 2},
            ],
        },
    ],
}

# ---------------------------------------------------------------------------
# B.4 dialect rewriting: test/util/RegexHelperTest.java
# ---------------------------------------------------------------------------
regexhelper = {
    "source": "test/util/RegexHelperTest.java:8-38",
    "quoteLiteralAsRegexp": [["", ""], ["(foo)", "\\(foo\\)"], ["[foo]", "\\[foo\\]"], ["a\\b", "a\\\\b"]],
    "massageRegexpForAutomaton": [
        ["", ""],
        ["[\\w]+", "[a-zA-Z_0-9]+"],
        ["\\w+", "[a-zA-Z_0-9]+"],
        ["[\\d\\s]+", "[0-9 \b\f\n\r\t]+"],
    ],
    "massageRegexpForJDK": [
        ["", ""],
        ["stuff([ab]+([de]+))", "stuff(?:[ab]+(?:[de]+))"],
        ["stuff\\(sic\\)", "stuff\\(sic\\)"],
    ],
}

# ---------------------------------------------------------------------------
# Benchmark definitions (BASELINE.json configs 1-2).  Not reference test
# vectors: definitions are the reference's own sample/README text; the
# expected values are the SURVEY's hand-derived known answers (Appendix B.5).
# ---------------------------------------------------------------------------
def readme_rule(verb_piece):
    return [T("["), X("timestamp", NUMD), T("]: "), X("verb", verb_piece), T(" "),
            X("timeTakenInMsec", NUMD), T("ms "), X("path", PHS)]


configs = {
    "source": "samples/simple.grp:1-23 ; README.md:114-135 ; SURVEY.md Appendix B.5",
    "simple_grp": {
        "extractions": [
            {"name": "sampleMatch",
             "pieces": [T("<"), NUMD, T(">"), X("eventTimeStamp", PHS), T(" ("), X("authStatus", T("Accepted")),
                        T(") ")]},
        ],
        "cases": [
            {"input": "<86>2015-05-12T20:57:53.302858+00:00 (Accepted) ", "match": [0],
             "id": "sampleMatch", "map": {"eventTimeStamp": "2015-05-12T20:57:53.302858+00:00",
                                          "authStatus": "Accepted"}},
            {"input": "<86>2015-05-12T20:57:53.302858+00:00 (Accepted)", "match": []},
            {"input": "<86>2015-05-12T20:57:53.302858+00:00 (Failed) ", "match": []},
        ],
    },
    "readme_3": {
        "extractions": [
            {"name": "PutRequest", "pieces": readme_rule(T("PUT")), "append": {"marker": "EXTRACTED"}},
            {"name": "GetRequest", "pieces": readme_rule(T("GET")), "append": {"marker": "EXTRACTED"}},
            {"name": "OtherRequest", "pieces": readme_rule(P("\\w+")), "append": {"marker": "EXTRACTED"}},
        ],
        "cases": [
            {"input": "[1]: GET 5ms /x", "match": [1, 2], "id": "GetRequest"},
            {"input": "[1]: PUT 5ms /x", "match": [0, 2], "id": "PutRequest"},
            {"input": "[1]: POST 5ms /x", "match": [2], "id": "OtherRequest"},
            {"input": "[1]: GETX 5ms /x", "match": [2], "id": "OtherRequest"},
            {"input": "1: GET 5ms /x", "match": []},
            {"input": "102456879: GET 123ms 200 /rest-service/v1/endpoint?foo=bar", "match": []},
        ],
    },
}

if __name__ == "__main__":
    for name, obj in [("multipattern", multipattern), ("polymatch", polymatch), ("full_extraction", full),
                      ("regexhelper", regexhelper), ("configs", configs)]:
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(obj, f, indent=1, ensure_ascii=True)
            f.write("\n")
    print("wrote fixtures to", HERE)
