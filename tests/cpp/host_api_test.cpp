// Exercises include/gorp.hpp (the C++ host-side mirror).  Reads like test/FullExtractionTest.java.
//   host_api_test            : host-only checks (definition parsing, RegexHelper, errors) -- no GPU needed
//   host_api_test --gpu      : also runs extract() / extractSafe() / extractBatch() on the device
#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>

#include "gorp.hpp"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

static const char* DEF =
    "pattern %word ([a-zA-Z]+)\n"
    "template @base %word\n"
    "extract double {  \n"
    "  template @base value=$value(%word) value2=$value2(%word)\n"
    "}\n"
    "extract single {  \n"
    "  template value=$value(%word)\n"
    "  append \"marker\" : \"EXTRACTED\"\n"
    "}\n";

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && strcmp(argv[1], "--gpu") == 0;
    using namespace gorp;
    // test/util/RegexHelperTest.java
    CHECK(RegexHelper::quoteLiteralAsRegexp("(foo)") == "\\(foo\\)");
    CHECK(RegexHelper::massageRegexpForAutomaton("\\w+") == "[a-zA-Z_0-9]+");
    CHECK(RegexHelper::massageRegexpForJDK("stuff([ab]+([de]+))") == "stuff(?:[ab]+(?:[de]+))");
    try { RegexHelper::massageRegexpForAutomaton("\\q"); CHECK(false); } catch (std::invalid_argument&) {}

    auto def = DefinitionReader::reader(DEF).read(gpu ? 0 : GX_CREATE_HOST_ONLY);
    CHECK(def->getExtractions().size() == 2);
    CHECK(def->getExtractions()[0].getName() == "double" && def->getExtractions()[1].getName() == "single");
    CHECK(def->getExtractions()[0].extractorNames.size() == 2 && def->getExtractions()[0].extractorNames[1] == "value2");
    CHECK(def->getExtractions()[1].appendJson == "{\"marker\":\"EXTRACTED\"}");
    try {
        DefinitionReader::reader("pattern %a a\ntemplate @base (%a:foo)\n").read(GX_CREATE_HOST_ONLY);
        CHECK(false);
    } catch (DefinitionParseException& e) { CHECK(std::string(e.what()).find("No extraction definitions found") != std::string::npos); }

    if (!gpu) {
        // without a device the extract path must fail loudly (there is no CPU fallback)
        try { def->extract("value=foobar"); CHECK(false); } catch (GorpError& e) { CHECK(e.code == GX_E_DEVICE); }
        printf("host_api_test: host-only checks ok\n");
        return 0;
    }
    // test/FullExtractionTest.java:26-41
    auto result = def->extract("value=foobar");
    CHECK(result);
    CHECK(result->getId() == "single");
    auto stuff = result->asMap("id");
    // (this DEF adds `append "marker":"EXTRACTED"` to the reference test's second extraction: one more entry, typed)
    CHECK(stuff.size() == 3 && stuff[0].first == "id" && stuff[0].second == "single" && stuff[1].first == "value" && stuff[1].second == "foobar");
    CHECK(stuff[2].first == "marker" && stuff[2].second.kind == MapValue::Json && stuff[2].second.text == "\"EXTRACTED\"");
    result = def->extract("prefix value=a value2=b");
    CHECK(result && result->getId() == "double");
    stuff = result->asMap("id");
    CHECK(stuff.size() == 3 && stuff[1].second == "a" && stuff[2].second == "b");
    CHECK(!def->extract("nothing here"));

    // DFA says yes, regexp says no ('.' vs carriage return): exception, or null from extractSafe
    auto dot = DefinitionReader::reader("extract r {\n template a$x(%{.*})b\n}\n").read();
    try { dot->extract("a\rb"); CHECK(false); } catch (ExtractionException& e) { CHECK(e.getInput() == "a\rb"); }
    CHECK(!dot->extractSafe("a\rb"));

    // batch
    const char* lines = "value=xprefix value=a value2=bnope";
    const uint32_t off[4] = {0, 7, 30, 34};
    int32_t mid[3], caps[3 * 4];
    def->extractBatch(reinterpret_cast<const uint8_t*>(lines), off, 3, mid, caps);
    CHECK(mid[0] == 1 && mid[1] == 0 && mid[2] == -1);
    CHECK(caps[0] == 6 && caps[1] == 7 && caps[4] == 13 && caps[5] == 14 && caps[6] == 22 && caps[7] == 23);
    // the steps either side: raw text -> lines -> extract (terminators ignored) -> JSON Lines
    const std::string text = "value=x\r\nprefix value=a value2=b\nnope\rvalue=q\"uote";
    auto offs = splitLines(reinterpret_cast<const uint8_t*>(text.data()), text.size());
    CHECK(offs.size() == 5 && offs[1] == 9 && offs[2] == 33 && offs[3] == 38 && offs[4] == text.size());
    gx_batch_opts bo{};
    bo.struct_size = sizeof bo;
    bo.strip_eol = 1;
    std::vector<int32_t> m2(4), c2(4 * 4);
    def->extractBatch(reinterpret_cast<const uint8_t*>(text.data()), offs.data(), 4, m2.data(), c2.data(), &bo);
    CHECK(m2[0] == 1 && m2[1] == 0 && m2[2] == -1 && m2[3] == -1);   // value=q"uote: %word is letters only
    const std::string jl = def->resultsToJsonl(reinterpret_cast<const uint8_t*>(text.data()), offs.data(), 4, m2.data(), c2.data(), "id");
    CHECK(jl == "{\"id\":\"single\",\"value\":\"x\",\"marker\":\"EXTRACTED\"}\n{\"id\":\"double\",\"value\":\"a\",\"value2\":\"b\"}\n");
    // the plugin seam's product: one extraction's regexp alone
    auto one = def->matchExtraction(1, "value=foobar");
    CHECK(one && one->getId() == "single" && one->value(0) == "foobar");
    CHECK(!def->matchExtraction(0, "value=foobar"));
    uint64_t nl = 0, nm = 0, nx = 0;
    CHECK(def->textToJsonl(text, "id", &nl, &nm, &nx) == jl && nl == 4 && nm == 2 && nx == 0);
    printf("host_api_test: GPU checks ok\n");
    return 0;
}
