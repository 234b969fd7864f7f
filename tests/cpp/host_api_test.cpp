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
    CHECK(stuff.size() == 2 && stuff[0].first == "id" && stuff[0].second == "single" && stuff[1].first == "value" && stuff[1].second == "foobar");
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
    printf("host_api_test: GPU checks ok\n");
    return 0;
}
