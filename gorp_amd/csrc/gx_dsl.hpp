// gx_dsl.hpp -- native front-end for Gorp's definition language (see gx_dsl.cpp).
#pragma once
#include <functional>
#include <string>
#include <vector>

#include "gx_common.hpp"

namespace gx {
namespace dsl {

// One piece of a flattened extraction (core/model: LiteralText, LiteralPattern, ExtractorExpression).
struct Piece {
    enum Kind { TEXT, PATTERN, EXTRACTOR } kind = TEXT;
    std::string text;  // literal text, pattern source, or extractor name (UTF-8)
    std::vector<Piece> kids;
};

// core/model/FlattenedExtraction.java
struct Extraction {
    std::string name;
    std::vector<Piece> pieces;
    std::vector<std::string> extractor_names;  // pre-order == capture group order
    std::string append_json;                   // canonical JSON object text, or empty
};

// DefinitionReader.reader(text).read() up to (not including) Gorp.construct.  Throws GxError(GX_E_DEFINITION).
std::vector<Extraction> read_definition(const std::string& utf8_text, const std::string& source_ref);

// Gorp._buildExtractor over one flattened extraction (core/Gorp.java:94-129).
void build_regex_strings(const Extraction& x, std::string& automaton_rx, std::string& jdk_rx);

// JSON view of a definition at one of the reference's intermediate stages, for tests that replay the
// reference's own DSL unit tests: "uncooked" (readUncooked), "cooked" (resolvePatterns + resolveTemplates),
// "flattened" (everything, plus the two regex strings per extraction).
std::string dump_json(const std::string& utf8_text, const std::string& source_ref, const std::string& stage);

}  // namespace dsl
}  // namespace gx
