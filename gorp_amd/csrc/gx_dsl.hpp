// gx_dsl.hpp -- native front-end for Gorp's definition language (see gx_dsl.cpp).
#pragma once
#include <functional>
#include <string>
#include <utility>
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

// JSON helpers shared with result materialisation (gx_api.cpp: gx_results_to_jsonl).
// Entries of a canonical JSON object text in order: (key decoded to UTF-8, value as raw JSON text).
std::vector<std::pair<std::string, std::string>> json_object_entries(const std::string& canonical_object);
// JSON string literal with Jackson's default escaping: \" \\ \b \t \n \f \r, other controls as \u00XX (upper-case
// hex), everything else (including non-ASCII UTF-8) verbatim.
std::string json_quote(const std::string& utf8);
// Parses a JSON object text and returns it in canonical (compact, duplicate keys merged) form; throws GxError(GX_E_ARG).
std::string canonical_json_object(const std::string& text);

}  // namespace dsl
}  // namespace gx
