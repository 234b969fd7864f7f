// gorp.hpp -- C++ host-side mirror of the reference's API for the match-and-extract path, header-only over the
// C ABI of libgorp_hip.so (include/gorp_hip.h).  Same names, argument meaning and error behaviour as
// salesforce/gorp (core/ = gorp-core/src/main/java/com/salesforce/gorp/):
//
//   gorp::DefinitionReader::reader(text).read()  core/DefinitionReader.java:58-84
//   gorp::Gorp::extract / extractSafe            core/Gorp.java:145-186
//   gorp::ExtractionResult::getId / asMap        core/ExtractionResult.java:39-88
//   gorp::ExtractionException                    core/ExtractionException.java:15-34
//   gorp::DefinitionParseException               core/DefinitionParseException.java
//   gorp::RegexHelper                            core/util/RegexHelper.java
//
// plus the batch entry points the GPU needs (Gorp::extractBatch over a CSR byte buffer, splitLines before it,
// Gorp::resultsToJsonl after it).  All matching runs in the
// HIP kernels; nothing here computes a match on the CPU.  Link: -lgorp_hip -lamdhip64.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gorp_hip.h"

namespace gorp {

struct DefinitionParseException : std::runtime_error {
    int code;
    DefinitionParseException(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct ExtractionException : std::runtime_error {
    std::string input;
    ExtractionException(std::string in, const std::string& m) : std::runtime_error(m), input(std::move(in)) {}
    const std::string& getInput() const { return input; }
};

struct GorpError : std::runtime_error {
    int code;
    GorpError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// core/model/CookedExtraction.java (data only)
struct CookedExtraction {
    std::string name;
    std::vector<std::string> extractorNames;
    std::string appendJson;  // getExtra() as JSON object text; empty when there is none
    std::vector<std::pair<std::string, std::string>> extra;  // getExtra() entry by entry: key, value as JSON text
    const std::string& getName() const { return name; }
};

// A value of ExtractionResult.asMap(): the captured text, null (Matcher.group() == null), or -- for `append`
// entries, whose values are typed -- JSON text.
struct MapValue {
    enum Kind { Null, String, Json } kind;
    std::string text;
    bool operator==(const char* s) const { return kind == String && text == s; }
};

// core/ExtractionResult.java.  Values are (present, text) because Matcher.group may return null.
class ExtractionResult {
public:
    ExtractionResult(const CookedExtraction* x, std::string input, std::vector<std::pair<bool, std::string>> values)
        : x_(x), input_(std::move(input)), values_(std::move(values)) {}
    const std::string& getId() const { return x_->name; }
    const std::string& getInput() const { return input_; }
    const CookedExtraction& getMatchedExtraction() const { return *x_; }
    // core/ExtractionResult.java:65-88 on a LinkedHashMap: the id (optional) first, every extractor name -> its text
    // or null in group order, then the `append` entries; a key put again keeps its position and takes the new value
    std::vector<std::pair<std::string, MapValue>> asMap(const char* idAs = nullptr) const {
        std::vector<std::pair<std::string, MapValue>> m;
        auto put = [&m](const std::string& key, MapValue v) {
            for (auto& e : m) if (e.first == key) { e.second = std::move(v); return; }
            m.emplace_back(key, std::move(v));
        };
        if (idAs) put(idAs, MapValue{MapValue::String, x_->name});
        for (size_t i = 0; i < values_.size(); ++i)
            put(x_->extractorNames[i], values_[i].first ? MapValue{MapValue::String, values_[i].second} : MapValue{MapValue::Null, ""});
        for (auto& kv : x_->extra) put(kv.first, MapValue{MapValue::Json, kv.second});
        return m;
    }
    bool has(size_t group) const { return group < values_.size() && values_[group].first; }
    const std::string& value(size_t group) const { return values_[group].second; }

private:
    const CookedExtraction* x_;
    std::string input_;
    std::vector<std::pair<bool, std::string>> values_;
};

struct RegexHelper {
    static std::string quoteLiteralAsRegexp(const std::string& t) { return call(gx_quote_literal_as_regexp, t); }
    static std::string massageRegexpForAutomaton(const std::string& p) { return call(gx_massage_regexp_for_automaton, p); }
    static std::string massageRegexpForJDK(const std::string& p) { return call(gx_massage_regexp_for_jdk, p); }

private:
    typedef int (*Fn)(const char*, char*, size_t, size_t*);
    static std::string call(Fn fn, const std::string& in) {
        size_t n = 0;
        std::string out(8 * in.size() + 64, '\0');
        int rc = fn(in.c_str(), &out[0], out.size(), &n);
        if (rc == GX_E_ARG && n + 1 > out.size()) { out.assign(n + 1, '\0'); rc = fn(in.c_str(), &out[0], out.size(), &n); }
        if (rc == GX_E_REGEX_SYNTAX) throw std::invalid_argument(gx_last_error());  // IllegalArgumentException
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        out.resize(n);
        return out;
    }
};

// core/Gorp.java
class Gorp {
public:
    ~Gorp() { gx_destroy(h_); }
    Gorp(const Gorp&) = delete;
    Gorp& operator=(const Gorp&) = delete;

    const std::vector<CookedExtraction>& getExtractions() const { return extractions_; }

    // Gorp.extract(String): the line as UTF-8 (converted to UTF-16 code units, which is what the reference walks).
    // Returns nullptr for "no match"; throws ExtractionException when the matcher and the capture regexp disagree.
    std::unique_ptr<ExtractionResult> extract(const std::string& input, bool allowFallbacks = false) const {
        std::u16string u = to_utf16(input);
        int32_t id = 0;
        std::vector<int32_t> caps(2 * static_cast<size_t>(gx_max_groups(h_)) + 2, -1);
        int rc = gx_extract_one_utf16(h_, reinterpret_cast<const uint16_t*>(u.data()), static_cast<int32_t>(u.size()), &id, caps.data());
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        return materialise(input, u, id, caps.data(), allowFallbacks);
    }
    std::unique_ptr<ExtractionResult> extractSafe(const std::string& input) const { return extract(input, true); }

    // CookedExtraction.match(String) (core/model/CookedExtraction.java:61) for extraction k: its capture regexp alone,
    // no matcher stage; nullptr when the regexp does not match the whole line.
    std::unique_ptr<ExtractionResult> matchExtraction(size_t k, const std::string& input) const {
        std::u16string u = to_utf16(input);
        int32_t matched = 0;
        std::vector<int32_t> caps(2 * static_cast<size_t>(gx_max_groups(h_)) + 2, -1);
        int rc = gx_capture_one_utf16(h_, static_cast<int32_t>(k), reinterpret_cast<const uint16_t*>(u.data()), static_cast<int32_t>(u.size()),
                                      &matched, caps.data());
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        return matched ? materialise(input, u, static_cast<int32_t>(k), caps.data(), false) : nullptr;
    }

    // The batch path: lines as one Latin-1 byte buffer + offsets[n+1]; match_id[n], caps[n * 2*maxGroups()].
    void extractBatch(const uint8_t* bytes, const uint32_t* offsets, uint64_t n, int32_t* match_id, int32_t* caps,
                      const gx_batch_opts* opts = nullptr) const {
        int rc = gx_extract_batch(h_, bytes, offsets, n, match_id, caps, opts);
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
    }
    // Result materialisation for a finished batch: asMap(idAs) of every matched line as JSON Lines (gx_results_to_jsonl)
    std::string resultsToJsonl(const uint8_t* bytes, const uint32_t* offsets, uint64_t n, const int32_t* match_id, const int32_t* caps,
                               const char* idAs = nullptr, const gx_batch_opts* opts = nullptr) const {
        uint64_t size = 0;
        int rc = gx_results_to_jsonl(h_, bytes, offsets, n, match_id, caps, idAs, nullptr, 0, &size, nullptr, opts);
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        std::string out(static_cast<size_t>(size), '\0');
        rc = gx_results_to_jsonl(h_, bytes, offsets, n, match_id, caps, idAs, reinterpret_cast<uint8_t*>(&out[0]), size, &size, nullptr, opts);
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        return out;
    }
    // Whole files: raw text in, JSON Lines of the matched lines out (gx_text_to_jsonl); counts are optional
    std::string textToJsonl(const std::string& text, const char* idAs = nullptr, uint64_t* nLines = nullptr, uint64_t* nMatched = nullptr,
                            uint64_t* nExceptions = nullptr) const {
        uint64_t size = 0;
        const uint8_t* p = reinterpret_cast<const uint8_t*>(text.data());
        int rc = gx_text_to_jsonl(h_, p, text.size(), idAs, nullptr, 0, &size, nLines, nMatched, nExceptions, nullptr);
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        std::string out(static_cast<size_t>(size), '\0');
        rc = gx_text_to_jsonl(h_, p, text.size(), idAs, reinterpret_cast<uint8_t*>(&out[0]), size, &size, nLines, nMatched, nExceptions, nullptr);
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        return out;
    }
    int maxGroups() const { return gx_max_groups(h_); }
    gx_handle* handle() const { return h_; }

private:
    friend class DefinitionReader;
    Gorp(gx_handle* h, std::vector<CookedExtraction> x) : h_(h), extractions_(std::move(x)) {}
    gx_handle* h_;
    std::vector<CookedExtraction> extractions_;

    std::unique_ptr<ExtractionResult> materialise(const std::string& input, const std::u16string& u, int32_t id, const int32_t* caps,
                                                  bool safe) const {
        if (id == -1) return nullptr;
        if (id <= -2) {
            const CookedExtraction& x = extractions_[static_cast<size_t>(-2 - id)];
            if (safe) return nullptr;  // core/Gorp.java:178-185
            throw ExtractionException(input, "Internal error: high-level match for extraction #" + std::to_string(-2 - id) + " (" + x.name +
                                                 ") failed to match generated regexp");
        }
        const CookedExtraction& x = extractions_[static_cast<size_t>(id)];
        std::vector<std::pair<bool, std::string>> values;
        const int ng = gx_num_groups(h_, id);
        for (int g = 0; g < ng; ++g) {
            const int32_t b = caps[2 * g], e = caps[2 * g + 1];
            if (b < 0) values.emplace_back(false, std::string());
            else values.emplace_back(true, to_utf8(u.substr(static_cast<size_t>(b), static_cast<size_t>(e - b))));
        }
        return std::unique_ptr<ExtractionResult>(new ExtractionResult(&x, input, std::move(values)));
    }

    static std::u16string to_utf16(const std::string& s) {
        std::u16string out;
        for (size_t i = 0; i < s.size();) {
            uint32_t cp = static_cast<unsigned char>(s[i]);
            int extra = cp < 0x80 ? 0 : (cp >> 5) == 6 ? 1 : (cp >> 4) == 14 ? 2 : 3;
            cp = extra == 0 ? cp : cp & (0x3F >> extra);
            ++i;
            for (int k = 0; k < extra && i < s.size(); ++k, ++i) cp = (cp << 6) | (static_cast<unsigned char>(s[i]) & 0x3F);
            if (cp > 0xFFFF) {
                cp -= 0x10000;
                out.push_back(static_cast<char16_t>(0xD800 | (cp >> 10)));
                out.push_back(static_cast<char16_t>(0xDC00 | (cp & 0x3FF)));
            } else out.push_back(static_cast<char16_t>(cp));
        }
        return out;
    }
    static std::string to_utf8(const std::u16string& s) {
        std::string out;
        for (size_t i = 0; i < s.size(); ++i) {
            uint32_t cp = s[i];
            if (cp >= 0xD800 && cp <= 0xDBFF && i + 1 < s.size()) { cp = 0x10000 + ((cp & 0x3FF) << 10) + (s[i + 1] & 0x3FF); ++i; }
            if (cp < 0x80) out += static_cast<char>(cp);
            else if (cp < 0x800) { out += static_cast<char>(0xC0 | (cp >> 6)); out += static_cast<char>(0x80 | (cp & 0x3F)); }
            else if (cp < 0x10000) {
                out += static_cast<char>(0xE0 | (cp >> 12)); out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); out += static_cast<char>(0x80 | (cp & 0x3F));
            } else {
                out += static_cast<char>(0xF0 | (cp >> 18)); out += static_cast<char>(0x80 | ((cp >> 12) & 0x3F));
                out += static_cast<char>(0x80 | ((cp >> 6) & 0x3F)); out += static_cast<char>(0x80 | (cp & 0x3F));
            }
        }
        return out;
    }
};

// Line ingestion (gx_split_lines): BufferedReader.readLine() boundaries of a raw text buffer as CSR offsets; the lines
// keep their terminators, so pass gx_batch_opts.strip_eol = 1 to extractBatch.
inline std::vector<uint32_t> splitLines(const uint8_t* bytes, uint64_t size) {
    std::vector<uint32_t> offsets(static_cast<size_t>(size) + 2);
    uint64_t n = 0;
    int rc = gx_split_lines(bytes, size, offsets.data(), size + 1, &n, nullptr, nullptr);
    if (rc != GX_OK) throw GorpError(rc, gx_last_error());
    offsets.resize(static_cast<size_t>(n) + 1);
    return offsets;
}

// core/DefinitionReader.java
class DefinitionReader {
public:
    static DefinitionReader reader(const std::string& contents, const std::string& sourceRef = "<input string>") {
        return DefinitionReader(contents, sourceRef);
    }
    // flags: 0, or GX_CREATE_HOST_ONLY to parse and compile without touching a GPU
    std::unique_ptr<Gorp> read(uint32_t flags = 0) const {
        gx_handle* h = nullptr;
        int rc = gx_create_from_definition(text_.c_str(), ref_.c_str(), flags, &h);
        if (rc == GX_E_DEFINITION || rc == GX_E_REGEX_SYNTAX || rc == GX_E_UNSUPPORTED_CONSTRUCT || rc == GX_E_LIMIT)
            throw DefinitionParseException(rc, gx_last_error());
        if (rc != GX_OK) throw GorpError(rc, gx_last_error());
        std::vector<CookedExtraction> xs;
        for (int32_t k = 0; k < gx_num_extractions(h); ++k) {
            CookedExtraction x;
            x.name = gx_extraction_name(h, k);
            for (int32_t g = 0; gx_extractor_name(h, k, g); ++g) x.extractorNames.push_back(gx_extractor_name(h, k, g));
            if (const char* a = gx_extraction_append_json(h, k)) x.appendJson = a;
            for (int32_t j = 0; j < gx_extraction_append_count(h, k); ++j)
                x.extra.emplace_back(gx_extraction_append_key(h, k, j), gx_extraction_append_value_json(h, k, j));
            xs.push_back(std::move(x));
        }
        return std::unique_ptr<Gorp>(new Gorp(h, std::move(xs)));
    }

private:
    DefinitionReader(std::string t, std::string r) : text_(std::move(t)), ref_(std::move(r)) {}
    std::string text_, ref_;
};

}  // namespace gorp
