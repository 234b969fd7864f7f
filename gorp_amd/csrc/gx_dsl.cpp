// gx_dsl.cpp -- native front-end for Gorp's definition language (.grp files): definition text -> flattened extractions.
//
// the step BEFORE the hot path, so that a .grp file can drive the GPU engine with no JVM.
// What it has to agree with is the reference's BEHAVIOUR -- the language (core/DefinitionReader.java:74-181,189-640), its
// resolution rules (core/model/CookedDefinitions.java:57-453), line joining (core/io/InputLineReader.java:69-150) and the
// error texts the reference's own tests assert (replayed by tests/test_dsl.py) -- not with its structure.  Here:
//
//   LineSource   physical rows -> logical lines (comments and blanks dropped, backslash continuations joined), each with
//                a segment table that maps an offset back to (row, column) of the file
//   tokenize     a template body -> flat tokens: text, the punctuation ( ) , and the three sigils with their payload
//                (%name %{inline} @name @3 $name $3); doubled sigils are already plain text.  Lexical trouble becomes a
//                token too, so that it is reported when -- and only if -- the parser gets there
//   BodyParser   recursive descent over the tokens -> a tree of Nodes per declaration.  What a parenthesis means depends
//                on where the parser is (text in a template, the end of an extractor's body inside one, an argument
//                separator inside a parameter list), so the parser, not the lexer, decides
//   Linker       patterns: memoised depth-first search over the reference graph, three-colour marking for cycles;
//                templates: one pass in declaration order (a reference to a template that is not finished yet freezes it
//                as empty -- the reference's forward-reference quirk); extractions: substitution of
//                parameter bindings and expansion into flat pieces
#include "gx_dsl.hpp"

#include <cstdio>
#include <map>
#include <set>

namespace gx {
namespace dsl {

std::string json_merge(const std::string& a, const std::string& b);              // gx_json.cpp
std::string parse_append_value(std::string raw, std::string* not_object);         // gx_json.cpp

namespace {

std::string u8(const ustr& s) { return u16_to_utf8(s); }

// ======================================================================================================================
// Logical lines
// ======================================================================================================================
struct LogicalLine {
    std::string source;
    ustr text;                            // the continuation segments, joined
    struct Segment { int at, row; };      // text[at ...] comes from physical row `row`, from its first column on
    std::vector<Segment> segments;        // ascending `at`; never empty

    int first_row() const { return segments.front().row; }
    int size() const { return static_cast<int>(text.size()); }
    // "[source (row,column)]" of an offset, both 1-based
    std::string where(int offset) const {
        size_t k = segments.size() - 1;
        while (k > 0 && segments[k].at > offset) --k;
        return "[" + source + " (" + std::to_string(segments[k].row) + "," + std::to_string(offset - segments[k].at + 1) + ")]";
    }
};
typedef std::shared_ptr<const LogicalLine> LineRef;

[[noreturn]] void fail_at(const LineRef& line, int offset, const std::string& message) {
    throw GxError(GX_E_DEFINITION, "(" + (line ? line->where(offset) : std::string("N/A")) + "): " + message);
}

class LineSource {
public:
    LineSource(const ustr& text, const std::string& source) : source_(source) {
        // the line ends a BufferedReader knows: \n, \r, \r\n; a final terminator does not open another line
        size_t begin = 0;
        for (size_t i = 0; i < text.size(); ++i) {
            if (text[i] != u'\n' && text[i] != u'\r') continue;
            rows_.push_back(text.substr(begin, i - begin));
            if (text[i] == u'\r' && i + 1 < text.size() && text[i + 1] == u'\n') ++i;
            begin = i + 1;
        }
        if (begin < text.size()) rows_.push_back(text.substr(begin));
    }

    // the next logical line, or null at the end of the input
    LineRef next() {
        while (cursor_ < rows_.size() && skippable(rows_[cursor_])) ++cursor_;
        if (cursor_ >= rows_.size()) return nullptr;
        auto line = std::make_shared<LogicalLine>();
        line->source = source_;
        // a row that ends in a backslash goes on in the next row, whatever that row holds (comment, blank, anything)
        for (bool more = true; more; ++cursor_) {
            if (cursor_ >= rows_.size()) fail_here("Unexpected end-of-input when expecting line continuation'");
            ustr row = rows_[cursor_];
            more = !row.empty() && row.back() == u'\\';
            if (more) row.pop_back();
            line->segments.push_back({static_cast<int>(line->text.size()), static_cast<int>(cursor_) + 1});
            line->text += row;
        }
        return line;
    }
    // an error about the input as a whole: reported with the number of rows read so far
    [[noreturn]] void fail_here(const std::string& message) const {
        throw GxError(GX_E_DEFINITION, "(" + source_ + ", row " + std::to_string(std::min(cursor_, rows_.size())) + "): " + message);
    }

private:
    static bool skippable(const ustr& row) {  // blank, or a comment
        for (char16_t c : row) if (c > 0x20) return c == u'#';
        return true;
    }
    std::string source_;
    std::vector<ustr> rows_;
    size_t cursor_ = 0;
};

// ======================================================================================================================
// Characters and names
// ======================================================================================================================
bool is_blank(char16_t c) { return c <= u' '; }
bool is_digit(char16_t c) { return c >= u'0' && c <= u'9'; }
// Java identifier characters: exact on ASCII; beyond it letters are approximated as "U+00A0 and up, minus the Latin-1
// symbols", and identifier-ignorable controls continue a name
bool starts_name(char16_t c) {
    if (c < 0x80) return (c >= u'a' && c <= u'z') || (c >= u'A' && c <= u'Z') || c == u'_' || c == u'$';
    if (c < 0xA0) return false;
    if (c < 0x100) return (c >= 0xA2 && c <= 0xA5) || c == 0xAA || c == 0xB5 || c == 0xBA || (c >= 0xC0 && c != 0xD7 && c != 0xF7);
    return true;
}
bool continues_name(char16_t c) {
    return c == u'-' || starts_name(c) || is_digit(c) || c <= 0x08 || (c >= 0x0E && c <= 0x1B) || (c >= 0x7F && c <= 0x9F) || c == 0xAD;
}
std::string describe_char(char16_t c) {
    char code[32];
    if (c < 0x20 || (c >= 0x7F && c <= 0x9F)) { snprintf(code, sizeof code, "code 0x%04x", static_cast<int>(c)); return code; }
    snprintf(code, sizeof code, " (code 0x%04x)", static_cast<int>(c));
    return "'" + u8(ustr(1, c)) + "'" + code;
}
int skip_blanks(const ustr& s, int at) {
    while (at < static_cast<int>(s.size()) && is_blank(s[at])) ++at;
    return at;
}

// A name as the language writes it: quoted ('...' or "...", anything inside), an identifier (hyphens allowed behind the
// first character), or a run of digits (a positional parameter -- whether one is allowed is the reader's business).
struct Name {
    enum State { ABSENT, GIVEN, DIGITS, NOTHING_LEFT, UNCLOSED } state = ABSENT;
    ustr text;
    char16_t quote = 0;
    int at = 0, end = 0;    // where the name was looked for, and where the scan stopped
    bool present() const { return state == GIVEN || state == DIGITS; }
    std::string shown() const { return present() ? u8(text) : "null"; }  // (the reference's messages print its null)
};
Name scan_name(const ustr& s, int at) {
    const int n = static_cast<int>(s.size());
    Name name;
    name.at = name.end = at;
    if (at >= n) { name.state = Name::NOTHING_LEFT; return name; }
    const char16_t c = s[at];
    if (c == u'"' || c == u'\'') {
        const size_t close = s.find(c, at + 1);
        name.quote = c;
        if (close == ustr::npos) { name.state = Name::UNCLOSED; return name; }
        name.text = s.substr(at + 1, close - at - 1);
        name.state = Name::GIVEN;
        name.end = static_cast<int>(close) + 1;
    } else if (starts_name(c) || is_digit(c)) {
        const bool digits = is_digit(c);
        int e = at + 1;
        while (e < n && (digits ? is_digit(s[e]) : continues_name(s[e]))) ++e;
        name.text = s.substr(at, e - at);
        name.state = digits ? Name::DIGITS : Name::GIVEN;
        name.end = e;
    }
    return name;
}
// the two ways a name can be lexically broken, and a positional where none may stand -- `what` is the noun of the place
void require_name(const LineRef& line, const Name& name, const char* what, bool positional_ok) {
    if (name.state == Name::NOTHING_LEFT) fail_at(line, line->size(), std::string("Missing ") + what + " name");
    if (name.state == Name::UNCLOSED)
        fail_at(line, line->size(), std::string("Missing closing quote ('") + static_cast<char>(name.quote) + "') for " + what + " name");
    if (name.state == Name::DIGITS && !positional_ok)
        fail_at(line, name.at, std::string("Invalid variable reference instead of ") + what +
                                   " name: can not use variable references here (missing parenthesis after template name?)");
}
// a declared name: blanks (or the end of the line) must follow; returns where the rest of the line begins
int end_of_declared_name(const LineRef& line, const Name& name, const char* what) {
    const ustr& s = line->text;
    if (name.end >= line->size()) return name.end;
    if (!is_blank(s[name.end])) fail_at(line, name.end, std::string("Missing space character after ") + what + " name '" + name.shown() + "'");
    return skip_blanks(s, name.end);
}
// the value of a run of digits, saturating (1 .. 999999 is what the language accepts)
long positional_value(const ustr& digits) {
    long v = 0;
    for (char16_t c : digits) { v = v * 10 + (c - u'0'); if (v > 2000000000L) return 2000000000L; }
    return v;
}

// ======================================================================================================================
// Tokens of a template body
// ======================================================================================================================
struct Token {
    enum Kind { TEXT, PUNCT, PATTERN_NAME, PATTERN_INLINE, AT, DOLLAR, ORPHAN, UNCLOSED_INLINE, END } kind = END;
    ustr text;        // TEXT: the characters (doubled sigils already single); PUNCT: one of ( ) ,; PATTERN_INLINE: its source
    Name name;        // PATTERN_NAME, AT, DOLLAR
    int start = 0;    // offset of the token's first character (the sigil, for sigil tokens)
    int at = 0;       // offset of its payload (behind the sigil / the opening brace)
    int end = 0;      // offset just behind the token
    char16_t sigil = 0;
};

// raw_first: the character at `from` is a sigil even when the one behind it is the same (inside a parameter list nothing
// is escaped: "$$x" there is the extractor name "$x")
std::vector<Token> tokenize(const ustr& s, int from, bool raw_first = false) {
    std::vector<Token> out;
    const int n = static_cast<int>(s.size());
    auto is_sigil = [](char16_t c) { return c == u'%' || c == u'@' || c == u'$'; };
    auto is_punct = [](char16_t c) { return c == u'(' || c == u')' || c == u','; };
    int i = from;
    while (i < n) {
        Token t;
        t.start = t.at = i;
        const char16_t c = s[i];
        if (is_punct(c)) {
            t.kind = Token::PUNCT;
            t.text = ustr(1, c);
            t.end = ++i;
        } else if (is_sigil(c) && (!(i + 1 < n && s[i + 1] == c) || (raw_first && i == from))) {
            t.sigil = c;
            t.at = i + 1;
            if (i + 1 >= n) {
                t.kind = Token::ORPHAN;
                t.end = i = n;
            } else if (c == u'%' && s[i + 1] == u'{') {
                t.at = i + 2;
                int depth = 1, k = i + 2;
                for (; k < n; ++k) {   // braces nest; a backslash hides the next character
                    if (s[k] == u'\\') { ++k; continue; }
                    if (s[k] == u'{') ++depth;
                    else if (s[k] == u'}' && --depth == 0) break;
                }
                if (k >= n) { t.kind = Token::UNCLOSED_INLINE; t.end = i = n; }
                else { t.kind = Token::PATTERN_INLINE; t.text = s.substr(i + 2, k - i - 2); t.end = i = k + 1; }
            } else {
                t.kind = c == u'%' ? Token::PATTERN_NAME : c == u'@' ? Token::AT : Token::DOLLAR;
                t.name = scan_name(s, i + 1);
                t.end = i = t.name.end;
            }
        } else {
            t.kind = Token::TEXT;
            while (i < n && !is_punct(s[i])) {
                if (is_sigil(s[i])) {
                    if (!(i + 1 < n && s[i + 1] == s[i])) break;
                    ++i;  // a doubled sigil stands for itself
                }
                t.text.push_back(s[i++]);
            }
            t.end = i;
        }
        out.push_back(t);
    }
    Token end;
    end.start = end.at = end.end = n;
    out.push_back(end);
    return out;
}

// ======================================================================================================================
// The tree
// ======================================================================================================================
struct Node;
typedef std::shared_ptr<Node> NodeRef;
struct Node {
    enum Kind { TEXT, PATTERN, PATTERN_REF, TEMPLATE_REF, TEMPLATE_PARAM, EXTRACTOR_PARAM, EXTRACTOR } kind = TEXT;
    LineRef line;
    int at = 0;
    ustr text;                  // the text / the pattern / the referenced name / the extractor's name
    int position = -1;          // parameters, and extractors named by a parameter: 1-based
    bool with_arguments = false;  // TEMPLATE_REF of a parametric template
    std::vector<NodeRef> kids;  // an extractor's contents / a reference's arguments
};
NodeRef new_node(Node::Kind kind, const LineRef& line, int at, const ustr& text) {
    auto n = std::make_shared<Node>();
    n->kind = kind; n->line = line; n->at = at; n->text = text;
    return n;
}

// the parameters a parametric template uses, by position: '@' (a template) or '$' (an extractor name)
struct Signature {
    std::string kinds;
    void use(const LineRef& line, int at, long position, char kind) {
        const size_t slot = static_cast<size_t>(position - 1);
        if (slot >= kinds.size()) kinds.resize(slot + 1, '\0');
        else if (kinds[slot] != '\0' && kinds[slot] != kind)
            fail_at(line, at, "Inconsistent references to parameter " + std::to_string(position) + ": " + kinds[slot] + " vs " + kind);
        kinds[slot] = kind;
    }
};

struct Declaration {     // a pattern, a template, or the template of an extraction
    LineRef line;
    ustr name;
    int body_at = 0;
    bool parametric = false;
    Signature signature;
    std::vector<NodeRef> body;
};
struct ExtractionDecl {
    ustr name;
    Declaration body;
    std::string append;   // canonical JSON object text, or empty
};

// insertion-ordered map in which a repeated key keeps its first slot (what the reference's LinkedHashMaps do)
template <typename V> class Registry {
public:
    bool put(const ustr& key, V value) {  // true: the key was there already
        auto it = slot_.find(key);
        if (it != slot_.end()) { items_[it->second].second = std::move(value); return true; }
        slot_[key] = items_.size();
        items_.push_back({key, std::move(value)});
        return false;
    }
    V* find(const ustr& key) { auto it = slot_.find(key); return it == slot_.end() ? nullptr : &items_[it->second].second; }
    std::vector<std::pair<ustr, V>>& items() { return items_; }
private:
    std::vector<std::pair<ustr, V>> items_;
    std::map<ustr, size_t> slot_;
};

struct Definitions {
    Registry<Declaration> patterns, templates;
    Registry<ExtractionDecl> extractions;
};

// ======================================================================================================================
// Bodies
// ======================================================================================================================
// A pattern's body is a regular expression in which only '%' is special: %name refers to another pattern, %% is a percent.
void parse_pattern_body(Declaration& d) {
    const ustr& s = d.line->text;
    const int n = d.line->size();
    if (s.find(u'%', d.body_at) == ustr::npos) {   // (also when nothing follows the name: one empty piece)
        d.body.push_back(new_node(Node::PATTERN, d.line, d.body_at, s.substr(std::min<size_t>(d.body_at, s.size()))));
        return;
    }
    ustr run;
    auto flush = [&] { if (!run.empty()) { d.body.push_back(new_node(Node::PATTERN, d.line, d.body_at, run)); run.clear(); } };
    for (int i = d.body_at; i < n;) {
        if (s[i] != u'%') { run.push_back(s[i++]); continue; }
        if (i + 1 >= n) fail_at(d.line, n, "Orphan '%' at end of pattern '" + u8(d.name) + "' definition");
        if (s[i + 1] == u'%') { run.push_back(u'%'); i += 2; continue; }
        const Name ref = scan_name(s, i + 1);
        require_name(d.line, ref, "pattern", false);
        flush();
        d.body.push_back(new_node(Node::PATTERN_REF, d.line, i + 1, ref.present() ? ref.text : ustr()));
        i = ref.end;   // (nothing name-like behind the '%': an unnamed reference, and the scan goes on right there)
    }
    flush();
}

class BodyParser {
public:
    // signature: null where positional parameters may not appear (anything but a parametric template's definition)
    BodyParser(Definitions& defs, const LineRef& line, int from, Signature* signature)
        : defs_(defs), line_(line), tokens_(tokenize(line->text, from)), signature_(signature) {}

    // the whole rest of the line: a template's definition, or an extraction's template
    std::vector<NodeRef> parse(const std::string& context) {
        std::vector<NodeRef> out;
        sequence(out, context, false);
        return out;
    }

private:
    const Token& peek() const { return tokens_[next_]; }
    Token take() { return tokens_[next_ == tokens_.size() - 1 ? next_ : next_++]; }

    // Pieces up to the end of the line, or -- inside an extractor -- up to the parenthesis that closes it.  Everything that
    // is not a sigil construct is text: punctuation too, and an extractor counts parentheses so that "f(x)" may stand in it.
    void sequence(std::vector<NodeRef>& out, const std::string& context, bool in_extractor) {
        ustr text;
        int text_at = peek().start, depth = 1;
        auto flush = [&] { if (!text.empty()) { out.push_back(new_node(Node::TEXT, line_, text_at, text)); text.clear(); } };
        for (;;) {
            const Token t = take();
            switch (t.kind) {
            case Token::END:
                flush();
                if (in_extractor) fail_at(line_, t.start, "Missing closing parenthesis at end of " + context);
                return;
            case Token::TEXT:
                text += t.text;
                continue;
            case Token::PUNCT:
                if (in_extractor && t.text[0] == u'(') ++depth;
                if (in_extractor && t.text[0] == u')' && --depth == 0) { flush(); return; }
                text += t.text;
                continue;
            case Token::ORPHAN:
                fail_at(line_, t.end, std::string("Orphan '") + static_cast<char>(t.sigil) + "' at end of " + context);
            case Token::UNCLOSED_INLINE:
                fail_at(line_, t.at, "Missing closing '{' for inline pattern");
            case Token::PATTERN_INLINE:
                flush();
                out.push_back(new_node(Node::PATTERN, line_, t.at, t.text));
                break;
            case Token::PATTERN_NAME:
                require_name(line_, t.name, "pattern", false);
                flush();
                out.push_back(new_node(Node::PATTERN_REF, line_, t.at, t.name.present() ? t.name.text : ustr()));
                break;
            case Token::AT:
                flush();
                out.push_back(template_reference(t, context));
                break;
            case Token::DOLLAR:
                flush();
                out.push_back(extractor(t, context));
                break;
            }
            text_at = peek().start;
        }
    }

    // 1 .. 999999 behind a sigil inside a parametric template's definition
    long positional(const Token& t, char kind, const std::string& what, const std::string& context) {
        const long value = positional_value(t.name.text);
        if (value < 1 || value > 999999) fail_at(line_, t.end, "Invalid " + what + " " + std::to_string(value) + " in " + context);
        signature_->use(line_, t.end, value, kind);
        return value;
    }

    // @3 (inside a parametric template), @name, or @name(arguments) when `name` is parametric
    NodeRef template_reference(const Token& t, const std::string& context) {
        require_name(line_, t.name, "template parameter", signature_ != nullptr);
        if (t.name.state == Name::DIGITS) {
            const long value = positional(t, '@', "template parameter", context);
            NodeRef n = new_node(Node::TEMPLATE_PARAM, line_, t.end, t.name.text);
            n->text = utf8_to_u16(std::to_string(value).c_str());
            n->position = static_cast<int>(value);
            return n;
        }
        Declaration* target = defs_.templates.find(t.name.text);   // (no name at all looks up the empty one)
        if (!target) fail_at(line_, t.end, "Referencing non-existing template '@" + t.name.shown() + "' from '" + context + "'");
        NodeRef ref = new_node(Node::TEMPLATE_REF, line_, t.end, t.name.text);
        if (target->parametric) arguments(*ref, context);
        return ref;
    }

    // "(" [ argument { "," argument } ] ")" right behind the name; an argument is @reference or $extractor-name
    void arguments(Node& ref, const std::string& context) {
        const std::string shown = u8(ref.text);
        const Token open = take();
        if (!(open.kind == Token::PUNCT && open.text[0] == u'(' && open.start == ref.at))
            fail_at(line_, ref.at, "Missing parameter list for template reference '@" + shown + "'");
        auto first_char = [&](const Token& t) { return line_->text[t.start]; };
        // an argument starts with its sigil, doubled or not: a text token that begins with one is read again from there
        auto take_raw = [&]() {
            Token t = take();
            if (t.kind == Token::TEXT && (first_char(t) == u'@' || first_char(t) == u'$')) {
                tokens_ = tokenize(line_->text, t.start, true);
                next_ = 0;
                t = take();
            }
            return t;
        };
        for (int count = 0;; ++count) {
            Token t = take_raw();
            if (t.kind == Token::END) break;
            if (t.kind == Token::PUNCT && t.text[0] == u')') { ref.with_arguments = ref.with_arguments || count > 0; return; }
            if (count > 0) {
                if (!(t.kind == Token::PUNCT && t.text[0] == u','))
                    fail_at(line_, t.start + 1, "Unexpected character " + describe_char(first_char(t)) + " in template parameter list for '@" + shown +
                                                    "': expected either ',' or ')')'");
                t = take_raw();
                if (t.kind == Token::END) break;
            }
            ref.with_arguments = true;
            if (t.kind == Token::ORPHAN && t.sigil != u'%')   // the sigil is the line's last character
                fail_at(line_, line_->size(), std::string("Missing ") + (t.sigil == u'@' ? "template" : "extractor") + " parameter name");
            if (t.kind == Token::AT) ref.kids.push_back(template_reference(t, context));
            else if (t.kind == Token::DOLLAR) ref.kids.push_back(extractor_argument(t, context));
            else
                fail_at(line_, t.start + 1, "Unexpected character " + describe_char(first_char(t)) + " in template parameter list for '@" + shown +
                                                "': expected either type marker '@' or closing ')'");
        }
        fail_at(line_, line_->size(), "Unexpected end of line within parameter list for template '@" + shown + "'");
    }

    // $name or $3 as an argument: the NAME an extractor of the referenced template will carry
    NodeRef extractor_argument(const Token& t, const std::string& context) {
        require_name(line_, t.name, "extractor parameter", signature_ != nullptr);
        if (t.name.state == Name::DIGITS) {
            const long value = positional(t, '$', "extractor parameter", context);
            NodeRef n = new_node(Node::EXTRACTOR_PARAM, line_, t.end, utf8_to_u16(std::to_string(value).c_str()));
            n->position = static_cast<int>(value);
            return n;
        }
        return new_node(Node::EXTRACTOR, line_, t.end, t.name.present() ? t.name.text : ustr());
    }

    // $name(contents), or $3(contents) inside a parametric template: an extractor whose name is the third argument
    NodeRef extractor(const Token& t, const std::string& context) {
        require_name(line_, t.name, "extractor", signature_ != nullptr);
        NodeRef n;
        if (t.name.state == Name::DIGITS) {
            const long value = positional(t, '$', "extractor name parameter", context);
            n = new_node(Node::EXTRACTOR, line_, t.end, utf8_to_u16(std::to_string(value).c_str()));
            n->position = static_cast<int>(value);
        } else n = new_node(Node::EXTRACTOR, line_, t.end, t.name.present() ? t.name.text : ustr());
        const Token open = take();
        if (!(open.kind == Token::PUNCT && open.text[0] == u'(' && open.start == t.end))
            fail_at(line_, t.end, "Invalid declaration for extractor '" + u8(n->text) + "': missing opening parenthesis");
        sequence(n->kids, "extractor '" + u8(n->text) + "' expression", true);
        return n;
    }

    Definitions& defs_;
    LineRef line_;
    std::vector<Token> tokens_;
    size_t next_ = 0;
    Signature* signature_;
};

// ======================================================================================================================
// Declarations: one logical line each (an extraction: a block of them)
// ======================================================================================================================
class DeclarationReader {
public:
    DeclarationReader(const ustr& text, const std::string& source) : lines_(text, source) {}

    Definitions read() {
        while (LineRef line = lines_.next()) {
            // keyword = the first run of word characters; the declaration follows it
            const ustr& s = line->text;
            auto is_space = [](char16_t c) { return c == u' ' || (c >= 9 && c <= 13); };
            auto is_word = [](char16_t c) { return is_digit(c) || (c >= u'a' && c <= u'z') || (c >= u'A' && c <= u'Z') || c == u'_'; };
            int b = 0, n = line->size();
            while (b < n && is_space(s[b])) ++b;
            int e = b;
            while (e < n && is_word(s[e])) ++e;
            int rest = e;
            while (rest < n && is_space(s[rest])) ++rest;
            const std::string keyword = u8(s.substr(b, e - b));
            if (keyword == "pattern") pattern(line, rest);
            else if (keyword == "template") templ(line, rest);
            else if (keyword == "extract") extraction(line, rest);
            else fail_at(line, 0, "Unrecognized keyword \"" + keyword + "\" encountered; expected one of (pattern, template, extract)");
        }
        // bodies are read once every name is known: a template may be used above its declaration
        for (auto& p : defs_.patterns.items()) parse_pattern_body(p.second);
        for (auto& t : defs_.templates.items()) {
            Declaration& d = t.second;
            d.body = BodyParser(defs_, d.line, d.body_at, d.parametric ? &d.signature : nullptr).parse("template '" + u8(d.name) + "' definition");
        }
        for (auto& x : defs_.extractions.items()) {
            Declaration& d = x.second.body;
            d.body = BodyParser(defs_, d.line, d.body_at, nullptr).parse("extraction template for '" + u8(d.name) + "'");
        }
        return std::move(defs_);
    }

private:
    // the sigil of a declared name: the first one before any blank
    static int find_sigil(const ustr& s, int from, char16_t sigil) {
        for (int i = from; i < static_cast<int>(s.size()) && !is_blank(s[i]); ++i) if (s[i] == sigil) return i;
        return -1;
    }
    // "only `brace` (and blanks) from here on": the position where that stops being true, -1 when there is no brace.
    // (A second brace ends the scan, so "{{" at the very end passes -- as it does in the reference.)
    static int only_brace(const ustr& s, int from, char16_t brace) {
        bool seen = false;
        int i = from;
        while (i < static_cast<int>(s.size())) {
            const char16_t c = s[i++];
            if (c == brace) { if (seen) break; seen = true; }
            else if (!is_blank(c)) break;
        }
        return seen ? i : -1;
    }

    void pattern(const LineRef& line, int from) {
        const int sigil = find_sigil(line->text, from, u'%');
        if (sigil < 0) fail_at(line, from, "Pattern name must be prefixed with '%'");
        const Name name = scan_name(line->text, sigil + 1);
        require_name(line, name, "pattern", false);
        Declaration d;
        d.line = line;
        d.name = name.text;
        d.body_at = end_of_declared_name(line, name, "pattern");
        if (defs_.patterns.put(d.name, d)) fail_at(line, sigil + 1, "Duplicate pattern definition for name '" + u8(d.name) + "'");
    }

    void templ(const LineRef& line, int from) {
        const ustr& s = line->text;
        const int sigil = find_sigil(s, from, u'@');
        if (sigil < 0) fail_at(line, from, "Template name must be prefixed with '@'");
        const Name name = scan_name(s, sigil + 1);
        require_name(line, name, "template", false);
        Declaration d;
        d.line = line;
        d.name = name.text;
        int at = name.end;
        if (at + 1 < line->size() && s[at] == u'(' && s[at + 1] == u')') { d.parametric = true; at += 2; }   // "()": takes parameters
        d.body_at = skip_blanks(s, at);
        if (d.body_at == at) fail_at(line, at, "Missing space character after template name '" + u8(d.name) + "'");
        if (defs_.templates.put(d.name, d)) fail_at(line, sigil + 1, "Duplicate template definition for name '" + u8(d.name) + "'");
    }

    // extract Name {  /  template ...  /  append ...  /  }
    void extraction(LineRef line, int from) {
        const Name name = scan_name(line->text, from);
        require_name(line, name, "extraction", false);
        const int rest = end_of_declared_name(line, name, "extraction");
        if (only_brace(line->text, rest, u'{') != line->size())
            fail_at(line, rest, "Unexpected content for extraction '" + name.shown() + "': expected only opening '{'");
        ExtractionDecl x;
        x.name = name.text;
        bool has_template = false;
        int at = rest;
        for (;;) {
            line = lines_.next();
            if (!line) lines_.fail_here("Unexpected end-of-input in extraction '" + name.shown() + "' definition");
            const ustr& s = line->text;
            const int closed = only_brace(s, 0, u'}');
            if (closed >= 0) {
                if (closed >= line->size()) { at = closed; break; }
                fail_at(line, at, "Unexpected content after closing '}' for extraction '" + name.shown() + "'");
            }
            const Name property = scan_name(s, skip_blanks(s, 0));
            require_name(line, property, "extraction", false);
            at = end_of_declared_name(line, property, "extraction");
            const std::string p = property.shown();
            if (p == "template") {
                if (has_template) fail_at(line, at, "More than one 'template' specified for '" + name.shown() + "'");
                has_template = true;
                x.body.line = line;
                x.body.body_at = at;
            } else if (p == "append") {
                x.append = append(line, at, x.append);
            } else
                fail_at(line, at, "Unrecognized extraction property \"" + p + "\" encountered; expected one of (template, append)");
        }
        if (!has_template) fail_at(line, at, "Missing 'template' for extraction '" + name.shown() + "'");
        defs_.extractions.put(x.name, x);   // a repeated name replaces the earlier extraction, in its slot
    }

    // append <JSON object | "key": value, ...>: merged into what earlier append lines of the extraction gave
    std::string append(const LineRef& line, int at, const std::string& so_far) {
        const ustr& s = line->text;
        std::string canonical, other;
        try {
            canonical = parse_append_value(u8(s.substr(std::min<size_t>(at, s.size()))), &other);
        } catch (GxError&) {
            throw;
        } catch (std::runtime_error& e) {
            fail_at(line, at, std::string("Invalid JSON content to 'append': ") + e.what());
        }
        if (!other.empty()) fail_at(line, at, "Invalid 'append' value: must be JSON Object, or sequence of key/value pairs; was parsed as " + other);
        return canonical.empty() ? so_far : json_merge(so_far, canonical);
    }

    LineSource lines_;
    Definitions defs_;
};

// ======================================================================================================================
// Linking
// ======================================================================================================================
class Linker {
public:
    explicit Linker(Definitions& defs) : defs_(defs) {}

    // ---- patterns: every %reference replaced by the referenced pattern's text --------------------------------------
    void link_patterns() {
        for (auto& p : defs_.patterns.items()) pattern(p.first, nullptr);
    }
    const NodeRef* linked_pattern(const ustr& name) const {
        auto it = pattern_text_.find(name);
        return it == pattern_text_.end() ? nullptr : &it->second;
    }

    // ---- templates: pattern references replaced, references to plain templates inlined -------------------------------
    void link_templates() {
        for (auto& t : defs_.templates.items()) {
            if (template_parts_.count(t.first)) continue;   // (frozen empty by a reference from above its declaration)
            std::vector<NodeRef> parts;
            inline_into(parts, t.second.body, t.first, t.first, "");
            template_parts_[t.first] = std::move(parts);
        }
    }
    const std::vector<NodeRef>& linked_template(const ustr& name) { return template_parts_[name]; }

    // ---- extractions: parametric references expanded, extractor names collected ----------------------------------------
    std::vector<Extraction> link_extractions() {
        std::vector<Extraction> out;
        for (auto& x : defs_.extractions.items()) {
            std::vector<NodeRef> parts;
            inline_into(parts, x.second.body.body, x.second.body.name, x.second.body.name, "");
            Extraction flat;
            flat.name = u8(x.second.name);
            flat.append_json = x.second.append;
            Expansion run;
            std::vector<NodeRef> pieces;
            expand(parts, pieces, run, nullptr, true);
            for (auto& n : run.names) flat.extractor_names.push_back(u8(n));
            for (auto& p : pieces) flat.pieces.push_back(to_piece(p));
            out.push_back(std::move(flat));
        }
        return out;
    }

private:
    // Depth-first over the pattern reference graph.  WHITE: not seen; GREY: on the current path (meeting one again is a
    // cycle); BLACK: text known.  `via` is the reference that led here (null for a declaration visited on its own).
    enum Colour { WHITE, GREY, BLACK };
    NodeRef pattern(const ustr& name, const NodeRef& via) {
        Colour& colour = pattern_colour_[name];
        if (colour == BLACK) return pattern_text_[name];
        if (colour == GREY || !defs_.patterns.find(name)) {
            std::string path = "(";
            for (auto& on_path : path_) path += "%" + u8(on_path) + "->";
            path += "%" + u8(name) + ")";
            fail_at(via->line, via->at, std::string(colour == GREY ? "Cyclic pattern reference to" : "Referencing non-existing pattern") + " '%" + u8(name) + "' " + path);
        }
        colour = GREY;
        path_.push_back(name);
        const Declaration& d = *defs_.patterns.find(name);
        NodeRef result;
        if (d.body.size() == 1 && d.body[0]->kind == Node::PATTERN) result = d.body[0];
        else {
            ustr joined;
            for (auto& piece : d.body) joined += piece->kind == Node::PATTERN ? piece->text : pattern(piece->text, piece)->text;
            result = d.body.size() == 1 ? pattern(d.body[0]->text, d.body[0]) : new_node(Node::PATTERN, d.line, d.body.empty() ? 0 : d.body[0]->at, joined);
        }
        path_.pop_back();
        pattern_colour_[name] = BLACK;
        pattern_text_[name] = result;
        return result;
    }

    // `from`: the template being linked (a reference back to it is a cycle); `top`: the name the messages quote;
    // `inside`: "" at a template's top level, else "(@extractor)" -- the trail the messages carry
    void inline_into(std::vector<NodeRef>& out, const std::vector<NodeRef>& body, const ustr& from, const ustr& top, const std::string& inside) {
        for (const NodeRef& n : body) {
            switch (n->kind) {
            case Node::TEXT: case Node::PATTERN: case Node::TEMPLATE_PARAM:
                out.push_back(n);
                break;
            case Node::PATTERN_REF: {
                const NodeRef* text = linked_pattern(n->text);
                if (!text) fail_at(n->line, n->at, "Referencing non-existing pattern '%" + u8(n->text) + "' from template '" + u8(top) + "' " + inside);
                out.push_back(*text);
                break;
            }
            case Node::TEMPLATE_REF:
                if (n->with_arguments) { out.push_back(n); break; }   // expanded per extraction, with its bindings
                if (!template_parts_.count(n->text)) {
                    // not linked yet: itself (a cycle), or declared further down -- which freezes it as EMPTY, here and
                    // for everything after (what the reference does; kept, quirk and all)
                    const std::string trail = "(@" + u8(from) + "->@" + u8(n->text) + ")";
                    if (n->text == from) fail_at(n->line, n->at, "Cyclic template reference to '%" + u8(n->text) + "' " + trail);
                    if (!defs_.templates.find(n->text)) fail_at(n->line, n->at, "Referencing non-existing template '%" + u8(n->text) + "' " + trail);
                    template_parts_[n->text] = {};
                }
                for (auto& p : template_parts_[n->text]) out.push_back(p);
                break;
            case Node::EXTRACTOR: {
                NodeRef copy = new_node(Node::EXTRACTOR, n->line, n->at, n->text);
                copy->position = n->position;
                inline_into(copy->kids, n->kids, from, top, "(@" + u8(n->text) + ")");
                out.push_back(copy);
                break;
            }
            default:
                fail_at(n->line, 0, "Internal error: unexpected definition type when resolving template definition '" + u8(top) + "'");
            }
        }
    }

    // what a parametric template's parameters are bound to, for one reference to it
    struct Bindings {
        std::string kinds;
        std::vector<NodeRef> values;
        NodeRef at(int position) const { return position >= 1 && position <= static_cast<int>(values.size()) ? values[position - 1] : nullptr; }
    };
    struct Expansion { std::set<ustr> seen; std::vector<ustr> names; };   // extractor names, in pre-order

    void expand(const std::vector<NodeRef>& in, std::vector<NodeRef>& out, Expansion& run, const Bindings* bound, bool top_level) {
        for (NodeRef n : in) {
            if (n->kind == Node::TEMPLATE_PARAM) {
                const std::string shown = std::to_string(n->position);
                if (top_level) fail_at(n->line, n->at, "Internal error: should not encounter template parameter #" + shown);
                if (!bound) fail_at(n->line, n->at, "Invalid parameter variable reference @" + shown + "; template takes no parameters");
                NodeRef value = bound->at(n->position);
                if (!value) fail_at(n->line, n->at, "Invalid parameter variable reference @" + shown + "; template takes " + std::to_string(bound->kinds.size()) + " parameters");
                n = value;
            }
            switch (n->kind) {
            case Node::TEXT: case Node::PATTERN:
                out.push_back(n);
                break;
            case Node::PATTERN_REF: {
                const NodeRef* text = linked_pattern(n->text);
                if (!text) throw GxError(GX_E_DEFINITION, "Internal error: non-existing pattern '%" + u8(n->text) + "', should have been caught earlier");
                out.push_back(*text);
                break;
            }
            case Node::EXTRACTOR: {
                ustr name = n->text;
                if (n->position >= 0) {   // named by a parameter: the argument must be an extractor name
                    NodeRef value = bound ? bound->at(n->position) : nullptr;
                    if (!value || value->kind != Node::EXTRACTOR) fail_at(n->line, n->at, "Internal error: unexpected extractor parameter (expecting ExtractorExpression)");
                    if (value->position >= 0)
                        fail_at(n->line, n->at, "Internal error: positional extractor parameter (" + std::to_string(n->position) + ") resolves to another positional (" +
                                                    std::to_string(value->position) + ")");
                    name = value->text;
                }
                if (!run.seen.insert(name).second) fail_at(n->line, n->at, "Duplicate extractor name ($" + u8(name) + ")");
                run.names.push_back(name);
                NodeRef copy = new_node(Node::EXTRACTOR, n->line, n->at, name);
                expand(n->kids, copy->kids, run, bound, false);
                out.push_back(copy);
                break;
            }
            case Node::TEMPLATE_REF:
                expand_reference(*n, out, run, bound);
                break;
            default:
                fail_at(n->line, n->at, "Internal error: unrecognized DefPiece");
            }
        }
    }

    void expand_reference(const Node& ref, std::vector<NodeRef>& out, Expansion& run, const Bindings* incoming) {
        Declaration* target = defs_.templates.find(ref.text);
        if (!target || !template_parts_.count(ref.text)) fail_at(ref.line, ref.at, "Internal error: reference to unknown template '@" + u8(ref.text) + "'");
        if (!target->parametric) { expand(template_parts_[ref.text], out, run, nullptr, false); return; }
        Bindings bound;
        bound.kinds = target->signature.kinds;
        if (ref.kids.size() != bound.kinds.size())
            fail_at(ref.line, ref.at, "Parameter mismatch: template '@" + u8(ref.text) + "' expects " + std::to_string(bound.kinds.size()) + " parameters; " +
                                          std::to_string(ref.kids.size()) + " passed");
        for (size_t i = 0; i < ref.kids.size(); ++i) {
            const char want = bound.kinds[i];
            if (want != '@' && want != '$') throw GxError(GX_E_DEFINITION, "Internal error: unrecognized template parameter type");
            const Node::Kind need = want == '@' ? Node::TEMPLATE_REF : Node::EXTRACTOR;
            if (ref.kids[i]->kind != need)
                fail_at(ref.line, ref.at, std::string("Parameter mismatch: template '@") + u8(ref.text) + "' expects type '" + want + "' parameter");
            bound.values.push_back(substitute(ref.kids[i], incoming));
        }
        expand(template_parts_[ref.text], out, run, &bound, false);
    }

    // an argument in terms of the caller's own bindings (arguments may themselves mention the caller's parameters)
    NodeRef substitute(const NodeRef& n, const Bindings* bound) {
        switch (n->kind) {
        case Node::TEMPLATE_PARAM: case Node::EXTRACTOR_PARAM: {
            NodeRef value = bound ? bound->at(n->position) : nullptr;
            if (!value)
                fail_at(n->line, n->at, "Invalid parameter variable reference @" + std::to_string(n->position) + "; template has " +
                                            std::to_string(bound ? bound->kinds.size() : 0) + " parameters");
            return value;
        }
        case Node::TEMPLATE_REF: {
            if (!n->with_arguments) return n;
            NodeRef copy = new_node(Node::TEMPLATE_REF, n->line, n->at, n->text);
            copy->with_arguments = true;
            for (auto& k : n->kids) copy->kids.push_back(substitute(k, bound));
            return copy;
        }
        case Node::EXTRACTOR: {
            NodeRef copy = new_node(Node::EXTRACTOR, n->line, n->at, n->text);
            copy->position = n->position;
            for (auto& k : n->kids) copy->kids.push_back(substitute(k, bound));
            return copy;
        }
        default:
            fail_at(n->line, n->at, "Internal error: unexpected template parameter type");
        }
    }

    static Piece to_piece(const NodeRef& n) {
        Piece p;
        p.kind = n->kind == Node::TEXT ? Piece::TEXT : n->kind == Node::PATTERN ? Piece::PATTERN : Piece::EXTRACTOR;
        p.text = u8(n->text);
        for (auto& k : n->kids) p.kids.push_back(to_piece(k));
        return p;
    }

    Definitions& defs_;
    std::map<ustr, Colour> pattern_colour_;
    std::map<ustr, NodeRef> pattern_text_;
    std::vector<ustr> path_;
    std::map<ustr, std::vector<NodeRef>> template_parts_;
};

[[noreturn]] void fail_no_extractions() { throw GxError(GX_E_DEFINITION, "(N/A): No extraction definitions found from definition"); }

}  // namespace

std::vector<Extraction> read_definition(const std::string& utf8_text, const std::string& source_ref) {
    Definitions defs = DeclarationReader(utf8_to_u16(utf8_text.c_str()), source_ref).read();
    if (defs.extractions.items().empty()) fail_no_extractions();
    Linker linker(defs);
    linker.link_patterns();
    linker.link_templates();
    return linker.link_extractions();
}

// The two regular expressions of one flattened extraction: what Gorp._buildExtractor assembles (core/Gorp.java:94-129).
void build_regex_strings(const Extraction& x, std::string& automaton_rx, std::string& jdk_rx) {
    ustr a, j;
    std::function<void(const Piece&)> walk = [&](const Piece& p) {
        switch (p.kind) {
        case Piece::PATTERN: {
            const ustr t = utf8_to_u16(p.text.c_str());
            try {
                a += massage_regexp_for_automaton(t);
                j += massage_regexp_for_jdk(t);
            } catch (GxError& e) {
                throw GxError(GX_E_DEFINITION, std::string("Invalid pattern definition, problem (java.lang.IllegalArgumentException): ") + e.what());
            }
            break;
        }
        case Piece::TEXT: {
            const ustr q = quote_literal_as_regexp(utf8_to_u16(p.text.c_str()));
            a += q;
            j += q;
            break;
        }
        case Piece::EXTRACTOR:
            a += u'('; j += u'(';
            for (auto& k : p.kids) walk(k);
            a += u')'; j += u')';
            break;
        }
    };
    for (auto& p : x.pieces) walk(p);
    automaton_rx = u16_to_utf8(a);
    jdk_rx = u16_to_utf8(j);
}

// ======================================================================================================================
// JSON views of the intermediate stages (test support: the reference's DSL unit tests look at these structures)
// ======================================================================================================================
namespace {
std::string jstr(const std::string& s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o += static_cast<char>(c);
        }
    }
    return o + "\"";
}
std::string dump_nodes(const std::vector<NodeRef>& nodes);
std::string dump_node(const NodeRef& n) {
    static const char* class_of[] = {"LiteralText", "LiteralPattern", "PatternReference", "TemplateReference", "TemplateParameterReference",
                                     "ExtractorParameterReference", "ExtractorExpression"};
    std::string o = "{\"class\":" + jstr(class_of[n->kind]) + ",\"text\":" + jstr(u8(n->text));
    if (n->position >= 0) o += ",\"position\":" + std::to_string(n->position);
    if (n->kind == Node::TEMPLATE_REF) o += std::string(",\"takesParameters\":") + (n->with_arguments ? "true" : "false");
    if (!n->kids.empty() || n->kind == Node::EXTRACTOR) o += ",\"parts\":" + dump_nodes(n->kids);
    return o + "}";
}
std::string dump_nodes(const std::vector<NodeRef>& nodes) {
    std::string o = "[";
    for (size_t i = 0; i < nodes.size(); ++i) o += (i ? "," : "") + dump_node(nodes[i]);
    return o + "]";
}
std::string dump_piece(const Piece& p) {
    static const char* names[] = {"text", "pattern", "extractor"};
    std::string o = std::string("[") + jstr(names[p.kind]) + "," + jstr(p.text);
    if (p.kind == Piece::EXTRACTOR) {
        o += ",[";
        for (size_t i = 0; i < p.kids.size(); ++i) o += (i ? "," : "") + dump_piece(p.kids[i]);
        o += "]";
    }
    return o + "]";
}
template <typename Items, typename F> std::string dump_object(Items& items, F value) {
    std::string o = "{";
    bool first = true;
    for (auto& item : items) { o += (first ? "" : ",") + jstr(u8(item.first)) + ":" + value(item); first = false; }
    return o + "}";
}
}  // namespace

std::string dump_json(const std::string& utf8_text, const std::string& source_ref, const std::string& stage) {
    const ustr text = utf8_to_u16(utf8_text.c_str());
    if (stage == "lines") {  // the logical lines alone, with the rows they came from
        LineSource source(text, source_ref);
        std::string o = "{\"lines\":[";
        bool first = true;
        while (LineRef l = source.next()) {
            o += (first ? "" : ",") + std::string("{\"row\":") + std::to_string(l->first_row()) + ",\"rows\":" + std::to_string(l->segments.size()) +
                 ",\"contents\":" + jstr(u8(l->text)) + "}";
            first = false;
        }
        return o + "]}";
    }
    Definitions defs = DeclarationReader(text, source_ref).read();
    if (stage == "uncooked") {
        return "{\"patterns\":" + dump_object(defs.patterns.items(), [](auto& p) { return dump_nodes(p.second.body); }) +
               ",\"templates\":" + dump_object(defs.templates.items(), [](auto& t) {
                   return std::string("{\"hasParameters\":") + (t.second.parametric ? "true" : "false") + ",\"parameterTypes\":" +
                          jstr(t.second.parametric ? std::string(t.second.signature.kinds.c_str()) : "") + ",\"parts\":" + dump_nodes(t.second.body) + "}";
               }) +
               ",\"extractions\":" + dump_object(defs.extractions.items(), [](auto& x) {
                   return "{\"template\":" + dump_nodes(x.second.body.body) + ",\"append\":" + (x.second.append.empty() ? "null" : x.second.append) + "}";
               }) + "}";
    }
    Linker linker(defs);
    linker.link_patterns();
    if (stage == "cooked") {
        linker.link_templates();
        return "{\"patterns\":" + dump_object(defs.patterns.items(), [&](auto& p) { return jstr(u8((*linker.linked_pattern(p.first))->text)); }) +
               ",\"templates\":" + dump_object(defs.templates.items(), [&](auto& t) { return dump_nodes(linker.linked_template(t.first)); }) + "}";
    }
    if (defs.extractions.items().empty()) fail_no_extractions();
    linker.link_templates();
    std::string o = "{\"extractions\":[";
    bool first = true;
    for (const Extraction& x : linker.link_extractions()) {
        std::string a, j;
        build_regex_strings(x, a, j);
        o += (first ? "" : ",") + std::string("{\"name\":") + jstr(x.name) + ",\"pieces\":[";
        for (size_t k = 0; k < x.pieces.size(); ++k) o += (k ? "," : "") + dump_piece(x.pieces[k]);
        o += "],\"extractor_names\":[";
        for (size_t k = 0; k < x.extractor_names.size(); ++k) o += (k ? "," : "") + jstr(x.extractor_names[k]);
        o += "],\"append\":" + (x.append_json.empty() ? std::string("null") : x.append_json) + ",\"automaton_rx\":" + jstr(a) + ",\"jdk_rx\":" + jstr(j) + "}";
        first = false;
    }
    return o + "]}";
}

}  // namespace dsl
}  // namespace gx
