/*
 * gorp_hip.h -- C ABI of libgorp_hip.so: the MI355X (gfx950) implementation of
 * Gorp's combined-DFA match-and-extract hot path.
 *
 * Every entry point cites the reference interface it replaces
 * (core/ = gorp-core/src/main/java/com/salesforce/gorp/ in salesforce/gorp).
 * Plain pointers and sizes only; no exceptions cross this boundary; no torch
 * types.  INTEGRATION.md shows the JNI stub that binds these from Java.
 *
 * Per-line results are DATA, never call failures:
 *   match_id >= 0      index of the first-declared matching extraction
 *                      (matchIndexes[0], core/Gorp.java:166)
 *   match_id == -1     no extraction matches -> Gorp.extract returns null
 *                      (core/Gorp.java:162-164)
 *   match_id == -2-k   the DFA chose extraction k but its capture regex
 *                      rejected the line -> ExtractionException
 *                      (core/Gorp.java:173-177); extractSafe returns null
 *                      for exactly these lines (core/Gorp.java:178-185)
 * Captures: for g < gx_num_groups(h, match_id): caps[2g], caps[2g+1] are the
 * begin/end offsets of Matcher.group(g+1) in code units from the start of the
 * line (core/jdkre/JDKRegexpCookedExtraction.java:51-59); -1,-1 when group()
 * would return null.  Slots beyond the matched extraction's group count are -1.
 */
#ifndef GORP_HIP_H
#define GORP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gx_handle gx_handle;

/* Error classes (return values; 0 = OK).  Reference conventions they stand for:
 * invalid regex -> IllegalArgumentException "Invalid regexp, ..." wrapped in
 * DefinitionParseException (core/autom/PolyMatcher.java:79-81, core/Gorp.java:84-90). */
enum {
    GX_OK = 0,
    GX_E_REGEX_SYNTAX = 1,         /* a pattern does not parse in its dialect */
    GX_E_UNSUPPORTED_CONSTRUCT = 2,/* valid java.util.regex, outside Gorp's documented subset (README.md:209-224) */
    GX_E_DEVICE = 3,               /* no gfx950 device / HIP runtime failure */
    GX_E_ARG = 4,                  /* bad argument */
    GX_E_NOMEM = 5,
    GX_E_LIMIT = 6,                /* automaton exceeds a compile-time limit */
    GX_E_DEFINITION = 7            /* definition text rejected: DefinitionParseException (core/DefinitionParseException.java) */
};

/* gx_create flags */
#define GX_CREATE_HOST_ONLY 1u     /* compile tables only; do not touch the GPU (used to build the
                                      blob that is broadcast to other ranks, and by CPU-only checks) */
/* Kernel choice, for measurements and tests; results never depend on it.  By default the automaton rows live in
 * LDS when they fit, else in global memory (L2), and the capture automata are fused with the match automaton
 * when that product stays within the size limits. */
#define GX_CREATE_TIER_L2   2u     /* keep the automaton rows in global memory even when they would fit LDS */
#define GX_CREATE_NO_TILES  4u     /* per-line kernel only */
#define GX_CREATE_NO_FUSED  8u     /* two passes: match automaton, then the winning extraction's capture automaton */
#define GX_CREATE_TIER_RECORDS 16u /* sparse range records in LDS even when the dense rows would fit (the default when they do
                                      not: BASELINE configs[2], 64 extractions) */
#define GX_CREATE_TIER_RECORDS_GLOBAL 32u /* sparse range records in global memory (L1 / L2 resident) */
#define GX_CREATE_TIER_HOP  64u    /* build the hop tier's tables (run + literal chain per state, hot states in LDS, dense rows in
                                      global memory as the backstop) even when the dense rows fit LDS; the default for capture
                                      batches when they do not */

#define GX_CREATE_RESIDENT_ONE 128u /* gx_extract_one_utf16 without a kernel launch per call: while such calls keep coming the handle keeps
                                      one wave resident on the device (tables in LDS) that takes the line out of pinned host memory and
                                      puts the answer back -- a few stores and a spin on the host's side, about 5 us instead of 22.  The wave
                                      leaves by itself when no call has come for 0.3 ms (the next call starts it again, at the price of
                                      a launch) and after 20 ms in any case, so a device-wide synchronisation elsewhere in the process
                                      waits milliseconds at most.  For definitions whose dense rows fit LDS and Latin-1 lines of up to
                                      1 016 characters; every other call takes the usual path.  Off by default: it holds LDS and a
                                      wave slot of one CU while it is resident. */

/* Replaces Gorp.construct's per-extraction back half (core/Gorp.java:58-92):
 * PolyMatcher.create(automatonInputs) (core/autom/PolyMatcher.java:64-84 ->
 * Automata.construct, core/autom/Automata.java:57-124) and
 * ExtractionCooker.cook -> Pattern.compile (core/jdkre/JDKRegexpExtractionCooker.java:20-26).
 * Inputs are exactly the two regex strings per extraction that
 * Gorp._buildExtractor emits (core/Gorp.java:94-129), UTF-8 encoded.
 * jdk_rx may be NULL: matcher only (PolyMatcher.create), no captures.
 * The group count of extraction k is read from jdk_rx[k] itself
 * (Matcher.groupCount()). */
int gx_create_from_patterns(const char* const* automaton_rx, const char* const* jdk_rx,
                            int32_t n, uint32_t flags, gx_handle** out);

/* Packed, relocatable table blob (the RCCL broadcast payload): rank 0 compiles,
 * every rank calls gx_create_from_blob.  gx_blob_size returns the byte count;
 * gx_blob_copy writes it to dst. */
size_t gx_blob_size(const gx_handle* h);
int gx_blob_copy(const gx_handle* h, void* dst, size_t cap);
int gx_create_from_blob(const void* blob, size_t size, uint32_t flags, gx_handle** out);

/* Frees host and device tables.  (Java GC in the reference.) */
void gx_destroy(gx_handle* h);

/* Introspection (Automata.size(), core/autom/Automata.java:129-131; Matcher.groupCount()). */
int32_t gx_num_extractions(const gx_handle* h);
int32_t gx_num_groups(const gx_handle* h, int32_t k);
int32_t gx_max_groups(const gx_handle* h);
/* table statistics: 0 = match-DFA states, 1 = char classes, 2 = capture-automaton states (sum),
 * 3 = capture registers (max over extractions), 4 = blob bytes, 5 = LDS bytes the batch kernel stages,
 * 6 = waves per workgroup of the batch kernel, 7 = table tier of the batch kernel (1 = automaton rows in LDS,
 * 2 = rows in global memory / L2, 3 = sparse range records in LDS, 4 = range records in global memory, 0 = per-line
 * generic kernel), 9 = the same for match-only batches (a large definition keeps a second, smaller table image for them),
 * 8 = 1 when the handle has capture regexps, 10 / 11 = waves per workgroup of the lane kernel (captures with compact rows /
 * match only; 0 where it does not apply), 12 = bytes of the batch kernels' table image, 13 = bytes of a wave's register block;
 * the hop tier (run + chain records for capture batches of definitions whose dense rows do not fit LDS): 14 = its states
 * (0: the handle has no hop tables), 15 / 21 = states whose records are in LDS under the tile kernel / the hop slice kernel,
 * 16 = states that well-formed lines reach, 17 = states that have a chain, 18 / 19 = waves per workgroup of the tile kernel on
 * these tables / of the hop slice kernel, 20 = branching states whose dense row is in LDS too, 22 / 23 = states / states with
 * their records in LDS of the second hop image, built from the match automaton alone for match-only batches;
 * 24 = batches so far that broke their gx_batch_opts.max_line_bytes promise; 25 = the kernel the most recent batch ran on
 * (a GX_KERNEL_* value; 0: none yet); 26 = why capture batches have no hop tables (0: they have; 1: no fused automaton or no
 * capture regexps; 2: a step with capture programs other than one "register := position" -- groups that may match the empty string
 * write two registers in one step; 3: beyond a limit of the tier; 4: not built -- the dense rows fit LDS, or the caller named another
 * tier; 5: the tables leave no room for a wave); 27 = extractions whose capture automaton would be too large ahead of time and
 * whose regexp is therefore RUN as a program, thread lists in priority order (exact, linear in line x program; such a definition's
 * batches go through the per-line kernel); 28 = times the resident one-line wave was started (GX_CREATE_RESIDENT_ONE; -1: the
 * handle has none) */
int64_t gx_stat(const gx_handle* h, int32_t which);

typedef struct gx_batch_opts {
    uint32_t struct_size;      /* = sizeof(gx_batch_opts) */
    uint32_t device_pointers;  /* 1: bytes/offsets/match_id/caps are device pointers on the handle's device */
    uint32_t offsets64;        /* 1: offsets are uint64_t[n+1] instead of uint32_t[n+1] */
    uint32_t match_only;       /* 1: PolyMatcher.match only; caps may be NULL */
    void*    stream;           /* hipStream_t to launch on (NULL = the null stream) */
    uint32_t no_sync;          /* 1 (device pointers only): return after enqueueing */
    uint32_t line_bytes_hint;  /* typical line length in bytes; sizes the per-wave LDS staging area of the batch kernel.
                                  A wrong hint costs speed, never correctness.  0: the batch's mean line length -- read
                                  from the offsets (a small synchronous copy), or with no_sync the mean of the previous
                                  no_sync batch of this handle (200 until one has completed) */
    uint32_t strip_eol;        /* 1: every line carries its terminator ("\n", "\r\n" or "\r", as produced by
                                  gx_split_lines); it is not part of the String the reference would see, so
                                  it is ignored and capture offsets stay relative to the start of the line */
    uint32_t utf8_passthrough; /* gx_results_to_jsonl only.  0: line bytes are Latin-1 code units (the batch path's input
                                  model) and bytes >= 0x80 leave as two-byte UTF-8; 1: copy them unchanged (the
                                  input was UTF-8 all along and the patterns only look at its ASCII structure) */
    uint32_t utf16;            /* gx_extract_batch only.  1: `bytes` holds UTF-16 code units (uint16_t, host byte order), exactly
                                  the chars of the Java Strings, and offsets count code units.  On dense rows in LDS and on
                                  hop tables (gx_stat(h, 7) == 1 or gx_stat(h, 14) > 0, kernel AUTO) the batch kernels read
                                  the units themselves: no copy, no synchronisation, no_sync means what it says.  On the
                                  other tables, or with a kernel named in `kernel`, the units' low bytes go through the byte
                                  kernels as a narrowed copy, sized by a read of the offsets' two ends ON THE HOST: such a
                                  batch cannot be no_sync and is refused with GX_E_ARG when it asks for it (round 5; until
                                  then the flag was silently not honoured there).  Either way only the lines that hold a
                                  unit above 0xFF are walked again, per line, on the code units.  (With compact rows, an
                                  offset that does not fit is counted once per walk: such a line is walked twice.) */
    uint32_t kernel;           /* gx_extract_batch only: GX_KERNEL_AUTO (0) or one of the kernels below, for measurements and
                                  tests; results never depend on it.  (New fields are only ever appended: a caller compiled
                                  against an older, shorter layout passes its own struct_size and keeps working.) */
    uint32_t compact_results;  /* gx_extract_batch only.  1: results leave as compact rows -- `caps` points to
                                  uint16_t[n * (1 + 2*gx_max_groups(h))], per line the match id as int16 followed by the capture
                                  offsets, 0xFFFF = unset (the layout of gx_pack_results); match_id may be NULL.  Half the
                                  result bytes of the dense format; what the gather between GPUs sends.  An offset above
                                  65534 does not fit: it is stored as 65534 and counted in *overflow (take such a batch again
                                  in the dense format).  2: u8 rows instead -- `caps` points to uint8_t[n * (1 + 2*gx_max_groups(h))],
                                  the match id as int8 and the offsets with 0xFF = unset: a quarter of the dense bytes, for
                                  batches whose lines are shorter than 255 bytes (log lines mostly are) and definitions of at
                                  most 126 extractions (GX_E_ARG beyond).  An offset above 254 is stored as 254 and counted in
                                  *overflow.  Ignored with match_only. */
    uint32_t uneven_lines;     /* gx_extract_batch only.  Lines run in lock step in groups of 64: a group takes as long as its longest
                                  line.  2: the lines differ much in length -- the kernels then group lines of similar length where
                                  they can (results are the same); 1: they do not; 0: the library looks itself where it can see
                                  the offsets without waiting (host pointers; device pointers without no_sync and without a hint),
                                  and assumes 1 elsewhere.  Batches with a mean length above 255 bytes are taken as uneven. */
    void*    overflow;         /* with compact_results: uint64_t counter that the call ADDS to (the caller zeroes it); a device
                                  pointer with device_pointers, else a host pointer.  NULL: not counted. */
    uint32_t max_line_bytes;   /* gx_extract_batch with device_pointers.  The caller's PROMISE: no line of the batch -- offsets[i+1] -
                                  offsets[i], terminator included -- is longer than this many code units (0: no promise).
                                  gx_split_lines_max reports it for free; a log shipper knows its own cap.  Without it every batch
                                  kernel is followed by a second, nearly empty launch that takes the lines the batch kernel cannot
                                  stage (longer than a wave's staging area: about 64 x line_bytes_hint bytes; 65 535 for the lane
                                  and hop slice kernels); with it, and when it is within what the chosen kernel takes, that launch
                                  is dropped (1-2 % of a 10 M-line batch).  A promise that does not hold is detected, never
                                  silently wrong: without no_sync the call sees it when it synchronises, runs the follow-up then and
                                  returns the right results; with no_sync the longer line's result row is left UNWRITTEN, and the next
                                  gx_extract_batch on the same stream fails with GX_E_ARG (gx_stat(h, 24) counts such batches). */
} gx_batch_opts;

enum { GX_KERNEL_AUTO = 0, GX_KERNEL_TILES = 1, GX_KERNEL_SLICES = 2, GX_KERNEL_PER_LINE = 3, GX_KERNEL_LANES = 4,
       GX_KERNEL_HOPS = 5 /* the tile kernel on the hop tier's tables (where the handle has them: gx_stat(h, 14)) */,
       GX_KERNEL_HOP_SLICES = 6 /* the same tables under the slice kernel's staging: long and uneven lines */ };

/* Replaces the per-line loop "for each line: Gorp.extract(line)"
 * (core/Gorp.java:145-186 -> PolyMatcher.match core/autom/PolyMatcher.java:123-133
 *  -> JDKRegexpCookedExtraction.match core/jdkre/JDKRegexpCookedExtraction.java:36-59)
 * over a batch held as one CSR byte buffer: line i = bytes[offsets[i] .. offsets[i+1]),
 * each byte one Latin-1 code unit.  match_id[n]; caps[n * 2*gx_max_groups(h)] dense. */
int gx_extract_batch(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n,
                     int32_t* match_id, int32_t* caps, const gx_batch_opts* opts);

/* Line ingestion, the step before the path.  The reference has no counterpart: its callers pass
 * java.lang.Strings read from "a line-oriented input source" (README.md:26) -- BufferedReader.readLine(),
 * whose lines end at "\n", "\r" or "\r\n", the last line needing no terminator.  For a raw byte buffer this
 * writes offsets[0..n] with line i = bytes[offsets[i], offsets[i+1]) INCLUDING its terminator, ready for
 * gx_extract_batch with strip_eol = 1.  offsets holds cap_lines + 1 entries (uint32_t, or uint64_t with
 * opts->offsets64; a uint32_t buffer must be < 4 GiB).  line_flags (optional, cap_lines bytes) receives 1 for
 * every line containing a byte >= 0x80: such a line is Latin-1 only if the file is; UTF-8 text needs the
 * UTF-16 entry points.  Runs on the GPU (three bandwidth-bound passes); with opts->device_pointers = 1
 * bytes / offsets / line_flags are device pointers (bytes 16-byte aligned) and *n_lines (host) is written
 * after a stream synchronisation.  GX_E_LIMIT when the buffer holds more than cap_lines lines (*n_lines is
 * still set, so the caller can retry with a larger offsets array). */
int gx_split_lines(const uint8_t* bytes, uint64_t size, void* offsets, uint64_t cap_lines, uint64_t* n_lines,
                   uint8_t* line_flags, const gx_batch_opts* opts);
/* The same, and *max_line_bytes (host, optional) receives the length of the longest line, terminator included: what
 * gx_batch_opts.max_line_bytes wants to hear (the pass that writes the offsets sees every line end anyway). */
int gx_split_lines_max(const uint8_t* bytes, uint64_t size, void* offsets, uint64_t cap_lines, uint64_t* n_lines,
                       uint8_t* line_flags, uint64_t* max_line_bytes, const gx_batch_opts* opts);

/* Result materialisation, the step after the path: ExtractionResult.asMap(idAs)
 * (core/ExtractionResult.java:65-88) for every matched line of a finished batch, written as one JSON object per
 * line ("JSON Lines") the way Jackson serialises the LinkedHashMap: the id first when id_as != NULL, then
 * extractor name -> captured text in group order (null where Matcher.group() is null), then the extraction's
 * `append` entries (core/DefinitionReader.java:602-640); a key put twice keeps its first position and its last
 * value.  Lines with match_id < 0 produce no text.  bytes/offsets/match_id/caps are the arguments and results
 * of gx_extract_batch.  line_out_offsets (optional, n + 1 entries) receives where each line's text starts
 * (equal neighbours = no text).  *out_size receives the total; out == NULL only asks for the size; GX_E_LIMIT
 * when out_cap is too small.  With opts->device_pointers = 1 every buffer except out_size is a device pointer.
 * Needs the extraction names: a handle from gx_create_from_definition, or gx_set_extraction_meta first.
 * The call's device workspace (sizes, line offsets; for gx_text_to_jsonl also the lines' offsets, ids and capture rows) stays
 * allocated on the handle and is reused by later calls (it grows to what the largest batch asked for; gx_destroy frees it);
 * likewise the narrowed copy of a utf16 batch (a memory pool of the handle's own). */
int gx_results_to_jsonl(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, const int32_t* match_id,
                        const int32_t* caps, const char* id_as, uint8_t* out, uint64_t out_cap, uint64_t* out_size,
                        uint64_t* line_out_offsets, const gx_batch_opts* opts);

/* The three steps in one call, for whole files: raw text -> lines (gx_split_lines semantics) -> the match-and-extract
 * path (terminators ignored) -> JSON Lines (gx_results_to_jsonl semantics).  What the reference's caller writes as
 *     while ((line = reader.readLine()) != null) { r = gorp.extract(line); if (r != null) write(json(r.asMap(idAs))); }
 * (README.md:26,63-79), with extractSafe semantics for lines the capture regexp rejects (no text, counted in
 * *n_exceptions).  Intermediate buffers live and die on the device.  *n_lines / *n_matched / *n_exceptions (each
 * optional) receive the counts; *out_size the size of the text; out == NULL only asks for the size; GX_E_LIMIT when
 * out_cap is too small.  opts: device_pointers (text and out on the device), stream, utf8_passthrough.  Text of 4 GiB
 * and more must be split by the caller (at a line boundary). */
int gx_text_to_jsonl(gx_handle* h, const uint8_t* text, uint64_t size, const char* id_as, uint8_t* out, uint64_t out_cap,
                     uint64_t* out_size, uint64_t* n_lines, uint64_t* n_matched, uint64_t* n_exceptions, const gx_batch_opts* opts);

/* Compact result rows for transport between GPUs (the gather of SURVEY.md section 8(e)): per line one int16 match id
 * followed by `slots` (= 2 * gx_max_groups) uint16 offsets, 0xFFFF = unset: 2 + 2*slots bytes instead of 4 + 4*slots.
 * Device buffers only.  *n_overflow (host) receives the number of offsets above 65534, which do not fit (they are
 * stored saturated): when it is not 0 the caller sends that batch in the wide format.  gx_unpack_results is the
 * inverse.  Both run on the stream in opts (NULL = the null stream); pack synchronises it to read the counter. */
int gx_pack_results(const int32_t* match_id, const int32_t* caps, uint64_t n, int32_t slots, uint16_t* packed,
                    uint64_t* n_overflow, const gx_batch_opts* opts);
int gx_unpack_results(const uint16_t* packed, uint64_t n, int32_t slots, int32_t* match_id, int32_t* caps,
                      const gx_batch_opts* opts);
/* The same from the u8 rows of gx_batch_opts.compact_results = 2 (int8 id, uint8 offsets, 0xFF = unset). */
int gx_unpack_results8(const uint8_t* rows, uint64_t n, int32_t slots, int32_t* match_id, int32_t* caps,
                       const gx_batch_opts* opts);

/* Names for a handle built from regex strings (the caller did DefinitionReader's work itself and holds the
 * CookedExtraction data, core/model/CookedExtraction.java:18-66): extraction name, extractor names in group
 * order (n_names == gx_num_groups(h, k)), and the `append` object as JSON text (NULL = none). */
int gx_set_extraction_meta(gx_handle* h, int32_t k, const char* name, const char* const* extractor_names, int32_t n_names,
                           const char* append_json);

/* Replaces one Gorp.extract(String) call (core/Gorp.java:145-147): s is the
 * String's UTF-16 code units.  Runs on the GPU like the batch path.
 * caps has 2*gx_max_groups(h) slots. */
int gx_extract_one_utf16(gx_handle* h, const uint16_t* s, int32_t len, int32_t* match_id, int32_t* caps);

/* PolyMatcher.match over a batch with the full answer: states[i] = the product-DFA state line i ends in (-1: the dead
 * state, the reference's early return, core/autom/PolyMatcher.java:128-130); gx_state_accepts(h, state, ...) is
 * Automata.accept(state) (core/autom/Automata.java:137-139): all extraction indexes accepting there, ascending
 * (returns the count, <= cap written; 0 for -1).  first_match[i] = their first element or -1, as with
 * gx_extract_batch + match_only.  Runs on the tile kernel where the automaton's dense rows are the tables (in LDS, or in global
 * memory: a row is a state), else on the per-line kernel (the record and hop tables keep only the first match). */
int gx_match_batch(gx_handle* h, const uint8_t* bytes, const void* offsets, uint64_t n, int32_t* first_match,
                   int32_t* states, const gx_batch_opts* opts);
int gx_state_accepts(const gx_handle* h, int32_t state, int32_t* indexes, int32_t cap);

/* Replaces CookedExtraction.match(String) (core/model/CookedExtraction.java:61; JDKRegexpCookedExtraction.match,
 * core/jdkre/JDKRegexpCookedExtraction.java:36-39) -- the product of the reference's plugin seam, ExtractionCooker.cook
 * (core/ExtractionCooker.java:22): extraction k's capture regexp alone against one String, no matcher stage.
 * *matched = 1 and caps filled (2*gx_max_groups(h) slots) when the regexp matches the whole line, else 0 (the reference
 * returns null). */
int gx_capture_one_utf16(gx_handle* h, int32_t k, const uint16_t* s, int32_t len, int32_t* matched, int32_t* caps);

/* Replaces PolyMatcher.match(CharSequence) -> int[] (core/autom/PolyMatcher.java:123-133):
 * all matching extraction indexes, ascending.  Returns the count (<= cap written), or <0 on error. */
int gx_match_one_utf16(gx_handle* h, const uint16_t* s, int32_t len, int32_t* indexes, int32_t cap);

/* Definition-time string rewriting that defines the two regex dialects
 * (RegexHelper.quoteLiteralAsRegexp core/util/RegexHelper.java:20-70,
 *  massageRegexpForAutomaton :79-182, massageRegexpForJDK :210-237), used by
 * Gorp._buildExtractor (core/Gorp.java:94-129) to produce the inputs of
 * gx_create_from_patterns.  UTF-8 in, NUL-terminated UTF-8 out; *out_len
 * receives the length without the NUL.  GX_E_ARG with *out_len set when cap
 * is too small; GX_E_REGEX_SYNTAX for the IllegalArgumentException cases. */
int gx_quote_literal_as_regexp(const char* text, char* out, size_t cap, size_t* out_len);
int gx_massage_regexp_for_automaton(const char* pattern, char* out, size_t cap, size_t* out_len);
int gx_massage_regexp_for_jdk(const char* pattern, char* out, size_t cap, size_t* out_len);

/* Native front-end for the definition language (.grp text): replaces
 * DefinitionReader.reader(String).read() (core/DefinitionReader.java:58-84) -- tokenising
 * (:126-181,189-640), pattern / template / extraction resolution (core/model/CookedDefinitions.java:57-453)
 * and Gorp.construct (core/Gorp.java:50-92).  GX_E_DEFINITION carries the reference's
 * DefinitionParseException text, "([source (row,col)]): message".
 * gx_definition_to_json returns what the reference keeps in Java objects: stage "flattened" = per
 * extraction its name, flattened pieces, extractor names (capture-group order), `append` object and the two
 * regex strings; stages "uncooked" / "cooked" expose the intermediate piece lists that the reference's own
 * parser tests assert on.  NUL-terminated UTF-8 out; *out_len = length without the NUL (GX_E_ARG with
 * *out_len set when cap is too small). */
int gx_create_from_definition(const char* definition_text, const char* source_ref, uint32_t flags, gx_handle** out);
int gx_definition_to_json(const char* definition_text, const char* source_ref, const char* stage,
                          char* out, size_t cap, size_t* out_len);
/* Metadata of a handle built by gx_create_from_definition or completed with gx_set_extraction_meta (NULL otherwise /
 * out of range); strings live as long as the handle (or until the next gx_set_extraction_meta).  CookedExtraction.getName() (core/model/CookedExtraction.java:36), the extractor names in
 * capture-group order (FlattenedExtraction.getExtractorNames()), and getExtra() as JSON object text. */
const char* gx_extraction_name(const gx_handle* h, int32_t k);
const char* gx_extractor_name(const gx_handle* h, int32_t k, int32_t g);
const char* gx_extraction_append_json(const gx_handle* h, int32_t k);
/* getExtra() entry by entry, in order: the key (decoded, UTF-8) and the value as JSON text (a string value keeps
 * its quotes and escapes); count = 0 when the extraction appends nothing. */
int32_t gx_extraction_append_count(const gx_handle* h, int32_t k);
const char* gx_extraction_append_key(const gx_handle* h, int32_t k, int32_t j);
const char* gx_extraction_append_value_json(const gx_handle* h, int32_t k, int32_t j);

/* Thread-local message for the last failing call on this thread. */
const char* gx_last_error(void);

/* Devices.  A handle lives on the device that is current for the calling thread when it is created (HIP's per-thread
 * current device; device 0 unless gx_set_device was called on that thread) and every call on the handle runs there,
 * whichever thread makes it.  One process drives several GPUs with one handle per device -- built from the same
 * definition, or from one blob (gx_create_from_blob) -- and either its own threads, each calling gx_extract_batch on
 * its handle (Gorp "may be used concurrently", core/Gorp.java:22), or gx_extract_batch_multi below. */
int gx_device_count(void);
/* gx_split_lines / gx_split_lines_max keep one workspace per device between calls (an eighth of the largest text they have seen
 * there, a quarter with line flags); this gives a device's back.  Calls on different devices do not wait for each other. */
int gx_release_scratch(int device);
int gx_set_device(int device);
int gx_handle_device(const gx_handle* h);   /* -1 for a host-only handle */

/* One CSR batch in HOST memory over several devices: the lines are cut into n_handles contiguous shards of about
 * equal bytes (lines are independent: no exchange between the shards), shard k runs on handles[k]'s device through that
 * handle's host pipeline, all shards concurrently, and the results land in the caller's arrays in line order.  The
 * handles must come from the same definition.  opts as for gx_extract_batch (host pointers only; stream is ignored). */
int gx_extract_batch_multi(gx_handle* const* handles, int32_t n_handles, const uint8_t* bytes, const void* offsets, uint64_t n,
                           int32_t* match_id, int32_t* caps, const gx_batch_opts* opts);

/* The same for batches that are RESIDENT on the devices (the one-process-many-GPUs layout without the bus in the way): shard k is a
 * CSR batch in the memory of shards[k].handle's device -- bytes / offsets / n / match_id / caps as for gx_extract_batch with
 * device_pointers, `overflow` a device uint64_t on that device (or NULL), `stream` a hipStream_t of that device (NULL: a stream the
 * handle keeps for this purpose).  One call enqueues every shard on its device from the calling thread, then -- unless
 * opts->no_sync -- waits for all of them; the shards run concurrently.  No exchange between the shards: lines are independent.
 * opts as for gx_extract_batch (device_pointers is implied; stream and overflow are per shard, those of opts are ignored).  A shard
 * that fails does not stop the others; the first failure is returned. */
typedef struct gx_device_shard {
    gx_handle* handle;
    const uint8_t* bytes;
    const void* offsets;
    uint64_t n;
    int32_t* match_id;
    int32_t* caps;
    void* overflow;
    void* stream;
} gx_device_shard;
int gx_extract_batch_multi_device(const gx_device_shard* shards, int32_t n_shards, const gx_batch_opts* opts);

/* One process, all GPUs of a node (the reference's caller is ONE JVM, "fully thread-safe and may be used concurrently",
 * core/Gorp.java:22; BASELINE north_star: "broadcast of the DFA tables and a final gather over xGMI").  Ranks of a multi-process
 * job exchange the blob and the rows with RCCL (gorp_amd/dist.py); this is the same for the caller that owns every device itself.
 *
 * gx_create_on_devices: one handle per device from ONE blob (gx_blob_copy of a handle compiled once).  handles[0] is built from
 * the blob on devices[0]; the others take their device images -- the class maps, the LDS table images, the dense rows and hop
 * records in global memory: 9 KB for the README definition, 18 MB for 512 extractions -- from devices[0]'s copy, device to device
 * (hipMemcpyPeer: over xGMI between peers; gx_stat(h, 30) = bytes that came this way), their host-side tables built in threads of
 * their own.  All or nothing: on failure every handle that was created is destroyed and handles[] is NULL.
 *
 * gx_gather_rows: the result rows of a sharded batch (gx_extract_batch_multi_device with compact_results 1 or 2; any fixed row
 * size works: dense caps rows are 8 G bytes) onto ONE device, in shard order: dst_rows[sum of n over the shards before k ...].
 * Each shard's copy is enqueued on a copy stream of the SHARD's device behind the shard's kernel (an event on `stream`: the
 * stream that kernel was enqueued on; NULL = the stream gx_extract_batch_multi_device uses when the shard brings none), as one
 * hipMemcpyPeerAsync -- seven peers push into the root over seven links at once, nothing funnels through the host.  no_sync != 0:
 * returns when everything is enqueued; gx_gather_wait(handles) waits for those copies.  The next batch's kernels may be enqueued
 * (on the kernel streams) before the gather of this one is waited for -- a two-deep pipeline with two row buffers per shard: wait
 * for gather k before batch k + 2 writes the rows gather k reads ("One process, eight GPUs"). */
typedef struct gx_rows_shard {
    gx_handle* handle;      /* names the shard's device */
    const void* rows;       /* on that device: n rows of row_bytes bytes */
    uint64_t n;
    void* stream;           /* hipStream_t the shard's kernel was enqueued on (NULL: the handle's own multi-device stream) */
} gx_rows_shard;
int gx_create_on_devices(const void* blob, size_t size, const int32_t* devices, int32_t n_devices, uint32_t flags, gx_handle** handles);
int gx_gather_rows(const gx_rows_shard* shards, int32_t n_shards, uint32_t row_bytes, int32_t dst_device, void* dst_rows, int32_t no_sync);
int gx_gather_wait(gx_handle* const* handles, int32_t n_handles);

/* Host buffers that are handed to gx_extract_batch again and again (a JNI caller's direct ByteBuffers) can be pinned
 * once: copies from and to pinned memory run at the bus' rate without a staging copy by the CPU (hipHostRegister /
 * hipHostUnregister). */
int gx_host_register(void* p, size_t bytes);
int gx_host_unregister(void* p);

#ifdef __cplusplus
}
#endif
#endif /* GORP_HIP_H */
