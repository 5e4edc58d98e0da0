// Evaluator verdicts and the parsers that turn generated text into votes.
//
// Verdict mirrors the reference's `Feedback` enum (/root/reference/src/main.rs:7-12)
// and adds `Abstain` for members that failed or timed out (the reference has no
// failure path: every backend error is an `.expect()` panic, main.rs:97,138,178).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace lc {

enum class Verdict : uint8_t { Good = 0, NeedsRefinement = 1, Abstain = 2 };

const char* verdict_name(Verdict v);

enum class ParseMode : uint8_t {
  // Exactly the reference's rule (main.rs:139-153): split on '\n', drop empty lines,
  // strip every ' ' from the first remaining line, accept only the bare words
  // "Good" / "NeedsRefinement"; anything else counts as NeedsRefinement.
  Strict = 0,
  // Also accepts what a model imitating the reference's few-shot examples emits
  // ("Evaluation: Good", main.rs:129,135), case-insensitively, with trailing
  // punctuation, markdown emphasis, or the reasoning on the same line.
  Tolerant = 1,
};

struct ParsedVerdict {
  Verdict verdict = Verdict::NeedsRefinement;
  std::string reasoning;   // remaining non-empty lines joined by "\n\n" (main.rs:143)
  bool well_formed = false;  // false => the reference would have logged error! (main.rs:150)
};

// Never throws. An all-empty reply (the reference indexes result_parts[0] and would
// panic, main.rs:142) yields {NeedsRefinement, "", well_formed=false}.
ParsedVerdict parse_verdict(const std::string& reply, ParseMode mode);

// Multiple-choice answer extraction for the vote reducers. Looks for the last
// "answer is X" / "Answer: X" / "(X)" / leading bare "X" pattern where X is one of
// `choices` (e.g. "ABCD"). Returns the index into `choices`, or -1.
int extract_choice(const std::string& text, const std::string& choices);

// Free-form answers are compared after normalisation: lower-case, collapse
// whitespace, strip surrounding punctuation. Used to bucket identical answers for
// majority vote over non-multiple-choice outputs.
std::string normalize_answer(const std::string& text);

// Extract the final numeric answer (last number in the text, commas stripped,
// "#### 42" GSM-style marker preferred). Returns false if none.
bool extract_number(const std::string& text, double* out);

}  // namespace lc
