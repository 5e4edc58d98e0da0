// Prompt templates and the default persona set.
//
// Three templates, one per member role, with the same semantics as the reference's
// inline format strings: propose (/root/reference/src/main.rs:95), evaluate
// (main.rs:111-136) and refine (main.rs:166-175). The wording here is this
// project's own; the structure that matters for behaviour is kept:
//   * evaluate/refine put "Question" and "Answer" FIRST and the persona text last,
//     so evaluate prompts of different personas share a KV prefix;
//   * the evaluator must answer with one of exactly two words on the first line and
//     put reasoning on a following line;
//   * an evaluator whose domain is unrelated to the question answers Good;
//   * two few-shot examples are appended to the evaluate prompt;
//   * every double quote is removed from the rendered evaluate/refine prompt
//     (main.rs:136, :175).
#pragma once
#include <string>
#include <vector>

namespace lc {

struct Persona {
  std::string name;    // registry key (main.rs:18, :73)
  std::string domain;  // knowledge domain inserted into the prompts (main.rs:74)
  std::string tuning;  // multi-line bullet list, starts with '\n' (main.rs:75)
};

// The four personas the reference registers at start-up (main.rs:359-426): same
// names and domains, ten guidance bullets each.
std::vector<Persona> default_personas();

struct PromptTemplates {
  std::string propose;   // placeholders: {question}
  std::string evaluate;  // placeholders: {question} {answer} {domain} {tuning}
  std::string refine;    // placeholders: {question} {answer} {domain} {tuning} {reasoning}
  bool strip_quotes = true;
  static PromptTemplates defaults();
  // Same instructions with everything that depends on the persona moved to the END of the
  // evaluate prompt (the reference names the domain a third of the way in, main.rs:118, and
  // lists the tuning bullets before the verdict rules and examples, main.rs:120). With N
  // personas on one model the rules and examples are then prefilled once per answer instead
  // of N times: the shared prefix grows from roughly a quarter of the prompt to four fifths.
  static PromptTemplates persona_last();
  // The opposite order: everything that depends on the persona, the verdict rules and the
  // examples come FIRST, the question and the answer last. That head is the same for every
  // question a persona ever judges, so after the first question it is served from the prefix
  // cache (across questions of a step and across steps) and an evaluate call prefills only the
  // question and the answer. Nothing is shared between the personas of one answer.
  static PromptTemplates persona_first();
};

std::string render_propose(const PromptTemplates& t, const std::string& question);
std::string render_evaluate(const PromptTemplates& t, const std::string& question,
                            const std::string& answer, const Persona& p);
// `reasoning` is optional extra context (the reference does not forward evaluator
// reasoning to the refiner: RefineAnswer carries only question+answer, main.rs:56-61).
std::string render_refine(const PromptTemplates& t, const std::string& question,
                          const std::string& answer, const Persona& p,
                          const std::string& reasoning);

// The part of the rendered evaluate prompt that does not depend on the persona
// (everything before the first {domain}/{tuning} placeholder). Its token count is
// what a shared-prefix prefill can reuse across personas hosted on one model.
std::string evaluate_shared_prefix(const PromptTemplates& t, const std::string& question,
                                   const std::string& answer);

// Multiple-choice prompt used by the vote reducers on synthetic benchmarks.
std::string render_multiple_choice(const std::string& question,
                                   const std::vector<std::string>& options,
                                   const std::string& choice_letters, bool chain_of_thought);

}  // namespace lc
