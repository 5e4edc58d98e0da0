// Consensus protocol state machine: propose -> evaluate by all -> (unanimous Good ?
// done : a random dissenter refines -> evaluate again), at most `max_rounds`
// evaluation rounds.
//
// This is the reference's Coordinator actor (/root/reference/src/main.rs:187-348)
// rebuilt as a passive, per-question object: callers feed it member replies and it
// returns the next Action. Many sessions can be in flight at once (the reference
// holds a single `current_question`, main.rs:189-195) and nothing polls: completion
// is the `Done` action (the reference's REPL busy-spins and polls every 500 ms,
// main.rs:448-459).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "rng.hpp"
#include "verdict.hpp"

namespace lc {

enum class ProposerPolicy : uint8_t {
  Random = 0,      // uniform over members (main.rs:228-229)
  RoundRobin = 1,  // question i -> member i mod N
  Fixed = 2,       // always `fixed_proposer`
};

enum class FailurePolicy : uint8_t {
  Abstain = 0,          // a failed member neither blocks unanimity nor refines
  NeedsRefinement = 1,  // a failed member counts as a dissenter (closest to main.rs:149-152)
};

struct SessionConfig {
  int max_rounds = 5;  // main.rs:300 hard-codes 5 ("TODO: Make max count configurable")
  ProposerPolicy proposer_policy = ProposerPolicy::Random;
  int fixed_proposer = 0;
  FailurePolicy failure_policy = FailurePolicy::Abstain;
  bool forward_reasoning_to_refiner = false;  // reference: false (main.rs:56-61)
  uint64_t seed = 0;
};

struct Action {
  enum Kind : uint8_t {
    None = 0,      // waiting for more replies (fan-in barrier not reached, main.rs:265)
    Propose = 1,   // ask `member` to answer the question
    Evaluate = 2,  // ask EVERY member, the proposer included, to judge `answer` (main.rs:250-253)
    Refine = 3,    // ask `member` (a dissenter) to rewrite `answer` (main.rs:268-282)
    Done = 4,      // consensus reached or round budget exhausted
  };
  Kind kind = None;
  int member = -1;
};

struct VoteRecord {
  int member = -1;
  Verdict verdict = Verdict::NeedsRefinement;
  std::string reasoning;
};

struct RoundRecord {
  int round = 0;  // 1-based evaluation round
  std::string answer;
  std::vector<VoteRecord> votes;
  int refiner = -1;  // member asked to refine after this round, -1 if none
};

class ConsensusSession {
 public:
  ConsensusSession(std::vector<std::string> member_names, SessionConfig cfg,
                   uint64_t question_index = 0);

  // --- events (each returns the next action) ---
  Action ask(const std::string& question);                     // main.rs:220-240
  Action on_proposal(const std::string& answer);               // main.rs:242-257
  Action on_evaluation(int member, Verdict v, const std::string& reasoning);  // main.rs:259-291
  Action on_member_failure(int member);                        // no reference analogue
  Action on_refinement(const std::string& answer);             // main.rs:293-314
  Action abort_with_current_answer();                          // deadline hit: force completion

  // --- queries ---
  bool ready() const;                 // main.rs:316-325
  std::string answer() const;         // main.rs:327-336 (incl. the "System error" literal)
  void reset();                       // main.rs:197-203, :338-345

  const std::string& question() const { return question_; }
  bool has_answer() const { return has_answer_; }
  int evaluation_count() const { return evaluation_count_; }
  int member_calls() const { return member_calls_; }
  int num_members() const { return static_cast<int>(names_.size()); }
  int proposer() const { return proposer_; }
  bool forced() const { return forced_; }  // true if the round budget broke the loop
  const std::vector<std::string>& member_names() const { return names_; }
  const std::vector<RoundRecord>& transcript() const { return transcript_; }
  // Reasoning of the current round's dissenters, joined; empty unless configured.
  std::string refiner_context() const;

  static const char* kNotReadyMessage;

 private:
  Action fan_out_evaluate();

  std::vector<std::string> names_;
  SessionConfig cfg_;
  SplitMix64 rng_;
  uint64_t question_index_;

  std::string question_;
  std::string answer_;
  bool has_question_ = false;
  bool has_answer_ = false;
  std::vector<int8_t> feedback_;  // -1 = no vote yet, else Verdict
  int votes_in_ = 0;
  int evaluation_count_ = 0;
  int member_calls_ = 0;
  int proposer_ = -1;
  bool forced_ = false;
  bool awaiting_refinement_ = false;
  std::vector<RoundRecord> transcript_;
};

}  // namespace lc
