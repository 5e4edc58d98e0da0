#include "session.hpp"

#include <algorithm>
#include <stdexcept>
#include <utility>

namespace lc {

const char* ConsensusSession::kNotReadyMessage =
    "System error: Requested answer when answer was not ready.";

ConsensusSession::ConsensusSession(std::vector<std::string> member_names, SessionConfig cfg,
                                   uint64_t question_index)
    : names_(std::move(member_names)),
      cfg_(cfg),
      rng_(cfg.seed * 0x9E3779B97F4A7C15ull + question_index + 1),
      question_index_(question_index),
      feedback_(names_.size(), -1) {
  if (cfg_.max_rounds < 1) cfg_.max_rounds = 1;
}

void ConsensusSession::reset() {
  question_.clear();
  answer_.clear();
  has_question_ = false;
  has_answer_ = false;
  std::fill(feedback_.begin(), feedback_.end(), static_cast<int8_t>(-1));
  votes_in_ = 0;
  evaluation_count_ = 0;
  member_calls_ = 0;
  proposer_ = -1;
  forced_ = false;
  awaiting_refinement_ = false;
  transcript_.clear();
}

Action ConsensusSession::ask(const std::string& question) {
  if (names_.empty())
    throw std::runtime_error("ConsensusSession::ask: no members registered");
  question_ = question;
  has_question_ = true;
  const int n = num_members();
  switch (cfg_.proposer_policy) {
    case ProposerPolicy::Random:
      proposer_ = static_cast<int>(rng_.below(static_cast<uint64_t>(n)));
      break;
    case ProposerPolicy::RoundRobin:
      proposer_ = static_cast<int>(question_index_ % static_cast<uint64_t>(n));
      break;
    case ProposerPolicy::Fixed:
      proposer_ = ((cfg_.fixed_proposer % n) + n) % n;
      break;
  }
  ++member_calls_;
  return {Action::Propose, proposer_};
}

Action ConsensusSession::fan_out_evaluate() {
  std::fill(feedback_.begin(), feedback_.end(), static_cast<int8_t>(-1));
  votes_in_ = 0;
  ++evaluation_count_;
  member_calls_ += num_members();
  RoundRecord rec;
  rec.round = evaluation_count_;
  rec.answer = answer_;
  transcript_.push_back(std::move(rec));
  return {Action::Evaluate, -1};
}

Action ConsensusSession::on_proposal(const std::string& answer) {
  if (!has_question_)
    throw std::runtime_error("ConsensusSession::on_proposal: no question in flight");
  answer_ = answer;
  has_answer_ = true;
  return fan_out_evaluate();
}

Action ConsensusSession::on_evaluation(int member, Verdict v, const std::string& reasoning) {
  if (member < 0 || member >= num_members())
    throw std::out_of_range("ConsensusSession::on_evaluation: bad member index");
  if (!has_answer_ || transcript_.empty() || awaiting_refinement_ || forced_)
    return {Action::None, -1};  // stale vote (arrived after the round closed)
  if (v == Verdict::Abstain && cfg_.failure_policy == FailurePolicy::NeedsRefinement)
    v = Verdict::NeedsRefinement;
  if (feedback_[member] < 0) ++votes_in_;  // a repeated vote overwrites (HashMap::insert, main.rs:264)
  feedback_[member] = static_cast<int8_t>(v);
  auto& votes = transcript_.back().votes;
  bool replaced = false;
  for (auto& r : votes)
    if (r.member == member) {
      r.verdict = v;
      r.reasoning = reasoning;
      replaced = true;
    }
  if (!replaced) votes.push_back({member, v, reasoning});

  if (votes_in_ < num_members()) return {Action::None, -1};

  // Fan-in barrier reached (main.rs:265).
  std::vector<int> dissenters;
  for (int i = 0; i < num_members(); ++i)
    if (feedback_[i] == static_cast<int8_t>(Verdict::NeedsRefinement)) dissenters.push_back(i);
  if (dissenters.empty()) return {Action::Done, -1};

  int pick = dissenters[rng_.below(dissenters.size())];
  transcript_.back().refiner = pick;
  awaiting_refinement_ = true;
  ++member_calls_;
  return {Action::Refine, pick};
}

Action ConsensusSession::on_member_failure(int member) {
  return on_evaluation(member, Verdict::Abstain, "member failed");
}

Action ConsensusSession::on_refinement(const std::string& answer) {
  if (!awaiting_refinement_)
    throw std::runtime_error("ConsensusSession::on_refinement: no refinement was requested");
  awaiting_refinement_ = false;
  answer_ = answer;
  has_answer_ = true;
  if (evaluation_count_ < cfg_.max_rounds) return fan_out_evaluate();
  // Round budget exhausted: accept the last refinement unevaluated (main.rs:308-311).
  for (auto& f : feedback_) f = static_cast<int8_t>(Verdict::Good);
  votes_in_ = num_members();
  forced_ = true;
  return {Action::Done, -1};
}

Action ConsensusSession::abort_with_current_answer() {
  if (!has_answer_) return {Action::Done, -1};
  for (auto& f : feedback_) f = static_cast<int8_t>(Verdict::Good);
  votes_in_ = num_members();
  forced_ = true;
  awaiting_refinement_ = false;
  return {Action::Done, -1};
}

bool ConsensusSession::ready() const {
  if (!has_answer_ || votes_in_ == 0 || votes_in_ != num_members()) return false;
  for (auto f : feedback_)
    if (f == static_cast<int8_t>(Verdict::NeedsRefinement)) return false;
  return true;
}

std::string ConsensusSession::answer() const {
  return has_answer_ ? answer_ : std::string(kNotReadyMessage);
}

std::string ConsensusSession::refiner_context() const {
  if (!cfg_.forward_reasoning_to_refiner || transcript_.empty()) return {};
  std::string out;
  for (const auto& v : transcript_.back().votes) {
    if (v.verdict != Verdict::NeedsRefinement || v.reasoning.empty()) continue;
    if (!out.empty()) out += "\n";
    out += "- " + names_[v.member] + ": " + v.reasoning;
  }
  return out;
}

}  // namespace lc
