// pybind11 surface of the native consensus core (module llm_consensus_amd._native._core).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "kv_manager.hpp"
#include "metrics.hpp"
#include "placement.hpp"
#include "prompts.hpp"
#include "reducers.hpp"
#include "scheduler.hpp"
#include "session.hpp"
#include "tokenizer.hpp"
#include "verdict.hpp"

namespace py = pybind11;
using namespace lc;

namespace {

template <typename T>
py::array_t<T> to_array(const std::vector<T>& v) {
  py::array_t<T> a(static_cast<py::ssize_t>(v.size()));
  if (!v.empty()) std::copy(v.begin(), v.end(), a.mutable_data());
  return a;
}

using i32arr = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;
using f32arr = py::array_t<float, py::array::c_style | py::array::forcecast>;

}  // namespace

PYBIND11_MODULE(_core, m) {
  m.doc() = "Native consensus core: protocol state machine, parsers, reducers, KV manager, scheduler";

  // ---- verdicts / parsing ----
  py::enum_<Verdict>(m, "Verdict")
      .value("Good", Verdict::Good)
      .value("NeedsRefinement", Verdict::NeedsRefinement)
      .value("Abstain", Verdict::Abstain);
  py::enum_<ParseMode>(m, "ParseMode").value("Strict", ParseMode::Strict).value("Tolerant", ParseMode::Tolerant);
  py::class_<ParsedVerdict>(m, "ParsedVerdict")
      .def_readonly("verdict", &ParsedVerdict::verdict)
      .def_readonly("reasoning", &ParsedVerdict::reasoning)
      .def_readonly("well_formed", &ParsedVerdict::well_formed);
  m.def("parse_verdict", &parse_verdict, py::arg("reply"), py::arg("mode") = ParseMode::Strict);
  m.def("extract_choice", &extract_choice, py::arg("text"), py::arg("choices") = "ABCD");
  m.def("normalize_answer", &normalize_answer);
  m.def("extract_number", [](const std::string& s) -> py::object {
    double v;
    if (extract_number(s, &v)) return py::float_(v);
    return py::none();
  });

  // ---- prompts / personas ----
  py::class_<Persona>(m, "Persona")
      .def(py::init<>())
      .def(py::init([](std::string n, std::string d, std::string t) { return Persona{n, d, t}; }),
           py::arg("name"), py::arg("domain"), py::arg("tuning"))
      .def_readwrite("name", &Persona::name)
      .def_readwrite("domain", &Persona::domain)
      .def_readwrite("tuning", &Persona::tuning);
  m.def("default_personas", &default_personas);
  py::class_<PromptTemplates>(m, "PromptTemplates")
      .def(py::init(&PromptTemplates::defaults))
      .def_static("persona_last", &PromptTemplates::persona_last)
      .def_static("persona_first", &PromptTemplates::persona_first)
      .def_readwrite("propose", &PromptTemplates::propose)
      .def_readwrite("evaluate", &PromptTemplates::evaluate)
      .def_readwrite("refine", &PromptTemplates::refine)
      .def_readwrite("strip_quotes", &PromptTemplates::strip_quotes);
  m.def("render_propose", &render_propose);
  m.def("render_evaluate", &render_evaluate);
  m.def("render_refine", &render_refine, py::arg("templates"), py::arg("question"), py::arg("answer"),
        py::arg("persona"), py::arg("reasoning") = "");
  m.def("evaluate_shared_prefix", &evaluate_shared_prefix);
  m.def("render_multiple_choice", &render_multiple_choice, py::arg("question"), py::arg("options"),
        py::arg("choice_letters") = "ABCD", py::arg("chain_of_thought") = false);

  // ---- session ----
  py::enum_<ProposerPolicy>(m, "ProposerPolicy")
      .value("Random", ProposerPolicy::Random)
      .value("RoundRobin", ProposerPolicy::RoundRobin)
      .value("Fixed", ProposerPolicy::Fixed);
  py::enum_<FailurePolicy>(m, "FailurePolicy")
      .value("Abstain", FailurePolicy::Abstain)
      .value("NeedsRefinement", FailurePolicy::NeedsRefinement);
  py::class_<SessionConfig>(m, "SessionConfig")
      .def(py::init<>())
      .def_readwrite("max_rounds", &SessionConfig::max_rounds)
      .def_readwrite("proposer_policy", &SessionConfig::proposer_policy)
      .def_readwrite("fixed_proposer", &SessionConfig::fixed_proposer)
      .def_readwrite("failure_policy", &SessionConfig::failure_policy)
      .def_readwrite("forward_reasoning_to_refiner", &SessionConfig::forward_reasoning_to_refiner)
      .def_readwrite("seed", &SessionConfig::seed);
  py::class_<Action> action(m, "Action");
  py::enum_<Action::Kind>(action, "Kind")
      .value("None_", Action::None)
      .value("Propose", Action::Propose)
      .value("Evaluate", Action::Evaluate)
      .value("Refine", Action::Refine)
      .value("Done", Action::Done);
  action.def_readonly("kind", &Action::kind).def_readonly("member", &Action::member);
  py::class_<VoteRecord>(m, "VoteRecord")
      .def_readonly("member", &VoteRecord::member)
      .def_readonly("verdict", &VoteRecord::verdict)
      .def_readonly("reasoning", &VoteRecord::reasoning);
  py::class_<RoundRecord>(m, "RoundRecord")
      .def_readonly("round", &RoundRecord::round)
      .def_readonly("answer", &RoundRecord::answer)
      .def_readonly("votes", &RoundRecord::votes)
      .def_readonly("refiner", &RoundRecord::refiner);
  py::class_<ConsensusSession>(m, "ConsensusSession")
      .def(py::init<std::vector<std::string>, SessionConfig, uint64_t>(), py::arg("member_names"),
           py::arg("config") = SessionConfig{}, py::arg("question_index") = 0)
      .def("ask", &ConsensusSession::ask)
      .def("on_proposal", &ConsensusSession::on_proposal)
      .def("on_evaluation", &ConsensusSession::on_evaluation, py::arg("member"), py::arg("verdict"),
           py::arg("reasoning") = "")
      .def("on_member_failure", &ConsensusSession::on_member_failure)
      .def("on_refinement", &ConsensusSession::on_refinement)
      .def("abort_with_current_answer", &ConsensusSession::abort_with_current_answer)
      .def("ready", &ConsensusSession::ready)
      .def("answer", &ConsensusSession::answer)
      .def("reset", &ConsensusSession::reset)
      .def("refiner_context", &ConsensusSession::refiner_context)
      .def_property_readonly("question", &ConsensusSession::question)
      .def_property_readonly("has_answer", &ConsensusSession::has_answer)
      .def_property_readonly("evaluation_count", &ConsensusSession::evaluation_count)
      .def_property_readonly("member_calls", &ConsensusSession::member_calls)
      .def_property_readonly("num_members", &ConsensusSession::num_members)
      .def_property_readonly("proposer", &ConsensusSession::proposer)
      .def_property_readonly("forced", &ConsensusSession::forced)
      .def_property_readonly("member_names", &ConsensusSession::member_names)
      .def_property_readonly("transcript", &ConsensusSession::transcript);
  m.attr("NOT_READY_MESSAGE") = ConsensusSession::kNotReadyMessage;

  // ---- reducers ----
  py::enum_<TieBreak>(m, "TieBreak")
      .value("LowestClass", TieBreak::LowestClass)
      .value("FirstVoter", TieBreak::FirstVoter)
      .value("HighestWeight", TieBreak::HighestWeight);
  py::class_<VoteResult>(m, "VoteResult")
      .def_readonly("winner", &VoteResult::winner)
      .def_readonly("count", &VoteResult::count)
      .def_readonly("valid", &VoteResult::valid)
      .def_readonly("agreement", &VoteResult::agreement)
      .def_readonly("unanimous", &VoteResult::unanimous);
  m.def(
      "majority_vote",
      [](i32arr votes, int num_classes, TieBreak tie, py::object weights) {
        if (votes.ndim() != 1) throw std::invalid_argument("majority_vote: votes must be 1-D");
        if (weights.is_none())
          return majority_vote(votes.data(), static_cast<int>(votes.shape(0)), num_classes, tie);
        f32arr w = weights.cast<f32arr>();
        if (w.size() != votes.size()) throw std::invalid_argument("majority_vote: weights size mismatch");
        return majority_vote(votes.data(), static_cast<int>(votes.shape(0)), num_classes, tie, w.data());
      },
      py::arg("votes"), py::arg("num_classes"), py::arg("tie") = TieBreak::LowestClass,
      py::arg("weights") = py::none());
  m.def(
      "majority_vote_batch",
      [](i32arr votes, int num_classes, TieBreak tie, py::object weights) {
        if (votes.ndim() != 2) throw std::invalid_argument("majority_vote_batch: votes must be [R, N]");
        const int R = static_cast<int>(votes.shape(0)), N = static_cast<int>(votes.shape(1));
        if (weights.is_none()) return majority_vote_batch(votes.data(), R, N, num_classes, tie);
        f32arr w = weights.cast<f32arr>();
        if (w.size() != N) throw std::invalid_argument("majority_vote_batch: need one weight per voter");
        return majority_vote_batch(votes.data(), R, N, num_classes, tie, w.data());
      },
      py::arg("votes"), py::arg("num_classes"), py::arg("tie") = TieBreak::LowestClass,
      py::arg("weights") = py::none());
  py::class_<MedoidResult>(m, "MedoidResult")
      .def_readonly("index", &MedoidResult::index)
      .def_readonly("score", &MedoidResult::score)
      .def_readonly("mean_pairwise", &MedoidResult::mean_pairwise);
  m.def("cosine_medoid", [](f32arr E) {
    if (E.ndim() != 2) throw std::invalid_argument("cosine_medoid: E must be [n, d]");
    return cosine_medoid(E.data(), static_cast<int>(E.shape(0)), static_cast<int>(E.shape(1)));
  });
  py::class_<JudgeResult>(m, "JudgeResult")
      .def_readonly("winner", &JudgeResult::winner)
      .def_readonly("score", &JudgeResult::score)
      .def_readonly("unanimous", &JudgeResult::unanimous);
  m.def("judge_reduce", [](f32arr s) {
    if (s.ndim() != 2) throw std::invalid_argument("judge_reduce: scores must be [C, J]");
    return judge_reduce(s.data(), static_cast<int>(s.shape(0)), static_cast<int>(s.shape(1)));
  });
  py::class_<UnanimityResult>(m, "UnanimityResult")
      .def_readonly("all_good", &UnanimityResult::all_good)
      .def_readonly("dissenters", &UnanimityResult::dissenters);
  m.def("unanimity_from_logprobs", [](f32arr lp) {
    if (lp.ndim() != 2 || lp.shape(1) != 2) throw std::invalid_argument("unanimity_from_logprobs: lp must be [N, 2]");
    return unanimity_from_logprobs(lp.data(), static_cast<int>(lp.shape(0)));
  });

  // ---- KV manager ----
  py::class_<KVStats>(m, "KVStats")
      .def_readonly("prompt_tokens", &KVStats::prompt_tokens)
      .def_readonly("prefix_hit_tokens", &KVStats::prefix_hit_tokens)
      .def_readonly("cow_copies", &KVStats::cow_copies)
      .def_readonly("evictions", &KVStats::evictions)
      .def_readonly("blocks_total", &KVStats::blocks_total)
      .def_readonly("blocks_free", &KVStats::blocks_free)
      .def_readonly("blocks_cached", &KVStats::blocks_cached);
  py::class_<KVManager, std::shared_ptr<KVManager>>(m, "KVManager")
      .def(py::init<int, int, bool>(), py::arg("num_blocks"), py::arg("block_size"),
           py::arg("enable_prefix_cache") = true)
      .def("create_sequence",
           [](KVManager& kv, i32arr tokens, bool allow_cache_hit) {
             int cached = 0;
             int id = kv.create_sequence(tokens.data(), static_cast<int>(tokens.size()), &cached, allow_cache_hit);
             return py::make_tuple(id, cached);
           },
           py::arg("tokens"), py::arg("allow_cache_hit") = true)
      .def("reserve", &KVManager::reserve)
      .def("commit", [](KVManager& kv, int seq, i32arr tokens) {
        kv.commit(seq, tokens.data(), static_cast<int>(tokens.size()));
      })
      .def("extend_match", [](KVManager& kv, int seq, i32arr tokens) {
        return kv.extend_match(seq, tokens.data(), static_cast<int>(tokens.size()));
      })
      .def("fork", &KVManager::fork)
      .def("free_sequence", &KVManager::free_sequence)
      .def("num_tokens", &KVManager::num_tokens)
      .def("block_table", [](const KVManager& kv, int seq) { return kv.block_table(seq); })
      .def("shared_prefix_blocks", &KVManager::shared_prefix_blocks)
      .def("take_pending_copies", &KVManager::take_pending_copies)
      .def("drop_cache", &KVManager::drop_cache)
      .def("available_blocks", &KVManager::available_blocks)
      .def("stats", &KVManager::stats)
      .def_property_readonly("block_size", &KVManager::block_size);

  // ---- scheduler ----
  py::enum_<RequestKind>(m, "RequestKind")
      .value("Generate", RequestKind::Generate)
      .value("Score", RequestKind::Score)
      .value("Embed", RequestKind::Embed);
  py::enum_<FinishReason>(m, "FinishReason")
      .value("Length", FinishReason::Length)
      .value("Eos", FinishReason::Eos)
      .value("Stop", FinishReason::Stop)
      .value("PrefillOnly", FinishReason::PrefillOnly)
      .value("Aborted", FinishReason::Aborted);
  py::class_<SamplingParams>(m, "SamplingParams")
      .def(py::init<>())
      .def_readwrite("temperature", &SamplingParams::temperature)
      .def_readwrite("top_p", &SamplingParams::top_p)
      .def_readwrite("top_k", &SamplingParams::top_k)
      .def_readwrite("max_new_tokens", &SamplingParams::max_new_tokens)
      .def_readwrite("seed", &SamplingParams::seed)
      .def_readwrite("eos_id", &SamplingParams::eos_id)
      .def_readwrite("ignore_eos", &SamplingParams::ignore_eos)
      .def_readwrite("stop_ids", &SamplingParams::stop_ids);
  py::class_<SchedulerConfig>(m, "SchedulerConfig")
      .def(py::init<>())
      .def_readwrite("max_batch_seqs", &SchedulerConfig::max_batch_seqs)
      .def_readwrite("max_prefill_tokens", &SchedulerConfig::max_prefill_tokens)
      .def_readwrite("mix_decode", &SchedulerConfig::mix_decode)
      .def_readwrite("max_blocks_per_seq", &SchedulerConfig::max_blocks_per_seq)
      .def_readwrite("admit_window", &SchedulerConfig::admit_window)
      .def_readwrite("max_head_skips", &SchedulerConfig::max_head_skips)
      .def_readwrite("optimistic_admission", &SchedulerConfig::optimistic_admission)
      .def_readwrite("watermark_blocks", &SchedulerConfig::watermark_blocks);
  py::class_<ChainOutput>(m, "ChainOutput")
      .def_readonly("tokens", &ChainOutput::tokens)
      .def_readonly("cum_logprob", &ChainOutput::cum_logprob)
      .def_readonly("reason", &ChainOutput::reason);
  py::class_<FinishedRequest>(m, "FinishedRequest")
      .def_readonly("request_id", &FinishedRequest::request_id)
      .def_readonly("kind", &FinishedRequest::kind)
      .def_readonly("prompt_len", &FinishedRequest::prompt_len)
      .def_readonly("cached_tokens", &FinishedRequest::cached_tokens)
      .def_readonly("chains", &FinishedRequest::chains);
  py::class_<Scheduler::SampleSlot>(m, "SampleSlot")
      .def_readonly("last_row_index", &Scheduler::SampleSlot::last_row_index)
      .def_readonly("chain", &Scheduler::SampleSlot::chain)
      .def_readonly("seed", &Scheduler::SampleSlot::seed)
      .def_readonly("temperature", &Scheduler::SampleSlot::temperature)
      .def_readonly("top_p", &Scheduler::SampleSlot::top_p)
      .def_readonly("top_k", &Scheduler::SampleSlot::top_k);
  py::class_<StepPlan> plan(m, "StepPlan");
  py::enum_<StepPlan::Kind>(plan, "Kind")
      .value("Idle", StepPlan::Idle)
      .value("Prefill", StepPlan::Prefill)
      .value("Decode", StepPlan::Decode);
  plan.def_readonly("kind", &StepPlan::kind)
      .def_readonly("max_blocks", &StepPlan::max_blocks)
      .def_property_readonly("seq_ids", [](const StepPlan& p) { return to_array(p.seq_ids); })
      .def_property_readonly("request_ids", [](const StepPlan& p) { return to_array(p.request_ids); })
      .def_property_readonly("block_tables",
                             [](const StepPlan& p) {
                               auto a = to_array(p.block_tables);
                               const py::ssize_t B = static_cast<py::ssize_t>(p.seq_ids.size());
                               a.resize({B, static_cast<py::ssize_t>(p.max_blocks)});
                               return a;
                             })
      .def_property_readonly("context_lens", [](const StepPlan& p) { return to_array(p.context_lens); })
      .def_property_readonly("token_ids", [](const StepPlan& p) { return to_array(p.token_ids); })
      .def_property_readonly("positions", [](const StepPlan& p) { return to_array(p.positions); })
      .def_property_readonly("slot_mapping", [](const StepPlan& p) { return to_array(p.slot_mapping); })
      .def_property_readonly("cu_q", [](const StepPlan& p) { return to_array(p.cu_q); })
      .def_property_readonly("last_rows", [](const StepPlan& p) { return to_array(p.last_rows); })
      .def_property_readonly("last_row_seq", [](const StepPlan& p) { return to_array(p.last_row_seq); })
      .def_readonly("rider_first", &StepPlan::rider_first)
      .def_property_readonly("group_first", [](const StepPlan& p) { return to_array(p.group_first); })
      .def_property_readonly("group_count", [](const StepPlan& p) { return to_array(p.group_count); })
      .def_property_readonly("group_shared_blocks",
                             [](const StepPlan& p) { return to_array(p.group_shared_blocks); })
      .def_property_readonly("step_index", [](const StepPlan& p) { return to_array(p.step_index); })
      .def_property_readonly("seeds", [](const StepPlan& p) { return to_array(p.seeds); })
      .def_readonly("ahead", &StepPlan::ahead)
      .def_property_readonly("src_rows", [](const StepPlan& p) { return to_array(p.src_rows); })
      .def_property_readonly("copy_src", [](const StepPlan& p) { return to_array(p.copy_src); })
      .def_property_readonly("copy_dst", [](const StepPlan& p) { return to_array(p.copy_dst); });
  py::class_<Scheduler>(m, "Scheduler")
      .def(py::init<std::shared_ptr<KVManager>, SchedulerConfig>())
      .def("add_request",
           [](Scheduler& s, RequestKind kind, i32arr prompt, int n, const SamplingParams& p) {
             std::vector<int32_t> v(prompt.data(), prompt.data() + prompt.size());
             return s.add_request(kind, std::move(v), n, p);
           },
           py::arg("kind"), py::arg("prompt"), py::arg("n_chains") = 1,
           py::arg("params") = SamplingParams{})
      .def("abort_request", &Scheduler::abort_request)
      .def("schedule", &Scheduler::schedule)
      .def("schedule_ahead", &Scheduler::schedule_ahead)
      .def("set_mix_decode", &Scheduler::set_mix_decode)
      .def_property_readonly("mix_decode", &Scheduler::mix_decode)
      .def("prefill_sample_layout", &Scheduler::prefill_sample_layout)
      .def("complete_prefill",
           [](Scheduler& s, const StepPlan& plan, i32arr toks, f32arr lps) {
             if (toks.size() != lps.size()) throw std::invalid_argument("complete_prefill: size mismatch");
             s.complete_prefill(plan, toks.data(), lps.data(), static_cast<int>(toks.size()));
           })
      .def("complete_decode",
           [](Scheduler& s, const StepPlan& plan, i32arr toks, f32arr lps) {
             if (toks.size() != static_cast<py::ssize_t>(plan.seq_ids.size()) || lps.size() != toks.size())
               throw std::invalid_argument("complete_decode: need one token and logprob per plan row");
             return to_array(s.complete_decode(plan, toks.data(), lps.data()));
           })
      .def("take_finished", &Scheduler::take_finished)
      .def("has_work", &Scheduler::has_work)
      .def_property_readonly("num_waiting", &Scheduler::num_waiting)
      .def_property_readonly("num_running", &Scheduler::num_running)
      .def_property_readonly("num_preemptions", &Scheduler::num_preemptions)
      .def_property_readonly("num_head_skips", &Scheduler::num_head_skips)
      .def("tokens_so_far", &Scheduler::tokens_so_far)
      .def("params_of", &Scheduler::params_of, py::return_value_policy::copy)
      .def("kind_of", &Scheduler::kind_of);

  // ---- placement ----
  py::class_<MemberSpec>(m, "MemberSpec")
      .def(py::init<>())
      .def_readwrite("name", &MemberSpec::name)
      .def_readwrite("model_key", &MemberSpec::model_key)
      .def_readwrite("weight_bytes", &MemberSpec::weight_bytes)
      .def_readwrite("min_kv_bytes", &MemberSpec::min_kv_bytes)
      .def_readwrite("tp", &MemberSpec::tp)
      .def_readwrite("ep", &MemberSpec::ep);
  py::enum_<PlacementPolicy>(m, "PlacementPolicy")
      .value("Pack", PlacementPolicy::Pack)
      .value("Spread", PlacementPolicy::Spread);
  py::class_<MemberPlacement>(m, "MemberPlacement")
      .def_readonly("name", &MemberPlacement::name)
      .def_readonly("devices", &MemberPlacement::devices)
      .def_readonly("shares_weights", &MemberPlacement::shares_weights);
  py::class_<ReplicaPlan>(m, "ReplicaPlan")
      .def_readonly("devices", &ReplicaPlan::devices)
      .def_readonly("members", &ReplicaPlan::members);
  py::class_<PlacementPlan>(m, "PlacementPlan")
      .def_readonly("feasible", &PlacementPlan::feasible)
      .def_readonly("error", &PlacementPlan::error)
      .def_readonly("gpus_per_replica", &PlacementPlan::gpus_per_replica)
      .def_readonly("replicas", &PlacementPlan::replicas)
      .def_readonly("bytes_per_device", &PlacementPlan::bytes_per_device)
      .def_readonly("idle_devices", &PlacementPlan::idle_devices);
  m.def("plan_placement", &plan_placement, py::arg("members"), py::arg("num_gpus"),
        py::arg("hbm_bytes_per_gpu") = static_cast<int64_t>(288) * 1000 * 1000 * 1000,
        py::arg("reserve_fraction") = 0.08, py::arg("policy") = PlacementPolicy::Pack);

  // ---- tokenizer ----
  py::class_<ByteTokenizer>(m, "ByteTokenizer")
      .def(py::init<>())
      .def("encode", &ByteTokenizer::encode, py::arg("text"), py::arg("add_bos") = true)
      .def("decode",
           [](const ByteTokenizer& t, const std::vector<int32_t>& ids, bool fold) {
             return py::bytes(t.decode(ids, fold));
           },
           py::arg("ids"), py::arg("fold_out_of_range") = false)
      .def_property_readonly_static("bos_id", [](py::object) { return ByteTokenizer::kBos; })
      .def_property_readonly_static("eos_id", [](py::object) { return ByteTokenizer::kEos; })
      .def_property_readonly_static("pad_id", [](py::object) { return ByteTokenizer::kPad; })
      .def_property_readonly_static("vocab_size", [](py::object) { return ByteTokenizer::kVocab; });

  // ---- metrics ----
  py::class_<Metrics>(m, "Metrics")
      .def(py::init<>())
      .def("inc", &Metrics::inc, py::arg("name"), py::arg("by") = 1.0)
      .def("observe", &Metrics::observe)
      .def("counter", &Metrics::counter)
      .def("percentile", &Metrics::percentile)
      .def("mean", &Metrics::mean)
      .def("count", &Metrics::count)
      .def("snapshot", &Metrics::snapshot)
      .def("reset", &Metrics::reset);
}
