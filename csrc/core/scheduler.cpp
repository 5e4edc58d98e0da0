#include "scheduler.hpp"

#include <algorithm>
#include <stdexcept>

namespace lc {

Scheduler::Scheduler(std::shared_ptr<KVManager> kv, SchedulerConfig cfg)
    : kv_(std::move(kv)), cfg_(cfg) {}

int Scheduler::add_request(RequestKind kind, std::vector<int32_t> prompt, int n_chains,
                           SamplingParams p) {
  if (prompt.empty()) throw std::invalid_argument("Scheduler::add_request: empty prompt");
  if (n_chains < 1) n_chains = 1;
  if (kind != RequestKind::Generate) {
    n_chains = 1;
    p.max_new_tokens = 0;
  }
  Group g;
  g.id = next_id_++;
  g.kind = kind;
  g.prompt = std::move(prompt);
  g.p = std::move(p);
  g.n = n_chains;
  const int bs = kv_->block_size();
  const int P = static_cast<int>(g.prompt.size());
  // Worst case: the prompt's blocks once, plus per chain a private copy of the
  // partially filled tail block and the blocks its generated tokens spill into.
  int per_chain = ((P % bs) + g.p.max_new_tokens + bs - 1) / bs;
  g.budget_blocks = P / bs + n_chains * std::max(per_chain, (P % bs) ? 1 : 0);
  if (g.budget_blocks > kv_->stats().blocks_total || n_chains > cfg_.max_batch_seqs)
    throw std::invalid_argument("Scheduler::add_request: request can never fit in the KV pool");
  // One sequence's block table has a fixed width in the step buffers of the engine: a request whose
  // prompt + answer would need more blocks is refused here, not found out when a step is packed.
  if (cfg_.max_blocks_per_seq > 0 && (P + g.p.max_new_tokens + bs - 1) / bs > cfg_.max_blocks_per_seq)
    throw std::invalid_argument("Scheduler::add_request: prompt + max_new_tokens exceed the longest sequence served");
  int id = g.id;
  groups_.emplace(id, std::move(g));
  waiting_.push_back(id);
  return id;
}

std::vector<std::vector<int32_t>> Scheduler::tokens_so_far(int request_id) const {
  std::vector<std::vector<int32_t>> out;
  auto it = groups_.find(request_id);
  if (it == groups_.end() || !it->second.prefilled) return out;
  for (const auto& c : it->second.chains) out.push_back(c.out);
  return out;
}

const SamplingParams& Scheduler::params_of(int request_id) const { return groups_.at(request_id).p; }
RequestKind Scheduler::kind_of(int request_id) const { return groups_.at(request_id).kind; }

int Scheduler::num_running() const {
  int n = 0;
  for (int gid : running_) {
    const Group& g = groups_.at(gid);
    for (const auto& c : g.chains) n += c.done ? 0 : 1;
  }
  return n;
}

bool Scheduler::has_work() const {
  return !waiting_.empty() || !prefilling_.empty() || !running_.empty();
}

bool Scheduler::try_admit(Group& g) {
  if (promised_ + g.n > cfg_.max_batch_seqs) return false;
  if (cfg_.optimistic_admission) {
    // Blocks the prompt needs now, one per chain for the first generated tokens, and the blocks
    // the chains that are already running take at their very next step (those standing at a
    // block boundary): admitting a prompt into exactly those would make the next step preempt.
    int next_step = 0;
    const int bs = kv_->block_size();
    for (int gid : running_)
      for (const auto& c : groups_.at(gid).chains)
        if (!c.done && kv_->num_tokens(c.seq) % bs == 0) ++next_step;
    const int need = kv_->blocks_needed(static_cast<int>(g.prompt.size())) + g.n + next_step + cfg_.watermark_blocks;
    if (need > kv_->available_blocks() && promised_ > 0) return false;   // an empty engine always admits
  } else if (budget_used_ + g.budget_blocks > kv_->stats().blocks_total) {
    return false;
  }
  int cached = 0;
  int seq = kv_->create_sequence(g.prompt.data(), static_cast<int>(g.prompt.size()), &cached,
                                 g.kind != RequestKind::Embed);
  if (seq < 0) return false;
  g.cached = cached;
  g.computed = cached;
  g.admitted = true;
  g.chains.assign(1, Chain{});
  g.chains[0].seq = seq;
  budget_used_ += g.budget_blocks;
  promised_ += g.n;
  return true;
}

void Scheduler::admit_waiting() {
  if (waiting_.empty() || promised_ >= cfg_.max_batch_seqs) return;
  int looked = 0;
  bool passed_head = false;
  for (auto it = waiting_.begin(); it != waiting_.end() && looked < cfg_.admit_window; ++looked) {
    const bool is_head = it == waiting_.begin();
    Group& g = groups_.at(*it);
    if (try_admit(g)) {
      prefilling_.push_back(g.id);
      it = waiting_.erase(it);
      if (is_head) head_skips_ = 0;
      else passed_head = true;
      continue;
    }
    if (is_head && head_skips_ >= cfg_.max_head_skips) break;   // the head has waited long enough
    ++it;
  }
  if (passed_head) {
    ++head_skips_;
    ++head_skips_total_;
  }
}

void Scheduler::preempt(int gid) {
  Group& g = groups_.at(gid);
  for (auto& c : g.chains)
    if (c.seq >= 0) kv_->free_sequence(c.seq);
  g.chains.clear();
  g.admitted = g.prefilled = false;
  g.cached = g.computed = 0;
  budget_used_ -= g.budget_blocks;
  promised_ -= g.n;
  running_.erase(std::remove(running_.begin(), running_.end(), gid), running_.end());
  prefilling_.erase(std::remove(prefilling_.begin(), prefilling_.end(), gid), prefilling_.end());
  waiting_.push_front(gid);
  head_skips_ = 0;
  ++preemptions_;
}

void Scheduler::ensure_decode_capacity() {
  // Oldest groups first; a group that cannot get its blocks takes them from the youngest
  // admitted request: a prompt that is still being prefilled (it has produced nothing yet),
  // then the youngest running group. When the group that comes up short IS the youngest
  // running group, it is the one that starts over: the older groups keep what they hold.
  for (size_t gi = 0; gi < running_.size(); ++gi) {
    const int gid = running_[gi];
    bool restart = true;
    while (restart) {
      restart = false;
      Group& g = groups_.at(gid);
      for (auto& c : g.chains) {
        if (c.done || kv_->reserve(c.seq, 1)) continue;
        if (!prefilling_.empty()) {
          preempt(prefilling_.back());
        } else if (running_.size() > gi + 1) {
          preempt(running_.back());
        } else if (gi > 0) {
          preempt(gid);          // running_ ends before gi now: the pass is over
          break;
        } else {
          // this is the only group left: the request alone exceeds the pool
          throw std::runtime_error("Scheduler: KV pool too small for a single running request");
        }
        restart = true;
        break;
      }
    }
  }
}

namespace {
// Writes ragged block tables as a padded [B, max_blocks] matrix.
void pack_block_tables(StepPlan& plan, const KVManager& kv) {
  const int B = static_cast<int>(plan.seq_ids.size());
  int mb = 1;
  for (int s : plan.seq_ids) mb = std::max(mb, static_cast<int>(kv.block_table(s).size()));
  plan.max_blocks = mb;
  plan.block_tables.assign(static_cast<size_t>(B) * mb, -1);
  for (int b = 0; b < B; ++b) {
    const auto& bt = kv.block_table(plan.seq_ids[b]);
    std::copy(bt.begin(), bt.end(), plan.block_tables.begin() + static_cast<size_t>(b) * mb);
  }
}
}  // namespace

StepPlan Scheduler::schedule() {
  StepPlan plan;
  admit_waiting();
  const int bs = kv_->block_size();

  // Mixed steps: the riders' blocks are secured BEFORE any prompt is laid out in the plan, so
  // that a prompt admitted moments ago (whose blocks the riders now need) can still be the
  // victim; it returns to the head of the waiting queue and the step goes on without it.
  if (cfg_.mix_decode && !prefilling_.empty() && !running_.empty()) ensure_decode_capacity();

  if (!prefilling_.empty()) {
    plan.kind = StepPlan::Prefill;
    int budget = cfg_.max_prefill_tokens;
    if (cfg_.mix_decode) {
      // Riders count against the step's token budget: the budget is sized to whole GEMM row
      // tiles, and a few hundred rows on top of it would start another tile row in every GEMM.
      int riders = 0;
      for (int gid : running_)
        for (const auto& c : groups_.at(gid).chains) riders += c.done ? 0 : 1;
      budget = std::max(budget - riders, budget / 2);
    }
    plan.cu_q.push_back(0);
    for (int gid : prefilling_) {
      if (budget <= 0) break;
      Group& g = groups_.at(gid);
      const int P = static_cast<int>(g.prompt.size());
      const int seq = g.chains[0].seq;
      if (g.computed == g.cached && g.kind != RequestKind::Embed) {
        // Prompts admitted together looked the prefix cache up before any of them had run.
        // Blocks of the prompts placed earlier in this very step are registered by now (the
        // step writes their KV before any attention reads it), so look again.
        const int more = kv_->extend_match(seq, g.prompt.data(), P);
        g.cached += more;
        g.computed += more;
      }
      const int take = std::min(P - g.computed, budget);
      if (take <= 0) continue;
      const auto& bt = kv_->block_table(seq);
      const int row0 = static_cast<int>(plan.token_ids.size());
      for (int i = 0; i < take; ++i) {
        const int pos = g.computed + i;
        plan.token_ids.push_back(g.prompt[pos]);
        plan.positions.push_back(pos);
        plan.slot_mapping.push_back(bt[pos / bs] * bs + pos % bs);
      }
      plan.seq_ids.push_back(seq);
      plan.request_ids.push_back(gid);
      plan.context_lens.push_back(g.computed + take);
      plan.cu_q.push_back(row0 + take);
      // Registered now rather than on completion: steps of one engine execute in order, so
      // whoever finds these blocks in the prefix cache reads them after this step wrote them.
      kv_->commit(seq, g.prompt.data() + g.computed, take);
      if (g.computed + take == P) {
        plan.last_rows.push_back(row0 + take - 1);
        plan.last_row_seq.push_back(static_cast<int>(plan.seq_ids.size()) - 1);
      }
      budget -= take;
    }
    if (cfg_.mix_decode && !plan.seq_ids.empty() && !running_.empty()) {
      // The riders' blocks were reserved by the capacity pass at the top of this call.
      const int first = static_cast<int>(plan.seq_ids.size());
      for (int gid : running_) {
        Group& g = groups_.at(gid);
        for (size_t k = 0; k < g.chains.size(); ++k) {
          Chain& c = g.chains[k];
          if (c.done) continue;
          if (!kv_->reserve(c.seq, 1))
            throw std::logic_error("Scheduler: block reservation lost between the capacity pass and the plan");
          const int pos = kv_->num_tokens(c.seq);
          const auto& bt = kv_->block_table(c.seq);
          const int row = static_cast<int>(plan.token_ids.size());
          plan.token_ids.push_back(c.last_token);
          plan.positions.push_back(pos);
          plan.slot_mapping.push_back(bt[pos / bs] * bs + pos % bs);
          plan.seq_ids.push_back(c.seq);
          plan.request_ids.push_back(gid);
          plan.context_lens.push_back(pos + 1);
          plan.cu_q.push_back(row + 1);
          plan.step_index.push_back(static_cast<int32_t>(c.out.size()));
          plan.seeds.push_back(g.p.seed + k);
        }
      }
      if (static_cast<int>(plan.seq_ids.size()) > first) {
        plan.rider_first = first;
        auto copies = kv_->take_pending_copies();
        for (auto& c : copies) {
          plan.copy_src.push_back(c.first);
          plan.copy_dst.push_back(c.second);
        }
      }
    }
    pack_block_tables(plan, *kv_);
    return plan;
  }

  if (!running_.empty()) {
    ensure_decode_capacity();
  }
  if (!running_.empty()) build_decode_plan(plan, nullptr);
  return plan;
}

bool Scheduler::build_decode_plan(StepPlan& plan, const std::unordered_map<int, int>* prev_rows) {
  // prev_rows == nullptr: the ordinary decode step (blocks were reserved by the capacity pass).
  // Otherwise the step AFTER one that is still in flight: every chain is one token further than
  // the bookkeeping says, its input token is not known yet (src_rows names the row of the step
  // in flight that samples it), and chains which that step ends by length are left out.
  const int bs = kv_->block_size();
  const int ahead = prev_rows ? 1 : 0;
  plan.kind = StepPlan::Decode;
  plan.ahead = ahead != 0;
  for (int gid : running_) {
    Group& g = groups_.at(gid);
    const int first = static_cast<int>(plan.seq_ids.size());
    int live = 0;
    int min_pos = 1 << 30;
    for (size_t k = 0; k < g.chains.size(); ++k) {
      Chain& c = g.chains[k];
      if (c.done) continue;
      int src = -1;
      if (ahead) {
        if (static_cast<int>(c.out.size()) + 1 >= g.p.max_new_tokens) continue;   // ends in the step in flight
        auto it = prev_rows->find(c.seq);
        if (it == prev_rows->end()) return false;          // not part of the step in flight: no basis to run ahead
        src = it->second;
      }
      if (!kv_->reserve(c.seq, 1 + ahead)) {
        if (ahead) return false;
        throw std::logic_error("Scheduler: block reservation lost between the capacity pass and the plan");
      }
      const int pos = kv_->num_tokens(c.seq) + ahead;
      const auto& bt = kv_->block_table(c.seq);
      plan.seq_ids.push_back(c.seq);
      plan.request_ids.push_back(gid);
      plan.token_ids.push_back(ahead ? 0 : c.last_token);
      if (ahead) plan.src_rows.push_back(src);
      plan.positions.push_back(pos);
      plan.slot_mapping.push_back(bt[pos / bs] * bs + pos % bs);
      plan.context_lens.push_back(pos + 1);
      plan.step_index.push_back(static_cast<int32_t>(c.out.size()) + ahead);
      plan.seeds.push_back(g.p.seed + k);
      min_pos = std::min(min_pos, pos);
      ++live;
    }
    if (live == 0) continue;
    int shared = 0;
    if (live > 1) {
      // Longest common prefix of the live chains' block tables, limited to blocks
      // that are complete and no longer written by anyone.
      const int limit = min_pos / bs;
      const auto& bt0 = kv_->block_table(plan.seq_ids[first]);
      shared = limit;
      for (int r = first + 1; r < first + live; ++r) {
        const auto& bt = kv_->block_table(plan.seq_ids[r]);
        int m = 0;
        while (m < shared && bt[m] == bt0[m]) ++m;
        shared = m;
      }
    }
    plan.group_first.push_back(first);
    plan.group_count.push_back(live);
    plan.group_shared_blocks.push_back(shared);
  }
  auto copies = kv_->take_pending_copies();
  for (auto& c : copies) {
    plan.copy_src.push_back(c.first);
    plan.copy_dst.push_back(c.second);
  }
  pack_block_tables(plan, *kv_);
  return true;
}

StepPlan Scheduler::schedule_ahead(const StepPlan& inflight) {
  StepPlan plan;
  // Only a pure decode step with nothing waiting is continued blindly: a waiting or half-prefilled
  // prompt wants the slots and blocks that the step in flight may free, and a mixed step's rows
  // are not a decode batch.
  if (inflight.kind != StepPlan::Decode || !waiting_.empty() || !prefilling_.empty() || running_.empty())
    return plan;
  std::unordered_map<int, int> rows;
  rows.reserve(inflight.seq_ids.size() * 2);
  for (size_t r = 0; r < inflight.seq_ids.size(); ++r) rows.emplace(inflight.seq_ids[r], static_cast<int>(r));
  // A refusal half way leaves the blocks it reserved with their sequences (the ordinary step
  // reserves the same ones next) and any queued block copy in the KV manager's queue.
  if (!build_decode_plan(plan, &rows) || plan.seq_ids.empty()) return StepPlan();
  return plan;
}

std::vector<Scheduler::SampleSlot> Scheduler::prefill_sample_layout(const StepPlan& plan) const {
  std::vector<SampleSlot> out;
  for (size_t j = 0; j < plan.last_rows.size(); ++j) {
    const Group& g = groups_.at(plan.request_ids[plan.last_row_seq[j]]);
    if (g.kind != RequestKind::Generate || g.p.max_new_tokens <= 0) continue;
    for (int k = 0; k < g.n; ++k)
      out.push_back({static_cast<int>(j), k, g.p.seed + static_cast<uint64_t>(k), g.p.temperature,
                     g.p.top_p, g.p.top_k});
  }
  return out;
}

bool Scheduler::is_stop(const Group& g, int32_t tok, FinishReason* why) const {
  if (!g.p.ignore_eos && g.p.eos_id >= 0 && tok == g.p.eos_id) {
    *why = FinishReason::Eos;
    return true;
  }
  for (int32_t s : g.p.stop_ids)
    if (tok == s) {
      *why = FinishReason::Stop;
      return true;
    }
  return false;
}

void Scheduler::finish_chain(Group& g, Chain& c, FinishReason r) {
  (void)g;
  c.done = true;
  c.reason = r;
  if (c.seq >= 0) {
    kv_->free_sequence(c.seq);
    c.seq = -1;
  }
}

void Scheduler::maybe_retire(int gid) {
  Group& g = groups_.at(gid);
  for (const auto& c : g.chains)
    if (!c.done) return;
  FinishedRequest f;
  f.request_id = gid;
  f.kind = g.kind;
  f.prompt_len = static_cast<int>(g.prompt.size());
  f.cached_tokens = g.cached;
  for (auto& c : g.chains) f.chains.push_back({std::move(c.out), c.cum_logprob, c.reason});
  finished_.push_back(std::move(f));
  budget_used_ -= g.budget_blocks;
  promised_ -= g.n;
  running_.erase(std::remove(running_.begin(), running_.end(), gid), running_.end());
  prefilling_.erase(std::remove(prefilling_.begin(), prefilling_.end(), gid), prefilling_.end());
  groups_.erase(gid);
}

void Scheduler::complete_prefill(const StepPlan& plan, const int32_t* first_tokens,
                                 const float* first_logprobs, int n) {
  if (plan.kind != StepPlan::Prefill) throw std::logic_error("complete_prefill: not a prefill plan");
  int cursor = 0;  // into first_tokens, in prefill_sample_layout order
  std::vector<int> completed;
  size_t last_j = 0;
  const size_t n_prefill = plan.rider_first >= 0 ? static_cast<size_t>(plan.rider_first) : plan.seq_ids.size();
  for (size_t i = 0; i < n_prefill; ++i) {
    const int gid = plan.request_ids[i];
    Group& g = groups_.at(gid);
    const int take = plan.cu_q[i + 1] - plan.cu_q[i];
    g.computed += take;
    const bool ends = last_j < plan.last_row_seq.size() && plan.last_row_seq[last_j] == static_cast<int>(i);
    if (!ends) continue;
    ++last_j;
    g.prefilled = true;
    completed.push_back(gid);
    if (g.kind != RequestKind::Generate || g.p.max_new_tokens <= 0) {
      finish_chain(g, g.chains[0], FinishReason::PrefillOnly);
      continue;
    }
    const int seq0 = g.chains[0].seq;
    g.chains.resize(g.n);
    for (int k = 1; k < g.n; ++k) g.chains[k].seq = kv_->fork(seq0);
    for (int k = 0; k < g.n; ++k) {
      if (cursor >= n) throw std::invalid_argument("complete_prefill: too few sampled tokens");
      Chain& c = g.chains[k];
      const int32_t tok = first_tokens[cursor];
      c.cum_logprob += first_logprobs ? first_logprobs[cursor] : 0.0;
      ++cursor;
      c.out.push_back(tok);
      c.last_token = tok;
      FinishReason why;
      if (is_stop(g, tok, &why)) finish_chain(g, c, why);
      else if (static_cast<int>(c.out.size()) >= g.p.max_new_tokens) finish_chain(g, c, FinishReason::Length);
    }
  }
  // Riders: their sampled tokens follow those of the completed prompts, in plan order. Handled
  // before the newly prefilled groups join running_, exactly as a decode step would.
  std::vector<int> touched;
  for (size_t i = n_prefill; i < plan.seq_ids.size(); ++i) {
    const int gid = plan.request_ids[i];
    Group& g = groups_.at(gid);
    Chain* chain = nullptr;
    for (auto& c : g.chains)
      if (!c.done && c.seq == plan.seq_ids[i]) chain = &c;
    if (!chain) throw std::logic_error("complete_prefill: rider does not match scheduler state");
    if (cursor >= n) throw std::invalid_argument("complete_prefill: too few sampled tokens");
    Chain& c = *chain;
    kv_->commit(c.seq, &c.last_token, 1);
    const int32_t tok = first_tokens[cursor];
    c.cum_logprob += first_logprobs ? first_logprobs[cursor] : 0.0;
    ++cursor;
    c.out.push_back(tok);
    c.last_token = tok;
    FinishReason why;
    if (is_stop(g, tok, &why)) finish_chain(g, c, why);
    else if (static_cast<int>(c.out.size()) >= g.p.max_new_tokens) finish_chain(g, c, FinishReason::Length);
    if (touched.empty() || touched.back() != gid) touched.push_back(gid);
  }
  for (int gid : touched) maybe_retire(gid);
  for (int gid : completed) {
    prefilling_.erase(std::remove(prefilling_.begin(), prefilling_.end(), gid), prefilling_.end());
    running_.push_back(gid);
    maybe_retire(gid);
  }
}

std::vector<uint8_t> Scheduler::complete_decode(const StepPlan& plan, const int32_t* tokens, const float* logprobs) {
  if (plan.kind != StepPlan::Decode) throw std::logic_error("complete_decode: not a decode plan");
  std::vector<uint8_t> applied(plan.seq_ids.size(), 0);
  std::vector<int> touched;
  auto apply = [&](Group& g, Chain& c, size_t row) {
    kv_->commit(c.seq, &c.last_token, 1);
    const int32_t tok = tokens[row];
    c.cum_logprob += logprobs ? logprobs[row] : 0.0;
    c.out.push_back(tok);
    c.last_token = tok;
    FinishReason why;
    if (is_stop(g, tok, &why)) finish_chain(g, c, why);
    else if (static_cast<int>(c.out.size()) >= g.p.max_new_tokens) finish_chain(g, c, FinishReason::Length);
  };
  if (plan.ahead) {
    // The plan was made while the step before it was still running: a chain which that step
    // ended (stop token) has a row here whose result is dropped, and its group may be gone.
    for (size_t row = 0; row < plan.seq_ids.size(); ++row) {
      const int gid = plan.request_ids[row];
      auto it = groups_.find(gid);
      if (it == groups_.end()) continue;
      Group& g = it->second;
      for (auto& c : g.chains) {
        if (c.done || c.seq != plan.seq_ids[row]) continue;
        apply(g, c, row);
        applied[row] = 1;
        if (touched.empty() || touched.back() != gid) touched.push_back(gid);
        break;
      }
    }
  } else {
    size_t row = 0;
    for (size_t gi = 0; gi < plan.group_first.size(); ++gi) {
      const int gid = plan.request_ids[plan.group_first[gi]];
      Group& g = groups_.at(gid);
      touched.push_back(gid);
      for (auto& c : g.chains) {
        if (c.done) continue;
        if (row >= plan.seq_ids.size() || plan.seq_ids[row] != c.seq)
          throw std::logic_error("complete_decode: plan does not match scheduler state");
        apply(g, c, row);
        applied[row] = 1;
        ++row;
      }
    }
  }
  for (int gid : touched) maybe_retire(gid);
  return applied;
}

void Scheduler::abort_request(int request_id) {
  auto it = groups_.find(request_id);
  if (it == groups_.end()) return;
  Group& g = it->second;
  if (!g.admitted) {
    waiting_.erase(std::remove(waiting_.begin(), waiting_.end(), request_id), waiting_.end());
    FinishedRequest f;
    f.request_id = request_id;
    f.kind = g.kind;
    f.prompt_len = static_cast<int>(g.prompt.size());
    f.chains.push_back({{}, 0.0, FinishReason::Aborted});
    finished_.push_back(std::move(f));
    groups_.erase(it);
    return;
  }
  for (auto& c : g.chains)
    if (!c.done) finish_chain(g, c, FinishReason::Aborted);
  maybe_retire(request_id);
}

std::vector<FinishedRequest> Scheduler::take_finished() {
  std::vector<FinishedRequest> out;
  out.swap(finished_);
  return out;
}

}  // namespace lc
