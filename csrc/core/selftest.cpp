// Stand-alone stress driver for the native core, meant to be built with
// -fsanitize=address,undefined or -fsanitize=thread (build_native.py --sanitize=...).
//
// The reference keeps all coordinator state in one actor on one thread
// (/root/reference/src/main.rs:187-208, :347-348), so it has no data races by construction
// and no tests. Here many sessions are in flight, the scheduler and KV manager juggle
// ref-counted blocks, and Metrics is shared between front-end threads and the engine
// driver; this program exercises exactly those paths:
//   1. protocol: many sessions driven to completion with random verdicts and failures;
//   2. scheduler + KV manager: random request mix (Generate with K chains, Score, Embed),
//      a fake "model" that emits random tokens, block accounting checked at the end;
//   3. threads: each thread owns its sessions/scheduler (the design's ownership rule) while
//      all of them hammer one shared Metrics object.
// Exit status 0 = all invariants held (sanitizer findings abort on their own).
#include <algorithm>
#include <cstdio>
#include <stdexcept>
#include <memory>
#include <thread>
#include <vector>

#include "kv_manager.hpp"
#include "metrics.hpp"
#include "reducers.hpp"
#include "rng.hpp"
#include "scheduler.hpp"
#include "session.hpp"
#include "verdict.hpp"

using namespace lc;

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); \
      std::abort();                                                         \
    }                                                                       \
  } while (0)

static int run_sessions(uint64_t seed, int n_sessions, Metrics* metrics) {
  SplitMix64 rng(seed);
  int forced = 0;
  const std::vector<std::string> names = {"a", "b", "c", "d"};
  for (int s = 0; s < n_sessions; ++s) {
    SessionConfig cfg;
    cfg.max_rounds = 1 + static_cast<int>(rng.below(5));
    cfg.seed = seed + s;
    cfg.failure_policy = rng.below(2) ? FailurePolicy::Abstain : FailurePolicy::NeedsRefinement;
    ConsensusSession sess(names, cfg, s);
    CHECK(!sess.ready());
    CHECK(sess.answer() == ConsensusSession::kNotReadyMessage);
    Action a = sess.ask("q" + std::to_string(s));
    CHECK(a.kind == Action::Propose && a.member >= 0 && a.member < 4);
    a = sess.on_proposal("draft");
    int guard = 0;
    while (a.kind != Action::Done) {
      CHECK(++guard < 100);
      if (a.kind == Action::Evaluate) {
        for (int m = 0; m < 4; ++m) {
          const uint64_t r = rng.below(10);
          if (r == 0) a = sess.on_member_failure(m);
          else a = sess.on_evaluation(m, r < 6 ? Verdict::Good : Verdict::NeedsRefinement, "because");
        }
      } else if (a.kind == Action::Refine) {
        CHECK(a.member >= 0 && a.member < 4);
        a = sess.on_refinement("draft v" + std::to_string(guard));
      } else {
        CHECK(false);
      }
    }
    CHECK(sess.ready());
    CHECK(sess.evaluation_count() <= cfg.max_rounds);
    CHECK(!sess.answer().empty());
    forced += sess.forced();
    metrics->inc("sessions");
    metrics->observe("rounds", sess.evaluation_count());
    sess.reset();
    CHECK(!sess.ready());
  }
  return forced;
}

static void run_scheduler(uint64_t seed, int n_requests, Metrics* metrics) {
  SplitMix64 rng(seed);
  const int num_blocks = 96, bs = 16;
  auto kv = std::make_shared<KVManager>(num_blocks, bs, /*enable_prefix_cache=*/true);
  SchedulerConfig sc;
  sc.max_batch_seqs = 24;
  sc.max_prefill_tokens = 128;
  Scheduler sched(kv, sc);
  int submitted = 0, finished = 0;
  std::vector<int32_t> toks, first;
  std::vector<float> lps;
  int steps = 0;
  while (finished < n_requests) {
    CHECK(++steps < 200000);
    while (submitted < n_requests && sched.num_waiting() < 6) {
      const int len = 1 + static_cast<int>(rng.below(90));
      std::vector<int32_t> prompt(len);
      const bool shared = rng.below(3) == 0;   // a third of the prompts share a prefix (cache hits)
      for (int i = 0; i < len; ++i) prompt[i] = shared && i < 40 ? i : static_cast<int32_t>(rng.below(500));
      SamplingParams p;
      p.max_new_tokens = 1 + static_cast<int>(rng.below(12));
      p.seed = seed + submitted;
      p.eos_id = 7;
      p.ignore_eos = rng.below(2);
      const uint64_t k = rng.below(10);
      const RequestKind kind = k < 7 ? RequestKind::Generate : (k < 9 ? RequestKind::Score : RequestKind::Embed);
      const int n = kind == RequestKind::Generate ? 1 + static_cast<int>(rng.below(4)) : 1;
      sched.add_request(kind, std::move(prompt), n, p);
      ++submitted;
    }
    StepPlan plan = sched.schedule();
    if (plan.kind == StepPlan::Idle) {
      CHECK(!sched.has_work() || submitted < n_requests);
      continue;
    }
    const int B = static_cast<int>(plan.seq_ids.size());
    CHECK(static_cast<int>(plan.context_lens.size()) == B);
    CHECK(static_cast<int>(plan.block_tables.size()) == B * plan.max_blocks);
    CHECK(plan.copy_src.size() == plan.copy_dst.size());
    for (size_t i = 0; i < plan.slot_mapping.size(); ++i)
      CHECK(plan.slot_mapping[i] >= 0 && plan.slot_mapping[i] < num_blocks * bs);
    if (plan.kind == StepPlan::Prefill) {
      CHECK(plan.token_ids.size() == plan.positions.size());
      CHECK(static_cast<int>(plan.token_ids.size()) <= sc.max_prefill_tokens);
      auto layout = sched.prefill_sample_layout(plan);
      first.assign(layout.size(), 0);
      lps.assign(layout.size(), -1.f);
      for (auto& t : first) t = static_cast<int32_t>(rng.below(500));
      sched.complete_prefill(plan, first.data(), lps.data(), static_cast<int>(layout.size()));
      metrics->inc("prefill.tokens", static_cast<double>(plan.token_ids.size()));
    } else {
      CHECK(B <= sc.max_batch_seqs);
      toks.assign(B, 0);
      lps.assign(B, -1.f);
      for (auto& t : toks) t = static_cast<int32_t>(rng.below(40));   // hits eos_id 7 now and then
      sched.complete_decode(plan, toks.data(), lps.data());
      metrics->inc("decode.tokens", B);
    }
    for (auto& f : sched.take_finished()) {
      ++finished;
      for (auto& c : f.chains) CHECK(f.kind != RequestKind::Generate || !c.tokens.empty());
      metrics->observe("chains", static_cast<double>(f.chains.size()));
    }
  }
  CHECK(!sched.has_work());
  const KVStats st = kv->stats();
  CHECK(st.blocks_free + st.blocks_cached == num_blocks);   // nothing leaked, nothing double-freed
}

// The scheduler with every switch at random: small pools (optimistic admission has to preempt and
// recompute), mixed steps (decoding chains ride along in prefill steps), decode steps planned ahead
// of their results, aborts at step boundaries. Besides the block accounting at the end, no two
// tokens of a step may be given the same KV slot.
static void run_scheduler_hard(uint64_t seed, int n_requests, Metrics* metrics) {
  SplitMix64 rng(seed);
  const int bs = rng.below(2) ? 4 : 8;
  const int num_blocks = 32 + static_cast<int>(rng.below(64));
  auto kv = std::make_shared<KVManager>(num_blocks, bs, /*enable_prefix_cache=*/rng.below(2) != 0);
  SchedulerConfig sc;
  sc.max_batch_seqs = 4 + static_cast<int>(rng.below(12));
  sc.max_prefill_tokens = 16 << rng.below(3);
  sc.optimistic_admission = rng.below(2) != 0;
  sc.mix_decode = rng.below(2) != 0;
  Scheduler sched(kv, sc);
  int submitted = 0, settled = 0;      // settled = finished or aborted
  std::vector<int> live;
  std::vector<int32_t> toks, last_out;
  std::vector<float> lps;
  bool have_pending = false;
  StepPlan pending;
  int steps = 0;
  while (settled < n_requests || sched.has_work()) {
    CHECK(++steps < 400000);
    while (submitted < n_requests && sched.num_waiting() < 4 && rng.below(3) == 0) {
      const int len = 1 + static_cast<int>(rng.below(28));
      std::vector<int32_t> prompt(len);
      const bool shared = rng.below(3) == 0;
      for (int i = 0; i < len; ++i) prompt[i] = shared ? i : static_cast<int32_t>(rng.below(500));
      SamplingParams p;
      p.max_new_tokens = 1 + static_cast<int>(rng.below(10));
      p.eos_id = 7;
      p.ignore_eos = rng.below(2);
      const uint64_t k = rng.below(10);
      const RequestKind kind = k < 7 ? RequestKind::Generate : (k < 9 ? RequestKind::Score : RequestKind::Embed);
      const int n = kind == RequestKind::Generate ? 1 + static_cast<int>(rng.below(4)) : 1;
      ++submitted;
      try {
        live.push_back(sched.add_request(kind, std::move(prompt), n, p));
      } catch (const std::exception&) {
        ++settled;                      // can never fit this pool: refused at the door
      }
    }
    if (!have_pending && !live.empty() && rng.below(12) == 0) {
      const size_t i = rng.below(live.size());
      sched.abort_request(live[i]);     // its FinishedRequest (reason Aborted) is counted below
    }
    StepPlan plan;
    if (have_pending) {
      plan = std::move(pending);
      have_pending = false;
    } else {
      plan = sched.schedule();
    }
    if (plan.kind == StepPlan::Idle) {
      for (auto& f : sched.take_finished()) {
        ++settled;
        live.erase(std::remove(live.begin(), live.end(), f.request_id), live.end());
      }
      continue;
    }
    std::vector<int32_t> slots(plan.slot_mapping.begin(), plan.slot_mapping.end());
    slots.erase(std::remove_if(slots.begin(), slots.end(), [](int32_t v) { return v < 0; }), slots.end());
    std::sort(slots.begin(), slots.end());
    CHECK(std::adjacent_find(slots.begin(), slots.end()) == slots.end());
    CHECK(slots.empty() || slots.back() < num_blocks * bs);
    if (plan.kind == StepPlan::Prefill) {
      auto layout = sched.prefill_sample_layout(plan);
      int n = static_cast<int>(layout.size());
      if (plan.rider_first >= 0) n += static_cast<int>(plan.cu_q.size()) - 1 - plan.rider_first;
      toks.assign(n, 0);
      lps.assign(n, -1.f);
      for (auto& t : toks) t = static_cast<int32_t>(rng.below(40));
      sched.complete_prefill(plan, toks.data(), lps.data(), n);
      last_out.clear();
    } else {
      const int B = static_cast<int>(plan.seq_ids.size());
      CHECK(B > 0 && B <= sc.max_batch_seqs);
      if (plan.ahead) {
        CHECK(static_cast<int>(plan.src_rows.size()) == B);
        for (int r : plan.src_rows) CHECK(r >= 0 && r < static_cast<int>(last_out.size()));
      }
      toks.assign(B, 0);
      lps.assign(B, -1.f);
      for (auto& t : toks) t = static_cast<int32_t>(rng.below(40));
      if (rng.below(2)) {
        StepPlan nxt = sched.schedule_ahead(plan);
        if (nxt.kind == StepPlan::Decode) {
          pending = std::move(nxt);
          have_pending = true;
        }
      }
      sched.complete_decode(plan, toks.data(), lps.data());
      last_out = toks;
      metrics->inc("decode.tokens", B);
    }
    for (auto& f : sched.take_finished()) {
      ++settled;
      live.erase(std::remove(live.begin(), live.end(), f.request_id), live.end());
    }
  }
  CHECK(live.empty());
  const KVStats st = kv->stats();
  CHECK(st.blocks_free + st.blocks_cached == num_blocks);
}

int main(int argc, char** argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 4;
  const int scale = argc > 2 ? std::atoi(argv[2]) : 1;
  Metrics metrics;
  std::vector<std::thread> pool;
  std::vector<int> forced(threads, 0);
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      forced[t] = run_sessions(1000 + t, 300 * scale, &metrics);
      run_scheduler(2000 + t, 150 * scale, &metrics);
      for (int k = 0; k < 12 * scale; ++k) run_scheduler_hard(3000 + 100 * t + k, 60, &metrics);
    });
  for (auto& th : pool) th.join();
  CHECK(metrics.counter("sessions") == 300.0 * scale * threads);
  CHECK(metrics.count("rounds") == static_cast<size_t>(300 * scale * threads));
  // reducers on a fixed case
  const std::vector<int32_t> votes = {1, 1, 2, -1};
  const VoteResult vr = majority_vote(votes.data(), 4, 4, TieBreak::LowestClass);
  CHECK(vr.winner == 1 && vr.count == 2);
  int f = 0;
  for (int x : forced) f += x;
  std::printf("selftest ok: %d threads, %d forced sessions, %.0f decode tokens\n", threads, f,
              metrics.counter("decode.tokens"));
  return 0;
}
