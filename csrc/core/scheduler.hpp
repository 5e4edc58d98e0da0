// Continuous-batching scheduler for one member engine (one model on one device
// group). Replaces the reference's "one HTTPS call per message, one question at a
// time" execution (/root/reference/src/main.rs:82-86, :189-195): requests from many
// consensus sessions are admitted while KV blocks last, prompts are prefilled in
// token-budgeted chunks, and every running chain advances one token per decode step.
//
// A request is a *sequence group*: one prompt and `n` chains forked from it after
// prefill (n = K for self-consistency, 1 otherwise). The scheduler emits StepPlans
// made of flat int32 arrays that map 1:1 onto kernel arguments.
#pragma once
#include <cstdint>
#include <deque>
#include <memory>
#include <unordered_map>
#include <vector>

#include "kv_manager.hpp"

namespace lc {

enum class RequestKind : uint8_t {
  Generate = 0,  // sample up to max_new_tokens per chain
  Score = 1,     // prefill only; caller reads log-probs of target ids at the last position
  Embed = 2,     // prefill only; caller reads the pooled final hidden state
};

struct SamplingParams {
  float temperature = 1.0f;  // 0 => greedy
  float top_p = 1.0f;
  int top_k = 0;             // 0 => disabled
  int max_new_tokens = 16;
  uint64_t seed = 0;
  int eos_id = -1;
  bool ignore_eos = false;
  std::vector<int32_t> stop_ids;
};

struct SchedulerConfig {
  int max_batch_seqs = 256;        // decode batch ceiling
  int max_prefill_tokens = 8192;   // token budget of one prefill step
  int max_blocks_per_seq = 0;      // 0 => derived from the longest admitted request
  // Mixed steps: while prompts are being prefilled, the chains that are already decoding ride
  // along in the same step as one-token sequences, so the step's GEMMs stream the weights once
  // for both and a steady flow of requests never alternates between prefill-only and
  // decode-only steps.
  bool mix_decode = false;
  // Admission walks the waiting queue in arrival order and lets requests that fit pass one
  // that does not (a long prompt waiting for blocks does not hold up short ones behind it).
  // `admit_window` bounds how far past the head one schedule() call looks; after the head has
  // been passed `max_head_skips` times nothing else is admitted until it fits, so it cannot
  // starve.
  int admit_window = 64;
  int max_head_skips = 256;
  // false: a request is admitted only when the blocks it can need in the worst case (prompt +
  // every chain's max_new_tokens) are unpromised, so decode never runs out of blocks.
  // true: a request is admitted when the blocks its PROMPT needs (plus one per chain and
  // `watermark_blocks`) are available now; when a decode step then finds no block for a chain,
  // the youngest running request is preempted: its blocks are returned and it goes back to
  // the head of the waiting queue to be recomputed from its prompt (sampling is keyed by
  // seed and step index, so it regenerates the same tokens).
  bool optimistic_admission = false;
  int watermark_blocks = 0;
};

enum class FinishReason : uint8_t { Length = 0, Eos = 1, Stop = 2, PrefillOnly = 3, Aborted = 4 };

struct ChainOutput {
  std::vector<int32_t> tokens;
  double cum_logprob = 0.0;
  FinishReason reason = FinishReason::Length;
};

struct FinishedRequest {
  int request_id = -1;
  RequestKind kind = RequestKind::Generate;
  int prompt_len = 0;
  int cached_tokens = 0;
  std::vector<ChainOutput> chains;
};

struct StepPlan {
  enum Kind : uint8_t { Idle = 0, Prefill = 1, Decode = 2 };
  Kind kind = Idle;

  // ---- common ----
  std::vector<int32_t> seq_ids;        // [B] engine-visible sequence handles
  std::vector<int32_t> request_ids;    // [B]
  std::vector<int32_t> block_tables;   // [B, max_blocks] padded with -1
  int max_blocks = 0;
  std::vector<int32_t> context_lens;   // [B] KV length visible AFTER this step's tokens are written

  // ---- prefill ----
  std::vector<int32_t> token_ids;      // [T] flattened chunk tokens
  std::vector<int32_t> positions;      // [T]
  std::vector<int32_t> slot_mapping;   // [T] block*block_size + offset where each token's KV goes
  std::vector<int32_t> cu_q;           // [B+1] prefix sums of chunk lengths
  std::vector<int32_t> last_rows;      // rows in [0,T) whose logits/hidden are needed (prompt ends here)
  std::vector<int32_t> last_row_seq;   // [len(last_rows)] index into seq_ids
  // Mixed steps: sequences [rider_first, B) are decoding chains (one token each, their rows are
  // the last ones of the step); step_index / seeds below hold one entry per rider. -1 = none.
  int rider_first = -1;

  // ---- decode ----
  // positions[b] = index of the token being fed; slot_mapping[b] where its KV goes.
  // Shared-prefix groups: chains of one request that are all still running and whose
  // leading block-table entries are identical.
  std::vector<int32_t> group_first;    // [G] first batch row of the group (rows are contiguous)
  std::vector<int32_t> group_count;    // [G]
  std::vector<int32_t> group_shared_blocks;  // [G] number of leading blocks common to the group
  std::vector<int32_t> step_index;     // [B] how many tokens this chain has generated so far
  std::vector<uint64_t> seeds;         // [B] per-chain RNG stream id

  // A decode step planned while the step before it was still in flight (Scheduler::schedule_ahead):
  // token_ids are placeholders, src_rows[b] is the row of that earlier step whose sampled token
  // is row b's input.
  bool ahead = false;
  std::vector<int32_t> src_rows;       // [B] (ahead plans only)

  // Copy-on-write block copies that must run before this step's KV writes.
  std::vector<int32_t> copy_src, copy_dst;
};

class Scheduler {
 public:
  Scheduler(std::shared_ptr<KVManager> kv, SchedulerConfig cfg);

  // Returns the request id. Throws if the request can never fit.
  int add_request(RequestKind kind, std::vector<int32_t> prompt, int n_chains, SamplingParams p);
  void abort_request(int request_id);

  StepPlan schedule();
  // The decode step that follows `inflight` (a decode plan that has been launched and not yet
  // completed), planned without its results: every chain is assumed to go on unless it reaches
  // max_new_tokens in the step in flight. Returns an Idle plan when there is no basis for that
  // (prompts waiting or being prefilled, a mixed step in flight, no block for a chain); the
  // caller then completes the step in flight and calls schedule(). At most one step may be in
  // flight when this is called, and complete_decode must be given the plans in launch order.
  StepPlan schedule_ahead(const StepPlan& inflight);
  void set_mix_decode(bool on) { cfg_.mix_decode = on; }
  bool mix_decode() const { return cfg_.mix_decode; }

  // Prefill bookkeeping: marks the plan's chunks as computed. For groups whose
  // prompt completed, `first_tokens`/`first_logprobs` hold one sampled token per
  // chain, laid out in the order given by `prefill_sample_layout(plan)`.
  // Returns nothing; finished requests are collected by take_finished().
  void complete_prefill(const StepPlan& plan, const int32_t* first_tokens,
                        const float* first_logprobs, int n);
  // For a prefill plan: for each entry of last_rows, how many samples are needed
  // (the group's chain count for Generate, 0 otherwise) and their seeds.
  struct SampleSlot {
    int last_row_index;  // index into plan.last_rows
    int chain;           // chain number within the group
    uint64_t seed;
    float temperature, top_p;
    int top_k;
  };
  std::vector<SampleSlot> prefill_sample_layout(const StepPlan& plan) const;

  // Decode bookkeeping: one sampled token (+ its log-prob) per row of the plan. Returns, per
  // row, whether the token was taken (always, except for rows of an `ahead` plan whose chain
  // the step before it had already ended).
  std::vector<uint8_t> complete_decode(const StepPlan& plan, const int32_t* tokens, const float* logprobs);

  std::vector<FinishedRequest> take_finished();
  bool has_work() const;
  int num_waiting() const { return static_cast<int>(waiting_.size()); }
  int num_running() const;
  int64_t num_preemptions() const { return preemptions_; }
  int64_t num_head_skips() const { return head_skips_total_; }
  // Tokens generated so far by every chain of a request that is still running (empty when the
  // request is unknown, waiting, or has just been preempted): what a streaming front end shows.
  std::vector<std::vector<int32_t>> tokens_so_far(int request_id) const;
  const SamplingParams& params_of(int request_id) const;
  RequestKind kind_of(int request_id) const;

 private:
  struct Chain {
    int seq = -1;  // KVManager sequence id
    std::vector<int32_t> out;
    double cum_logprob = 0;
    bool done = false;
    FinishReason reason = FinishReason::Length;
    int32_t last_token = -1;
  };
  struct Group {
    int id = -1;
    RequestKind kind = RequestKind::Generate;
    std::vector<int32_t> prompt;
    SamplingParams p;
    int n = 1;
    int cached = 0;       // prompt tokens served by the prefix cache
    int computed = 0;     // prompt tokens with KV present (>= cached)
    int budget_blocks = 0;
    bool admitted = false;
    bool prefilled = false;
    std::vector<Chain> chains;  // chains[0] owns the prompt sequence until fork
  };
  bool try_admit(Group& g);
  void admit_waiting();
  // Makes room for one more token of every live chain of the running groups; preempts from
  // the youngest running group backwards when the pool is exhausted.
  void ensure_decode_capacity();
  bool build_decode_plan(StepPlan& plan, const std::unordered_map<int, int>* prev_rows);
  void preempt(int gid);
  void finish_chain(Group& g, Chain& c, FinishReason r);
  void maybe_retire(int gid);
  bool is_stop(const Group& g, int32_t tok, FinishReason* why) const;

  std::shared_ptr<KVManager> kv_;
  SchedulerConfig cfg_;
  std::unordered_map<int, Group> groups_;
  std::deque<int> waiting_;     // group ids not yet admitted
  std::vector<int> prefilling_; // admitted, prompt not fully computed
  std::vector<int> running_;    // prompt done, at least one chain live
  std::vector<FinishedRequest> finished_;
  int next_id_ = 0;
  int budget_used_ = 0;
  int promised_ = 0;            // chains of admitted groups (prefilling_ + running_)
  int head_skips_ = 0;          // times the current head of waiting_ has been passed
  int64_t head_skips_total_ = 0;
  int64_t preemptions_ = 0;
};

}  // namespace lc
