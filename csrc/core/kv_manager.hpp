// Paged KV-cache bookkeeping: refcounted blocks, fork-on-write sequence tables and
// a hash-chained prefix cache.
//
// Nothing like this exists in the reference (its model lives behind an HTTPS call,
// /root/reference/src/main.rs:82-86). It exists here because of how the reference's
// prompts are laid out: evaluate prompts put question+answer first and the persona
// text last (main.rs:112-120), so N personas on one model share a prompt prefix, and
// self-consistency forks K chains from one prompt. Both want blocks shared by
// reference and copied only when written.
//
// The manager only does bookkeeping (block ids); device memory is owned by the
// engine. Copy-on-write requests are queued as (src, dst) block pairs that the
// engine executes with a device copy before the next kernel that writes the block.
#pragma once
#include <cstdint>
#include <deque>
#include <unordered_map>
#include <utility>
#include <vector>

namespace lc {

class BlockAllocator {
 public:
  explicit BlockAllocator(int num_blocks);
  int alloc();             // -1 when exhausted
  void incref(int b);
  void decref(int b);      // returns the block to the free list at zero
  int refcount(int b) const { return ref_[b]; }
  int num_free() const { return static_cast<int>(free_.size()); }
  int num_total() const { return static_cast<int>(ref_.size()); }

 private:
  std::vector<int> ref_;
  std::vector<int> free_;
};

struct KVStats {
  int64_t prompt_tokens = 0;        // tokens submitted through create_sequence
  int64_t prefix_hit_tokens = 0;    // of those, served from cached blocks
  int64_t cow_copies = 0;
  int64_t evictions = 0;
  int blocks_total = 0;
  int blocks_free = 0;
  int blocks_cached = 0;            // held only by the prefix cache (evictable)
};

class KVManager {
 public:
  KVManager(int num_blocks, int block_size, bool enable_prefix_cache);

  // Creates a sequence for `tokens` (the prompt). Full leading blocks found in the
  // prefix cache are attached by reference. Returns the sequence id, or -1 if
  // blocks for the rest of the prompt cannot be allocated (nothing is leaked).
  // *cached_tokens receives how many leading tokens already have KV.
  // `allow_cache_hit=false` forces every token to be recomputed (requests that need the
  // hidden state of every prompt token, e.g. embeddings).
  int create_sequence(const int32_t* tokens, int n, int* cached_tokens, bool allow_cache_hit = true);

  // Second chance at the prefix cache for a sequence that has not computed anything of its
  // own yet: blocks registered since create_sequence (for instance by a prompt scheduled
  // earlier in the same prefill step) replace the sequence's unwritten blocks. Returns the
  // number of additional leading tokens that now have KV.
  int extend_match(int seq, const int32_t* tokens, int n);

  // Ensures the sequence has room for `n_new` more tokens, allocating blocks and
  // queueing a copy-on-write if its partially filled tail block is shared.
  bool reserve(int seq, int n_new);

  // Marks `n` more tokens (ids given, for prefix hashing) as written.
  void commit(int seq, const int32_t* tokens, int n);

  // New sequence sharing every block of `seq` (refcount + 1 each).
  int fork(int seq);

  void free_sequence(int seq);

  int num_tokens(int seq) const;
  const std::vector<int>& block_table(int seq) const;
  // Number of leading blocks of `seq` that are shared with at least one other
  // live sequence or the cache (refcount > 1) — the shared-prefix extent.
  int shared_prefix_blocks(int seq) const;

  std::vector<std::pair<int, int>> take_pending_copies();

  // Forgets every prefix-cache entry (hash -> block) and gives back the cache's block
  // references. Blocks still held by live sequences stay allocated; only the promise "this
  // hash has its KV in that block" is withdrawn. Called after a failed step: blocks are
  // registered when a step is scheduled, before the step has written them, so after a failure
  // an entry may point at KV that was never produced.
  void drop_cache();

  int block_size() const { return block_size_; }
  int blocks_needed(int n_tokens) const { return (n_tokens + block_size_ - 1) / block_size_; }
  // Free + evictable blocks.
  int available_blocks() const;
  KVStats stats() const;
  bool has_sequence(int seq) const { return seqs_.count(seq) != 0; }

 private:
  struct Seq {
    std::vector<int> blocks;
    int num_tokens = 0;
    std::vector<uint64_t> chain;  // hash of each FULL block given its predecessors
    std::vector<int32_t> tail;    // token ids of the partially filled tail block
  };
  int alloc_or_evict();
  static uint64_t hash_block(uint64_t prev, const int32_t* tok, int n);
  void register_full_block(Seq& s, int block_index);

  int block_size_;
  bool prefix_cache_;
  BlockAllocator alloc_;
  std::unordered_map<int, Seq> seqs_;
  int next_seq_ = 0;
  // prefix cache: chain hash -> block id. The cache owns one reference per entry.
  std::unordered_map<uint64_t, int> cache_;
  std::unordered_map<int, uint64_t> cached_block_hash_;
  std::deque<uint64_t> lru_;  // insertion order; entries may be stale
  std::vector<std::pair<int, int>> pending_copies_;
  KVStats stats_;
};

}  // namespace lc
