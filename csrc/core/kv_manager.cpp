#include "kv_manager.hpp"

#include <algorithm>
#include <stdexcept>

namespace lc {

BlockAllocator::BlockAllocator(int num_blocks) : ref_(num_blocks, 0) {
  free_.reserve(num_blocks);
  for (int b = num_blocks - 1; b >= 0; --b) free_.push_back(b);  // pop_back hands out 0,1,2,...
}

int BlockAllocator::alloc() {
  if (free_.empty()) return -1;
  int b = free_.back();
  free_.pop_back();
  ref_[b] = 1;
  return b;
}

void BlockAllocator::incref(int b) {
  if (ref_[b] <= 0) throw std::logic_error("BlockAllocator::incref on a free block");
  ++ref_[b];
}

void BlockAllocator::decref(int b) {
  if (ref_[b] <= 0) throw std::logic_error("BlockAllocator::decref on a free block");
  if (--ref_[b] == 0) free_.push_back(b);
}

KVManager::KVManager(int num_blocks, int block_size, bool enable_prefix_cache)
    : block_size_(block_size), prefix_cache_(enable_prefix_cache), alloc_(num_blocks) {
  if (block_size <= 0) throw std::invalid_argument("KVManager: block_size must be positive");
}

uint64_t KVManager::hash_block(uint64_t prev, const int32_t* tok, int n) {
  // FNV-1a over (prev, tokens) followed by a finalising mix.
  uint64_t h = 0xcbf29ce484222325ull ^ prev;
  for (int i = 0; i < n; ++i) {
    uint32_t t = static_cast<uint32_t>(tok[i]);
    for (int b = 0; b < 4; ++b) {
      h ^= (t >> (8 * b)) & 0xffu;
      h *= 0x100000001b3ull;
    }
  }
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h ? h : 1;
}

int KVManager::alloc_or_evict() {
  int b = alloc_.alloc();
  if (b >= 0) return b;
  // Evict cache-only blocks in insertion order.
  while (!lru_.empty()) {
    uint64_t h = lru_.front();
    lru_.pop_front();
    auto it = cache_.find(h);
    if (it == cache_.end()) continue;  // stale entry
    int blk = it->second;
    if (alloc_.refcount(blk) != 1) {
      lru_.push_back(h);  // still in use by a sequence; revisit later
      // Avoid spinning forever if every cached block is in use.
      bool any_evictable = false;
      for (const auto& kv : cache_)
        if (alloc_.refcount(kv.second) == 1) {
          any_evictable = true;
          break;
        }
      if (!any_evictable) return -1;
      continue;
    }
    cache_.erase(it);
    cached_block_hash_.erase(blk);
    alloc_.decref(blk);
    ++stats_.evictions;
    return alloc_.alloc();
  }
  return -1;
}

int KVManager::create_sequence(const int32_t* tokens, int n, int* cached_tokens, bool allow_cache_hit) {
  if (n <= 0) throw std::invalid_argument("KVManager::create_sequence: empty prompt");
  Seq s;
  int matched = 0;
  if (prefix_cache_ && allow_cache_hit) {
    // At least one token must be left to run so the step produces logits.
    int max_match = (n - 1) / block_size_;
    uint64_t prev = 0;
    for (int b = 0; b < max_match; ++b) {
      uint64_t h = hash_block(prev, tokens + static_cast<size_t>(b) * block_size_, block_size_);
      auto it = cache_.find(h);
      if (it == cache_.end()) break;
      alloc_.incref(it->second);
      s.blocks.push_back(it->second);
      s.chain.push_back(h);
      prev = h;
      ++matched;
    }
  }
  int need = blocks_needed(n);
  while (static_cast<int>(s.blocks.size()) < need) {
    int b = alloc_or_evict();
    if (b < 0) {
      for (int blk : s.blocks) alloc_.decref(blk);
      return -1;
    }
    s.blocks.push_back(b);
  }
  s.num_tokens = matched * block_size_;
  if (cached_tokens) *cached_tokens = s.num_tokens;
  stats_.prompt_tokens += n;
  stats_.prefix_hit_tokens += s.num_tokens;
  int id = next_seq_++;
  seqs_.emplace(id, std::move(s));
  return id;
}

int KVManager::extend_match(int seq, const int32_t* tokens, int n) {
  if (!prefix_cache_) return 0;
  Seq& s = seqs_.at(seq);
  int b = s.num_tokens / block_size_;
  if (s.num_tokens % block_size_ != 0 || !s.tail.empty() || static_cast<int>(s.chain.size()) != b) return 0;
  const int max_match = (n - 1) / block_size_;   // leave at least one token to run
  uint64_t prev = s.chain.empty() ? 0 : s.chain.back();
  int added = 0;
  for (; b < max_match && b < static_cast<int>(s.blocks.size()); ++b) {
    uint64_t h = hash_block(prev, tokens + static_cast<size_t>(b) * block_size_, block_size_);
    auto it = cache_.find(h);
    if (it == cache_.end()) break;
    if (it->second != s.blocks[b]) {
      alloc_.incref(it->second);
      alloc_.decref(s.blocks[b]);   // never written, never cached: goes back to the free list
      s.blocks[b] = it->second;
    }
    s.chain.push_back(h);
    prev = h;
    s.num_tokens += block_size_;
    ++added;
  }
  stats_.prefix_hit_tokens += static_cast<int64_t>(added) * block_size_;
  return added * block_size_;
}

bool KVManager::reserve(int seq, int n_new) {
  Seq& s = seqs_.at(seq);
  // Copy-on-write of a shared, partially filled tail block.
  if (n_new > 0 && s.num_tokens % block_size_ != 0) {
    int ti = s.num_tokens / block_size_;
    int tail = s.blocks[ti];
    if (alloc_.refcount(tail) > 1) {
      int nb = alloc_or_evict();
      if (nb < 0) return false;
      pending_copies_.emplace_back(tail, nb);
      alloc_.decref(tail);
      s.blocks[ti] = nb;
      ++stats_.cow_copies;
    }
  }
  int need = blocks_needed(s.num_tokens + n_new);
  size_t before = s.blocks.size();
  while (static_cast<int>(s.blocks.size()) < need) {
    int b = alloc_or_evict();
    if (b < 0) {
      while (s.blocks.size() > before) {
        alloc_.decref(s.blocks.back());
        s.blocks.pop_back();
      }
      return false;
    }
    s.blocks.push_back(b);
  }
  return true;
}

void KVManager::register_full_block(Seq& s, int block_index) {
  uint64_t prev = s.chain.empty() ? 0 : s.chain.back();
  uint64_t h = hash_block(prev, s.tail.data(), block_size_);
  s.chain.push_back(h);
  if (!prefix_cache_) return;
  int blk = s.blocks[block_index];
  if (cache_.count(h) || cached_block_hash_.count(blk)) return;
  cache_.emplace(h, blk);
  cached_block_hash_.emplace(blk, h);
  alloc_.incref(blk);
  lru_.push_back(h);
}

void KVManager::commit(int seq, const int32_t* tokens, int n) {
  Seq& s = seqs_.at(seq);
  if (blocks_needed(s.num_tokens + n) > static_cast<int>(s.blocks.size()))
    throw std::logic_error("KVManager::commit past the reserved blocks");
  for (int i = 0; i < n; ++i) {
    s.tail.push_back(tokens[i]);
    ++s.num_tokens;
    if (s.num_tokens % block_size_ == 0) {
      register_full_block(s, s.num_tokens / block_size_ - 1);
      s.tail.clear();
    }
  }
}

int KVManager::fork(int seq) {
  const Seq& src = seqs_.at(seq);
  Seq s = src;
  for (int b : s.blocks) alloc_.incref(b);
  int id = next_seq_++;
  seqs_.emplace(id, std::move(s));
  return id;
}

void KVManager::free_sequence(int seq) {
  auto it = seqs_.find(seq);
  if (it == seqs_.end()) return;
  for (int b : it->second.blocks) {
    alloc_.decref(b);
    if (alloc_.refcount(b) == 0 && !pending_copies_.empty()) {
      // A copy-on-write into a block that nobody holds any more must not run: the block may
      // be handed to another sequence before the queued copies execute.
      pending_copies_.erase(std::remove_if(pending_copies_.begin(), pending_copies_.end(),
                                           [b](const std::pair<int, int>& c) { return c.second == b; }),
                            pending_copies_.end());
    }
  }
  seqs_.erase(it);
}

int KVManager::num_tokens(int seq) const { return seqs_.at(seq).num_tokens; }

const std::vector<int>& KVManager::block_table(int seq) const { return seqs_.at(seq).blocks; }

int KVManager::shared_prefix_blocks(int seq) const {
  const Seq& s = seqs_.at(seq);
  int n = 0;
  for (int b : s.blocks) {
    if (alloc_.refcount(b) <= 1) break;
    ++n;
  }
  return n;
}

std::vector<std::pair<int, int>> KVManager::take_pending_copies() {
  std::vector<std::pair<int, int>> out;
  out.swap(pending_copies_);
  return out;
}

void KVManager::drop_cache() {
  for (const auto& kv : cache_) alloc_.decref(kv.second);
  cache_.clear();
  cached_block_hash_.clear();
  lru_.clear();
}

int KVManager::available_blocks() const {
  int n = alloc_.num_free();
  for (const auto& kv : cache_)
    if (alloc_.refcount(kv.second) == 1) ++n;
  return n;
}

KVStats KVManager::stats() const {
  KVStats s = stats_;
  s.blocks_total = alloc_.num_total();
  s.blocks_free = alloc_.num_free();
  s.blocks_cached = 0;
  for (const auto& kv : cache_)
    if (alloc_.refcount(kv.second) == 1) ++s.blocks_cached;
  return s;
}

}  // namespace lc
