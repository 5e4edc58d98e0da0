// Host implementations of the consensus reducers. These are the authoritative
// semantics; the HIP kernels in csrc/kernels/consensus.hip compute the same results
// on device for batches of requests and are tested against these.
//
// The reference has a single reducer: "every member voted Good"
// (/root/reference/src/main.rs:266, :323). Majority vote, judge scoring and the
// cosine medoid are the additional reducers this engine serves.
#pragma once
#include <cstdint>
#include <vector>

namespace lc {

enum class TieBreak : uint8_t {
  LowestClass = 0,   // smallest class id among the tied
  FirstVoter = 1,    // class of the earliest voter (member order) among the tied
  HighestWeight = 2, // plurality by count; ties go to the larger summed weight, then the lower class
};

struct VoteResult {
  int winner = -1;        // class id, -1 if no valid vote
  int count = 0;          // votes for the winner
  int valid = 0;          // votes that were not -1
  float agreement = 0.f;  // count / valid
  bool unanimous = false;
};

// votes[i] in [0, num_classes) or -1 (unparsed / abstain). weights (one per voter) may be
// null. With weights and LowestClass / FirstVoter the vote is a weighted plurality (largest
// weight sum wins, then the larger count, then the tie rule); with HighestWeight the weights
// only break ties in count. count / agreement / unanimous always count voters.
VoteResult majority_vote(const int32_t* votes, int n, int num_classes, TieBreak tie,
                         const float* weights = nullptr);

// Batched: votes is [R, N] row-major; weights [N] (the voters are the same in every row) or null.
std::vector<VoteResult> majority_vote_batch(const int32_t* votes, int R, int N, int num_classes,
                                            TieBreak tie, const float* weights = nullptr);

struct MedoidResult {
  int index = -1;          // row with the largest summed cosine similarity to the others
  float score = 0.f;       // that sum / (n-1)
  float mean_pairwise = 0.f;
};
// E is [n, d] row-major; rows need not be normalised (zero rows score 0).
MedoidResult cosine_medoid(const float* E, int n, int d);

struct JudgeResult {
  int winner = -1;          // candidate with the best combined score
  float score = 0.f;
  bool unanimous = false;   // every judge ranks `winner` first
};
// scores is [C candidates, J judges] row-major (log-probabilities; larger = better).
// Combination = mean over judges.
JudgeResult judge_reduce(const float* scores, int C, int J);

// Reference semantics over two-way verdict logprobs: lp is [N, 2] = (logp Good,
// logp NeedsRefinement) per member. all_good iff every member prefers Good.
// dissenters are the members preferring NeedsRefinement.
struct UnanimityResult {
  bool all_good = false;
  std::vector<int> dissenters;
};
UnanimityResult unanimity_from_logprobs(const float* lp, int N);

}  // namespace lc
