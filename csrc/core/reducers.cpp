#include "reducers.hpp"

#include <cmath>

namespace lc {

VoteResult majority_vote(const int32_t* votes, int n, int num_classes, TieBreak tie,
                         const float* weights) {
  VoteResult r;
  if (n <= 0 || num_classes <= 0) return r;
  std::vector<int> cnt(num_classes, 0);
  std::vector<float> wsum(num_classes, 0.f);   // float, voter order: the device kernel sums the same way
  std::vector<int> first(num_classes, n);
  for (int i = 0; i < n; ++i) {
    int c = votes[i];
    if (c < 0 || c >= num_classes) continue;
    ++cnt[c];
    wsum[c] += weights ? weights[i] : 1.f;
    if (first[c] == n) first[c] = i;
    ++r.valid;
  }
  if (r.valid == 0) return r;
  // Ranking keys. Weights with LowestClass / FirstVoter: a weighted plurality (weight sum, then
  // count, then the tie rule). HighestWeight: plurality by count, ties go to the larger weight
  // sum, then the lower class. No weights: count, then the tie rule.
  const bool weighted_primary = weights && tie != TieBreak::HighestWeight;
  const bool weight_secondary = tie == TieBreak::HighestWeight;
  int best = -1;
  for (int c = 0; c < num_classes; ++c) {
    if (cnt[c] == 0) continue;
    if (best < 0) {
      best = c;
      continue;
    }
    const float k1c = weighted_primary ? wsum[c] : static_cast<float>(cnt[c]);
    const float k1b = weighted_primary ? wsum[best] : static_cast<float>(cnt[best]);
    const float k2c = weighted_primary ? static_cast<float>(cnt[c]) : (weight_secondary ? wsum[c] : 0.f);
    const float k2b = weighted_primary ? static_cast<float>(cnt[best]) : (weight_secondary ? wsum[best] : 0.f);
    bool better;
    if (k1c != k1b) better = k1c > k1b;
    else if (k2c != k2b) better = k2c > k2b;
    else better = (tie == TieBreak::FirstVoter) ? first[c] < first[best] : false;   // else keep the lower id
    if (better) best = c;
  }
  r.winner = best;
  r.count = cnt[best];
  r.agreement = static_cast<float>(r.count) / static_cast<float>(r.valid);
  r.unanimous = (r.count == r.valid);
  return r;
}

std::vector<VoteResult> majority_vote_batch(const int32_t* votes, int R, int N, int num_classes,
                                            TieBreak tie, const float* weights) {
  std::vector<VoteResult> out(R);
  for (int r = 0; r < R; ++r)
    out[r] = majority_vote(votes + static_cast<size_t>(r) * N, N, num_classes, tie, weights);
  return out;
}

MedoidResult cosine_medoid(const float* E, int n, int d) {
  MedoidResult r;
  if (n <= 0 || d <= 0) return r;
  if (n == 1) {
    r.index = 0;
    r.score = 1.f;
    r.mean_pairwise = 1.f;
    return r;
  }
  std::vector<double> inv(n, 0.0);
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int k = 0; k < d; ++k) s += static_cast<double>(E[i * (size_t)d + k]) * E[i * (size_t)d + k];
    inv[i] = s > 0 ? 1.0 / std::sqrt(s) : 0.0;
  }
  std::vector<double> rowsum(n, 0.0);
  double total = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      double dot = 0;
      for (int k = 0; k < d; ++k) dot += static_cast<double>(E[i * (size_t)d + k]) * E[j * (size_t)d + k];
      double c = dot * inv[i] * inv[j];
      rowsum[i] += c;
      rowsum[j] += c;
      total += c;
    }
  int best = 0;
  for (int i = 1; i < n; ++i)
    if (rowsum[i] > rowsum[best]) best = i;
  r.index = best;
  r.score = static_cast<float>(rowsum[best] / (n - 1));
  r.mean_pairwise = static_cast<float>(total / (0.5 * n * (n - 1)));
  return r;
}

JudgeResult judge_reduce(const float* scores, int C, int J) {
  JudgeResult r;
  if (C <= 0 || J <= 0) return r;
  int best = 0;
  double best_s = -INFINITY;
  for (int c = 0; c < C; ++c) {
    double s = 0;
    for (int j = 0; j < J; ++j) s += scores[c * (size_t)J + j];
    s /= J;
    if (s > best_s) {
      best_s = s;
      best = c;
    }
  }
  r.winner = best;
  r.score = static_cast<float>(best_s);
  r.unanimous = true;
  for (int j = 0; j < J && r.unanimous; ++j) {
    int top = 0;
    for (int c = 1; c < C; ++c)
      if (scores[c * (size_t)J + j] > scores[top * (size_t)J + j]) top = c;
    if (top != best) r.unanimous = false;
  }
  return r;
}

UnanimityResult unanimity_from_logprobs(const float* lp, int N) {
  UnanimityResult r;
  for (int i = 0; i < N; ++i)
    if (lp[2 * i + 1] > lp[2 * i]) r.dissenters.push_back(i);
  r.all_good = r.dissenters.empty() && N > 0;
  return r;
}

}  // namespace lc
