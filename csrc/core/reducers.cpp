#include "reducers.hpp"

#include <cmath>

namespace lc {

VoteResult majority_vote(const int32_t* votes, int n, int num_classes, TieBreak tie,
                         const float* weights) {
  VoteResult r;
  if (n <= 0 || num_classes <= 0) return r;
  std::vector<int> cnt(num_classes, 0);
  std::vector<double> wsum(num_classes, 0.0);
  std::vector<int> first(num_classes, n);
  for (int i = 0; i < n; ++i) {
    int c = votes[i];
    if (c < 0 || c >= num_classes) continue;
    ++cnt[c];
    wsum[c] += weights ? weights[i] : 1.0;
    if (first[c] == n) first[c] = i;
    ++r.valid;
  }
  if (r.valid == 0) return r;
  int best = -1;
  for (int c = 0; c < num_classes; ++c) {
    if (cnt[c] == 0) continue;
    if (best < 0) {
      best = c;
      continue;
    }
    bool better;
    if (weights) {
      // weighted vote: compare weight sums first, then counts
      if (wsum[c] != wsum[best]) better = wsum[c] > wsum[best];
      else if (cnt[c] != cnt[best]) better = cnt[c] > cnt[best];
      else better = (tie == TieBreak::FirstVoter) ? first[c] < first[best] : false;
    } else if (cnt[c] != cnt[best]) {
      better = cnt[c] > cnt[best];
    } else {
      switch (tie) {
        case TieBreak::FirstVoter: better = first[c] < first[best]; break;
        default: better = false;  // LowestClass / HighestWeight(unweighted): keep the lower id
      }
    }
    if (better) best = c;
  }
  r.winner = best;
  r.count = cnt[best];
  r.agreement = static_cast<float>(r.count) / static_cast<float>(r.valid);
  r.unanimous = (r.count == r.valid);
  return r;
}

std::vector<VoteResult> majority_vote_batch(const int32_t* votes, int R, int N, int num_classes,
                                            TieBreak tie) {
  std::vector<VoteResult> out(R);
  for (int r = 0; r < R; ++r)
    out[r] = majority_vote(votes + static_cast<size_t>(r) * N, N, num_classes, tie);
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
